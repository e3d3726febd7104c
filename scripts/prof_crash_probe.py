"""Which hand-off path survives `rocprofv3 --kernel-trace`?  Prints before each phase (flushed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
def say(m): print(f"[{os.environ.get('TAG','')}] {m}", flush=True)
eng = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=4)
for s in range(4):
    eng.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
say("engine up")
for _ in range(5): eng.detect(0)
say("detect ok")
for _ in range(5):
    eng.submit(0, 1, h2d=False); eng.wait()
say("submit hbm-resident ok")
for _ in range(5):
    eng.submit(0, 4, h2d=True); eng.wait()
say("batched inline h2d ok")
for _ in range(5):
    eng.submit(0, 4, h2d=True, async_upload=True); eng.wait()
say("batched async upload ok")
eng.submit(0, 1, async_upload=True)
for i in range(12):
    eng.submit((i + 1) % 3, 1, async_upload=True)
    eng.wait_slots(i % 3, 1)
eng.wait()
say("pipelined async upload ok")
eng.close()
say("closed")
