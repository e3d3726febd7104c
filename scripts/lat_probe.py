"""Single-frame step time of the captured graph under a few switches (run several times with different env)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
eng = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=3)
for s in range(3):
    eng.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
for _ in range(20):
    eng.detect(0)
t0 = time.perf_counter()
for _ in range(200):
    eng.submit(0, 1, h2d=False); eng.wait()
t_res = (time.perf_counter() - t0) / 200
t0 = time.perf_counter()
for _ in range(200):
    eng.detect(0)
t_det = (time.perf_counter() - t0) / 200
res = []
for depth in (1, 2):
    for j in range(depth):
        eng.submit(j, 1, async_upload=True)
    t0 = time.perf_counter()
    n = 300
    for i in range(n):
        eng.submit((i + depth) % 3, 1, async_upload=True)
        eng.wait_slots(i % 3, 1)
    eng.wait()
    res.append((time.perf_counter() - t0) / n)
print(f"[{os.environ.get('TAG', '')}] streams {eng.num_streams}; step (HBM resident) {t_res*1e3:.4f} ms; detect (H2D inclusive) {t_det*1e3:.4f} ms; pipelined 3 slots: "
      f"2 in flight {res[0]*1e3:.4f} ms/frame = {1/res[0]:.0f} FPS, 3 in flight {res[1]*1e3:.4f} ms/frame = {1/res[1]:.0f} FPS", flush=True)
eng.close()
