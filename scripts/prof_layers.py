"""Per-layer HIP-event timings of one step (eager replay), sorted by time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
B = int(os.environ.get("SLOTS", "16"))
eng = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=B, num_streams=int(os.environ.get('STREAMS', '1')))
for s in range(B):
    eng.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
eng.submit(0, B); eng.wait()
runs = [eng.profile(0, B) for _ in range(6)][1:]
n = len(runs[0])
ms = np.array([[r[i]["ms"] for i in range(n)] for r in runs]).min(0)
tot = ms.sum()
print(f"B={B} sum of kernel times {tot*1e3:.1f} us")
rows = sorted(range(n), key=lambda i: -ms[i])
for i in rows[:int(os.environ.get("TOP", "45"))]:
    st = runs[0][i]
    tf = st["flops"] / (ms[i] * 1e-3) / 1e12 if st["flops"] else 0
    print(f"{st['layer']:22s} {st['name']:28s} {ms[i]*1e3:8.1f} us  {tf:7.1f} TF/s  {st['bytes']/(ms[i]*1e-3)/1e9:8.0f} GB/s")
import time
t0=time.perf_counter()
for _ in range(50): eng.submit(0, B, h2d=False)
eng.wait(); dt=(time.perf_counter()-t0)/50
print(f"graph step {dt*1e3:.3f} ms -> {B/dt:.0f} FPS")
