"""Time of the classical light extraction inside a bbox-only step (reference configuration)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
B = int(os.environ.get("SLOTS", "32"))
eng = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0, nk=0), num_slots=B, num_streams=1)
for s in range(B):
    eng.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
eng.submit(0, B); eng.wait()
nd = [len(eng.results(s)) for s in range(B)]
runs = [eng.profile(0, B) for _ in range(4)][1:]
for nm in ("light_extract", "nms_pnp", "decode"):
    t = min(st["ms"] for r in runs for st in r if st["name"] == nm)
    print(f"{nm:14s} {t*1e3:8.1f} us")
print("detections per frame: mean", np.mean(nd), "max", max(nd))
areas = []
for s in range(B):
    for a in eng.results(s):
        x1, y1, x2, y2 = a.bbox_xyxy
        areas.append(max(0, min(x2, 1280) - max(x1, 0)) * max(0, min(y2, 1024) - max(y1, 0)))
print("ROI area mean", np.mean(areas), "max", np.max(areas))
import time
t0 = time.perf_counter()
for _ in range(30): eng.submit(0, B, h2d=False)
eng.wait(); dt = (time.perf_counter() - t0) / 30
print(f"graph step {dt*1e3:.3f} ms -> {B/dt:.0f} FPS")

# realistic ROIs: boxes around the bright structures of a frame (what a trained detector would hand over)
import scipy.ndimage as ndi
from oracle import oracle
frame = frames.synthetic_frame(1)
rot = oracle.rotate180(frame)
lab, n = ndi.label(rot.max(2) >= 200, structure=np.ones((3, 3)))
sl = ndi.find_objects(lab)
boxes = np.array([(s[1].start - 60, s[0].start - 20, s[1].stop + 60, s[0].stop + 20) for s in sl][:16], np.float32)
eng.get_src_image_buffer(0)[:] = frame
eng.extract_armors(boxes)
t0 = time.perf_counter()
for _ in range(20):
    arm = eng.extract_armors(boxes)
dt = (time.perf_counter() - t0) / 20
print(f"extract_armors API: {len(boxes)} boxes, mean ROI {np.mean((boxes[:,2]-boxes[:,0])*(boxes[:,3]-boxes[:,1])):.0f} px, {dt*1e3:.3f} ms per call (frame H2D included), "
      f"lights found {[a.n_lights for a in arm]}")
