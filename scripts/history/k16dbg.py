import os, sys
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE"); os.environ.setdefault("OMP_NUM_THREADS", "8")
sys.path.insert(0, '/root/repo')
import numpy as np
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
from oracle import oracle
oracle.build()
blob = weights.synthetic_blob(0)
f = frames.synthetic_frame(0)
h32 = oracle.Net(blob).forward(oracle.preprocess(f, 640))
res = {}
for mode in ("1", "0", "1"):
    os.environ["IRMV_FUSED_HEAD"] = mode
    for slots in (1, 4):
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=slots) as e:
            names = [st["name"] for st in e.profile(0, 1) if "c16" in st["name"] or "kpt" in st["name"] or "f32" in st["name"]]
            for s in range(slots): e.get_src_image_buffer(s)[:] = f
            e.submit(0, slots); e.wait()
            h = e.read_head(0)
            e.detect(0)
            h1 = e.read_head(0)
            k = h[:, 78:]
            print(mode, slots, names, "kpt max|d| vs oracle: batched", np.abs(h[:, 78:] - h32[:, 78:]).max(), "detect", np.abs(h1[:, 78:] - h32[:, 78:]).max(),
                  "per level", [float(np.abs(h1[a:b, 78:] - h32[a:b, 78:]).max()) for a, b in ((0, 6400), (6400, 8000), (8000, 8400))], "box", np.abs(h1[:, :64] - h32[:, :64]).max(), flush=True)
            bad = np.argwhere(np.abs(h1[:, 78:] - h32[:, 78:]) > 0.05)
            print("   bad count", len(bad), "first", bad[:6].tolist(), "gpu", h1[bad[0][0], 78:] if len(bad) else None, "orc", h32[bad[0][0], 78:] if len(bad) else None)
