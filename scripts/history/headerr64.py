import os, sys, time
os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
import numpy as np
sys.path.insert(0, '/root/repo')
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
from oracle import oracle
oracle.build()
blob = weights.synthetic_blob(0)
net = oracle.Net(blob)
oracle.lib().orc_set_threads(8)
t_start = time.time()
errs = []
with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
    for fi in list(range(12, 40)):
        if time.time() - t_start > 200: break
        f = frames.synthetic_frame(fi)
        e.get_src_image_buffer()[:] = f
        e.detect()
        hg = e.read_head(0)
        h32 = net.forward(oracle.preprocess(f, 640))
        d = float(np.abs(hg - h32).max())
        errs.append((d, fi)); print(fi, round(d, 4), flush=True)
errs.sort(reverse=True)
print("worst ten (max|d|, frame):", [(round(d, 4), fi) for d, fi in errs[:10]])
print("frames over 0.03:", [(round(d, 4), fi) for d, fi in errs if d > 0.03], "of", len(errs), "; median", round(float(np.median([d for d, _ in errs])), 4))
