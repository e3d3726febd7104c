"""GPU head vs the fp32 oracle and vs the fp16-emulating oracle, per frame and per head section (diagnostic for the
tolerances of tests/test_gpu_engine.py).  Also prints decoded-box / keypoint differences per stride on shared survivors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
from oracle import oracle
oracle.build()
blob = weights.synthetic_blob(0)
net = oracle.Net(blob)
worst = {8: [0, 0], 16: [0, 0], 32: [0, 0]}
tot = [0, 0, 0]
with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
    for fi in [0, 1, 2, 3, 10, 11, 101, 103, 200, 263, 327]:
        f = frames.synthetic_frame(fi)
        e.get_src_image_buffer()[:] = f
        e.detect()
        hg = e.read_head(0)
        raw = e.read_raw(0)
        x = oracle.preprocess(f, 640)
        h32, h16 = net.forward(x), net.forward(x, emulate_fp16=True)
        d = [np.abs(hg - h32), np.abs(hg - h16), np.abs(h16 - h32)]
        for i in range(3):
            tot[i] = max(tot[i], float(d[i].max()))
        a = np.unravel_index(np.argmax(d[0]), d[0].shape)
        lvl = 0 if a[0] < 6400 else (1 if a[0] < 8000 else 2)
        sec = "box" if a[1] < 64 else ("cls" if a[1] < 78 else "kpt")
        print(f"frame {fi:3d}: gpu-fp32 {d[0].max():.4f} (mean {d[0].mean():.5f}; worst at level {lvl} {sec} ch {a[1]}, value {h32[a]:.2f})  "
              f"gpu-emu {d[1].max():.4f}  emu-fp32 {d[2].max():.4f}  | box {d[0][:, :64].max():.4f} cls {d[0][:, 64:78].max():.4f} kpt {d[0][:, 78:].max():.4f}", flush=True)
        ref = oracle.decode_nms(h32, 640, 14, 8)
        gi = {(int(a_), int(c)): i for i, (a_, c) in enumerate(zip(raw["anchors"], raw["classes"]))}
        for i, (a_, c) in enumerate(zip(ref["anchors"], ref["classes"])):
            j = gi.get((int(a_), int(c)))
            if j is None:
                continue
            s = 8 if a_ < 6400 else (16 if a_ < 8000 else 32)
            worst[s][0] = max(worst[s][0], float(np.abs(raw["boxes"][j] - ref["boxes"][i]).max()))
            worst[s][1] = max(worst[s][1], float(np.abs(raw["kpts"][j] - ref["kpts"][i]).max()))
print("max over frames: gpu-fp32 %.4f  gpu-emu %.4f  emu-fp32 %.4f" % tuple(tot))
print("shared survivors, max |d box| / |d kpt| px per stride:", worst)
