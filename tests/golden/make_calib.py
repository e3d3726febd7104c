"""Calibrate the synthetic-weight generator (writes irmv_detection_amd/data/synth_calib.json, or
synth_calib_shuffle.json with `--backbone 1`).

No trained YOLOv8n weights exist offline, so parity and
benchmarks run on seeded random weights.  Uncalibrated, 70-odd random conv
layers blow activations up by ~10^2 and the class head produces either no
candidate or a hundred thousand.  This script walks the torch-CPU statement of
the graph once on synthetic frame 0 and records, per layer, the gain that makes
the pre-activation standard deviation 1, and per Detect level the class bias that
lets a target number of (anchor, class) pairs pass score_thr = 0.25.  The result
is plain data (72 gains + 3 biases); irmv_detection_amd.weights multiplies its
seeded N(0,1) draws by them, so weight generation itself stays bit-deterministic
and torch-free.

Run:  python tests/golden/make_calib.py [--backbone 1]
"""
from __future__ import annotations

import json
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from irmv_detection_amd import arch, frames, weights  # noqa: E402
from oracle import oracle  # noqa: E402
from torch_ref import TorchNet  # noqa: E402

TARGET_PAIRS = {"0": 220, "1": 120, "2": 60}      # per Detect level, ~400 total
SCORE_THR = 0.25


class CalibNet(TorchNet):
    def __init__(self, blob):
        super().__init__(blob)
        self.scale = {}
        self.cls_logits = {}

    def conv(self, name, x):
        sp, w, b = self.p[name]
        y = F.conv2d(x, w, None, stride=sp.stride, padding=sp.k // 2, groups=sp.groups)
        parts = name.split(".")
        final = parts[1] == "22" and parts[4] == "2"
        target = 0.15 if (final and parts[2] == "cv4") else 1.0
        s = target / float(y.std())
        self.scale[name] = s
        y = y * s
        if final and parts[2] == "cv3":
            self.cls_logits[parts[3]] = y.clone()      # bias-free logits
            return y                                   # bias decided below
        y = y + b.view(1, -1, 1, 1)
        return F.silu(y) if sp.act == 1 else y


def main():
    backbone = int(sys.argv[sys.argv.index("--backbone") + 1]) if "--backbone" in sys.argv else arch.BACKBONE_C2F
    blob = weights.synthetic_blob(0, use_calib=False, backbone=backbone)
    net = CalibNet(blob)
    x = oracle.preprocess(frames.synthetic_frame(0), arch.NET_SIZE)
    net.forward(torch.from_numpy(x))
    gains = {}
    for sp in arch.conv_specs(backbone=backbone):
        default = 4.0 if sp.name == "model.0.conv" else (1.0 if sp.groups > 1 else 1.68)
        gains[sp.name] = round(default * net.scale[sp.name], 6)
    lt = math.log(SCORE_THR / (1 - SCORE_THR))
    cls_bias = {}
    for lvl, z in net.cls_logits.items():
        v = np.sort(z.numpy().reshape(-1))[::-1]
        k = TARGET_PAIRS[lvl]
        cls_bias[lvl] = round(float(lt - 0.5 * (v[k - 1] + v[k])), 6)
    out = dict(seed=0, frame=0, score_thr=SCORE_THR, gain=gains, cls_bias=cls_bias)
    path = os.path.join(ROOT, "irmv_detection_amd", "data", "synth_calib_shuffle.json" if backbone == arch.BACKBONE_SHUFFLE else "synth_calib.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, "cls_bias", cls_bias)


if __name__ == "__main__":
    main()
