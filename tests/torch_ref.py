"""Independent torch-CPU statement of the YOLOv8n graph and of its ShuffleNetV2-backbone variant
(irmv_detection_amd/arch.py BACKBONE_SHUFFLE).

Test infrastructure: used to cross-check oracle/orc_net.c (a different code
base: F.conv2d / F.max_pool2d / F.interpolate vs hand-written C loops) and by
tests/golden/make_calib.py.  Never imported by the product package.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from irmv_detection_amd import weights as W


class TorchNet:
    def __init__(self, blob: bytes, dtype=torch.float32):
        hdr, layers = W.parse_blob(blob)
        self.nc, self.nk, self.backbone = hdr["nc"], hdr["nk"], hdr["backbone"]
        self.p = {}
        for sp, w, b in layers:
            wt = torch.from_numpy(w.astype("float32")).permute(0, 3, 1, 2).contiguous().to(dtype)  # OIHW
            self.p[sp.name] = (sp, wt, torch.from_numpy(b.copy()).to(dtype))
        self.taps = {}

    def conv(self, name, x):
        sp, w, b = self.p[name]
        y = F.conv2d(x, w, b, stride=sp.stride, padding=sp.k // 2, groups=sp.groups)
        return F.silu(y) if sp.act == 1 else y

    @staticmethod
    def shuffle(x):
        """channel shuffle, 2 groups: [a0 .. a(n-1), b0 .. b(n-1)] -> [a0, b0, a1, b1, ...]"""
        n, c, h, w = x.shape
        return x.view(n, 2, c // 2, h, w).transpose(1, 2).reshape(n, c, h, w)

    def shuffle_down(self, prefix, x):
        b1 = self.conv(f"{prefix}.b1.pw", self.conv(f"{prefix}.b1.dw", x))
        b2 = self.conv(f"{prefix}.b2.pw2", self.conv(f"{prefix}.b2.dw", self.conv(f"{prefix}.b2.pw1", x)))
        return self.shuffle(torch.cat([b1, b2], 1))

    def shuffle_unit(self, prefix, x):
        x1, x2 = x.chunk(2, 1)
        b2 = self.conv(f"{prefix}.b2.pw2", self.conv(f"{prefix}.b2.dw", self.conv(f"{prefix}.b2.pw1", x2)))
        return self.shuffle(torch.cat([x1, b2], 1))

    def c2f(self, prefix, x, n, shortcut):
        y = list(self.conv(f"{prefix}.cv1", x).chunk(2, 1))
        for i in range(n):
            t = self.conv(f"{prefix}.m.{i}.cv2", self.conv(f"{prefix}.m.{i}.cv1", y[-1]))
            y.append(y[-1] + t if shortcut else t)
        return self.conv(f"{prefix}.cv2", torch.cat(y, 1))

    def sppf(self, x):
        a = self.conv("model.9.cv1", x)
        p1 = F.max_pool2d(a, 5, 1, 2)
        p2 = F.max_pool2d(p1, 5, 1, 2)
        p3 = F.max_pool2d(p2, 5, 1, 2)
        return self.conv("model.9.cv2", torch.cat([a, p1, p2, p3], 1))

    @torch.no_grad()
    def forward(self, in_chw):
        """in: [3, N, N] -> head [anchors, 64 + nc + nk]"""
        x = torch.as_tensor(in_chw)[None]
        t = self.taps = {}
        t["0"] = a0 = self.conv("model.0.conv", x)
        t["1"] = a1 = self.conv("model.1.conv", a0)
        if self.backbone == 1:      # ShuffleNetV2 stages; P3, P4, P5 = blocks 3, 6, 8
            t["2"] = a2 = self.shuffle_down("model.2", a1)
            t["3"] = a4 = self.shuffle_unit("model.3", a2)
            t["4"] = a3 = self.shuffle_down("model.4", a4)
            t["5"] = a5 = self.shuffle_unit("model.5", a3)
            t["6"] = a6 = self.shuffle_unit("model.6", a5)
            t["7"] = a7 = self.shuffle_down("model.7", a6)
            t["8"] = a8 = self.shuffle_unit("model.8", a7)
        else:
            t["2"] = a2 = self.c2f("model.2", a1, 1, True)
            t["3"] = a3 = self.conv("model.3.conv", a2)
            t["4"] = a4 = self.c2f("model.4", a3, 2, True)
            t["5"] = a5 = self.conv("model.5.conv", a4)
            t["6"] = a6 = self.c2f("model.6", a5, 2, True)
            t["7"] = a7 = self.conv("model.7.conv", a6)
            t["8"] = a8 = self.c2f("model.8", a7, 1, True)
        t["9"] = a9 = self.sppf(a8)
        up = lambda z: F.interpolate(z, scale_factor=2, mode="nearest")
        t["12"] = a12 = self.c2f("model.12", torch.cat([up(a9), a6], 1), 1, False)
        t["15"] = a15 = self.c2f("model.15", torch.cat([up(a12), a4], 1), 1, False)
        t["16"] = a16 = self.conv("model.16.conv", a15)
        t["18"] = a18 = self.c2f("model.18", torch.cat([a16, a12], 1), 1, False)
        t["19"] = a19 = self.conv("model.19.conv", a18)
        t["21"] = a21 = self.c2f("model.21", torch.cat([a19, a9], 1), 1, False)
        outs = []
        for i, p in enumerate((a15, a18, a21)):
            br = []
            for b in ("cv2", "cv3") + (("cv4",) if self.nk > 0 else ()):
                h = self.conv(f"model.22.{b}.{i}.0", p)
                h = self.conv(f"model.22.{b}.{i}.1", h)
                br.append(self.conv(f"model.22.{b}.{i}.2", h))
            o = torch.cat(br, 1)[0]                    # [no, H, W]
            outs.append(o.permute(1, 2, 0).reshape(-1, o.shape[0]))
        return torch.cat(outs, 0)
