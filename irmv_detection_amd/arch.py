"""YOLOv8n conv inventory for the armor-detection hot path (and its ShuffleNetV2-backbone variant).

The reference runs an opaque serialized TensorRT engine (reference
src/yolo_engine.cpp:28-36, :105); its README names the network as YOLOv8n
(README.md:9-16).  The architecture below is the published Ultralytics YOLOv8n
graph (scale n: depth 0.33, width 0.25) as written down in SURVEY.md Appendix A,
with nc = 14 armor classes (reference include/irmv_detection/armor.hpp:7) and an
optional YOLOv8-pose style 4-keypoint branch (Appendix A.4).

The reference's README also benchmarks "YOLOv8n (Shufflenet backbone)" (README.md:12,16, an external repository that
is not available offline; BASELINE configs[4]).  `BACKBONE_SHUFFLE` is this build's stand-in for it: the YOLOv8n stem,
neck and Detect head around ShuffleNetV2 stages (Ma et al., ECCV 2018: channel split, 1x1 -> depthwise 3x3 -> 1x1,
concat, channel shuffle with 2 groups) of the same widths as the C2f stages they replace (64 / 128 / 256 channels at
strides 8 / 16 / 32), so P3 / P4 / P5 keep their shapes.  Architecture [external], parity unpinned.

This module is only a *table of conv layers* in canonical order.  It is the
contract between the weight-blob writer (weights.py), the HIP engine
(csrc/engine.cpp, which re-derives the same table and refuses a blob that does
not match it) and the test-side restatements.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

NUM_CLASSES = 14          # B1..B5,BO,BS,R1..R5,RO,RS (armor.hpp:7)
NUM_KPT_CH = 8            # 4 keypoints x (x, y)
REG_MAX = 16
NET_SIZE = 640
STRIDES = (8, 16, 32)

ARMOR_CLASS_NAMES = (
    "B1", "B2", "B3", "B4", "B5", "BO", "BS",
    "R1", "R2", "R3", "R4", "R5", "RO", "RS", "UNKNOWN",
)

ACT_NONE = 0
ACT_SILU = 1


@dataclass(frozen=True)
class ConvSpec:
    name: str
    cin: int
    cout: int
    k: int
    stride: int
    act: int
    groups: int = 1       # > 1: depthwise (groups == cout, cin == 1 = input channels per group)

    @property
    def n_weights(self) -> int:
        return self.cout * self.cin * self.k * self.k

    @property
    def n_params(self) -> int:
        return self.n_weights + self.cout


def _c2f(prefix: str, c1: int, c2: int, n: int) -> List[ConvSpec]:
    c = c2 // 2
    out = [ConvSpec(f"{prefix}.cv1", c1, 2 * c, 1, 1, ACT_SILU)]
    for i in range(n):
        out.append(ConvSpec(f"{prefix}.m.{i}.cv1", c, c, 3, 1, ACT_SILU))
        out.append(ConvSpec(f"{prefix}.m.{i}.cv2", c, c, 3, 1, ACT_SILU))
    out.append(ConvSpec(f"{prefix}.cv2", (2 + n) * c, c2, 1, 1, ACT_SILU))
    return out


BACKBONE_C2F, BACKBONE_SHUFFLE = 0, 1
# ShuffleNetV2 stages of BACKBONE_SHUFFLE: (first block index, output channels, stride-1 units after the stride-2 block)
SHUFFLE_STAGES = ((2, 64, 1), (4, 128, 2), (7, 256, 1))


def _shuffle_down(prefix: str, c1: int, c2: int) -> List[ConvSpec]:
    """Stride-2 ShuffleNetV2 block: both branches see the whole input; out = shuffle(cat(b1, b2))."""
    bc = c2 // 2
    return [ConvSpec(f"{prefix}.b1.dw", 1, c1, 3, 2, ACT_NONE, c1), ConvSpec(f"{prefix}.b1.pw", c1, bc, 1, 1, ACT_SILU),
            ConvSpec(f"{prefix}.b2.pw1", c1, bc, 1, 1, ACT_SILU), ConvSpec(f"{prefix}.b2.dw", 1, bc, 3, 2, ACT_NONE, bc),
            ConvSpec(f"{prefix}.b2.pw2", bc, bc, 1, 1, ACT_SILU)]


def _shuffle_unit(prefix: str, c: int) -> List[ConvSpec]:
    """Stride-1 unit: x1, x2 = split(x); out = shuffle(cat(x1, b2(x2)))."""
    bc = c // 2
    return [ConvSpec(f"{prefix}.b2.pw1", bc, bc, 1, 1, ACT_SILU), ConvSpec(f"{prefix}.b2.dw", 1, bc, 3, 1, ACT_NONE, bc),
            ConvSpec(f"{prefix}.b2.pw2", bc, bc, 1, 1, ACT_SILU)]


def conv_specs(nc: int = NUM_CLASSES, nk: int = NUM_KPT_CH, backbone: int = BACKBONE_C2F) -> List[ConvSpec]:
    """All conv layers (BN folded) in canonical order: backbone, neck, then the
    Detect head branches box (cv2), cls (cv3), kpt (cv4; only when nk > 0)."""
    L: List[ConvSpec] = []
    L.append(ConvSpec("model.0.conv", 3, 16, 3, 2, ACT_SILU))
    L.append(ConvSpec("model.1.conv", 16, 32, 3, 2, ACT_SILU))
    if backbone == BACKBONE_SHUFFLE:
        c_prev = 32
        for first, c, units in SHUFFLE_STAGES:
            L += _shuffle_down(f"model.{first}", c_prev, c)
            for u in range(units):
                L += _shuffle_unit(f"model.{first + 1 + u}", c)
            c_prev = c
    else:
        L += _c2f("model.2", 32, 32, 1)
        L.append(ConvSpec("model.3.conv", 32, 64, 3, 2, ACT_SILU))
        L += _c2f("model.4", 64, 64, 2)
        L.append(ConvSpec("model.5.conv", 64, 128, 3, 2, ACT_SILU))
        L += _c2f("model.6", 128, 128, 2)
        L.append(ConvSpec("model.7.conv", 128, 256, 3, 2, ACT_SILU))
        L += _c2f("model.8", 256, 256, 1)
    L.append(ConvSpec("model.9.cv1", 256, 128, 1, 1, ACT_SILU))
    L.append(ConvSpec("model.9.cv2", 512, 256, 1, 1, ACT_SILU))
    L += _c2f("model.12", 384, 128, 1)
    L += _c2f("model.15", 192, 64, 1)
    L.append(ConvSpec("model.16.conv", 64, 64, 3, 2, ACT_SILU))
    L += _c2f("model.18", 192, 128, 1)
    L.append(ConvSpec("model.19.conv", 128, 128, 3, 2, ACT_SILU))
    L += _c2f("model.21", 384, 256, 1)
    ch = (64, 128, 256)
    c2 = max(16, ch[0] // 4, 4 * REG_MAX)          # 64
    c3 = max(ch[0], min(nc, 100))                  # 64
    for i, c in enumerate(ch):
        L.append(ConvSpec(f"model.22.cv2.{i}.0", c, c2, 3, 1, ACT_SILU))
        L.append(ConvSpec(f"model.22.cv2.{i}.1", c2, c2, 3, 1, ACT_SILU))
        L.append(ConvSpec(f"model.22.cv2.{i}.2", c2, 4 * REG_MAX, 1, 1, ACT_NONE))
    for i, c in enumerate(ch):
        L.append(ConvSpec(f"model.22.cv3.{i}.0", c, c3, 3, 1, ACT_SILU))
        L.append(ConvSpec(f"model.22.cv3.{i}.1", c3, c3, 3, 1, ACT_SILU))
        L.append(ConvSpec(f"model.22.cv3.{i}.2", c3, nc, 1, 1, ACT_NONE))
    if nk > 0:
        c4 = max(ch[0] // 4, nk)                   # 16
        for i, c in enumerate(ch):
            L.append(ConvSpec(f"model.22.cv4.{i}.0", c, c4, 3, 1, ACT_SILU))
            L.append(ConvSpec(f"model.22.cv4.{i}.1", c4, c4, 3, 1, ACT_SILU))
            L.append(ConvSpec(f"model.22.cv4.{i}.2", c4, nk, 1, 1, ACT_NONE))
    return L


def level_shapes(net: int = NET_SIZE):
    """[(H, W, stride)] of the three Detect levels."""
    return [(net // s, net // s, s) for s in STRIDES]


def num_anchors(net: int = NET_SIZE) -> int:
    return sum(h * w for h, w, _ in level_shapes(net))


def conv_out_hw(net: int = NET_SIZE, backbone: int = BACKBONE_C2F):
    """name -> output spatial size, used for FLOP accounting."""
    hw = {}
    s2, s4, s8, s16, s32 = net // 2, net // 4, net // 8, net // 16, net // 32
    for sp in conv_specs(backbone=backbone):
        n = sp.name
        idx = int(n.split(".")[1])
        if backbone == BACKBONE_SHUFFLE and 2 <= idx <= 8:
            out = s8 if idx <= 3 else (s16 if idx <= 6 else s32)
            down = idx in (2, 4, 7)
            hw[n] = out * 2 if (down and n.endswith("b2.pw1")) else out   # a stride-2 block's first 1x1 runs at the input size
        elif idx == 0:
            hw[n] = s2
        elif idx in (1, 2):
            hw[n] = s4
        elif idx in (3, 4, 15):
            hw[n] = s8
        elif idx in (5, 6, 12, 16, 18):
            hw[n] = s16
        elif idx in (7, 8, 9, 19, 21):
            hw[n] = s32
        elif idx == 22:
            lvl = int(n.split(".")[3])
            hw[n] = (s8, s16, s32)[lvl]
    return hw


def flops_per_frame(net: int = NET_SIZE, nc: int = NUM_CLASSES, nk: int = NUM_KPT_CH, backbone: int = BACKBONE_C2F) -> int:
    """2 * MACs over every conv of one frame (8.096 GFLOP at
    640, nc = 14, no kpt head; 8.343 GFLOP with the 4-kpt head)."""
    hw = conv_out_hw(net, backbone)
    tot = 0
    for sp in conv_specs(nc, nk, backbone):
        tot += 2 * hw[sp.name] ** 2 * sp.cout * sp.cin * sp.k * sp.k
    return tot
