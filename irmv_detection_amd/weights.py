"""`.irmw` weight blob: format, reader/writer and the seeded synthetic generator.

The reference loads `<model>.engine` next to the `.onnx` path it is given
(reference src/yolo_engine.cpp:28-40, :137-151) -- an opaque TensorRT plan that
is not in the repository.  This build loads `<stem>.irmw` from the same place:
a flat little-endian blob of BN-folded conv weights.

Layout (all offsets from the start of the blob, data 16-byte aligned):

    header  32 B : magic "IRMW", u32 version(1), nc, nk, reg_max, n_layers,
                   dtype (1 = fp16 weights / fp32 bias, 2 = int8 weights + per-channel scales),
                   backbone (0 = YOLOv8n C2f stages, 1 = ShuffleNetV2 stages: arch.py)
    table   n_layers x 72 B : char name[32]; u32 cin, cout, k, stride, act, groups (0 = 1;
                   depthwise: groups == cout, cin == 1); u64 w_off, b_off
    data    per layer: weights fp16 in OHWI order [cout][kh][kw][cin], then
                   bias fp32 [cout]

No trained weights exist offline, so `synthetic_blob`
produces seeded weights whose per-layer gains keep fp16 activations in range and
whose head biases make a few hundred (anchor, class) pairs pass the score
threshold with overlapping boxes, so NMS and PnP parity are not vacuous.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import arch

MAGIC = b"IRMW"
VERSION = 1
HEADER_FMT = "<4s7I"
HEADER_SIZE = struct.calcsize(HEADER_FMT)      # 32
LAYER_FMT = "<32s6I2Q"
LAYER_SIZE = struct.calcsize(LAYER_FMT)        # 72

_CALIB_PATH = os.path.join(os.path.dirname(__file__), "data", "synth_calib.json")
_CALIB_PATH_SHUFFLE = os.path.join(os.path.dirname(__file__), "data", "synth_calib_shuffle.json")


def _align(n: int, a: int = 16) -> int:
    return (n + a - 1) // a * a


def build_blob(specs: List[arch.ConvSpec], tensors: List[Tuple[np.ndarray, np.ndarray]],
               nc: int, nk: int, backbone: int = arch.BACKBONE_C2F) -> bytes:
    assert len(specs) == len(tensors)
    off = _align(HEADER_SIZE + LAYER_SIZE * len(specs))
    table = []
    chunks = []
    for sp, (w, b) in zip(specs, tensors):
        assert w.dtype == np.float16 and w.shape == (sp.cout, sp.k, sp.k, sp.cin), (sp, w.shape)
        assert b.dtype == np.float32 and b.shape == (sp.cout,)
        w_off = off
        off = _align(off + w.nbytes)
        b_off = off
        off = _align(off + b.nbytes)
        table.append(struct.pack(LAYER_FMT, sp.name.encode(), sp.cin, sp.cout, sp.k, sp.stride,
                                 sp.act, sp.groups if sp.groups > 1 else 0, w_off, b_off))
        chunks.append((w_off, w.tobytes()))
        chunks.append((b_off, b.tobytes()))
    buf = bytearray(off)
    buf[:HEADER_SIZE] = struct.pack(HEADER_FMT, MAGIC, VERSION, nc, nk, arch.REG_MAX,
                                    len(specs), 1, backbone)
    p = HEADER_SIZE
    for t in table:
        buf[p:p + LAYER_SIZE] = t
        p += LAYER_SIZE
    for o, data in chunks:
        buf[o:o + len(data)] = data
    return bytes(buf)


DTYPE_FP16, DTYPE_INT8 = 1, 2


def parse_blob(blob: bytes):
    """-> (header dict, [(ConvSpec, w fp16 OHWI, b fp32)]).  An int8 blob (dtype 2) is returned DEQUANTISED: the fp16
    weights every consumer (engine, oracle) computes with, w = fp16(q * scale)."""
    magic, ver, nc, nk, reg_max, n_layers, dtype, backbone = struct.unpack_from(HEADER_FMT, blob, 0)
    if magic != MAGIC or ver != VERSION or dtype not in (DTYPE_FP16, DTYPE_INT8):
        raise ValueError("not an IRMW v1 blob (fp16 or int8 weights)")
    out = []
    for i in range(n_layers):
        name, cin, cout, k, stride, act, groups, w_off, b_off = struct.unpack_from(
            LAYER_FMT, blob, HEADER_SIZE + i * LAYER_SIZE)
        sp = arch.ConvSpec(name.rstrip(b"\0").decode(), cin, cout, k, stride, act, max(1, groups))
        if dtype == DTYPE_FP16:
            w = np.frombuffer(blob, np.float16, sp.n_weights, w_off).reshape(cout, k, k, cin)
        else:
            q = np.frombuffer(blob, np.int8, sp.n_weights, w_off).reshape(cout, k, k, cin)
            scale = np.frombuffer(blob, np.float32, cout, w_off + _align(sp.n_weights, 4))
            w = (q.astype(np.float32) * scale[:, None, None, None]).astype(np.float16)
        b = np.frombuffer(blob, np.float32, cout, b_off)
        out.append((sp, w, b))
    return dict(nc=nc, nk=nk, reg_max=reg_max, n_layers=n_layers, dtype=dtype, backbone=backbone), out


def quantize_int8(w: np.ndarray):
    """Per-output-channel symmetric int8: scale[o] = max|w[o]| / 127, q = round(w / scale) in [-127, 127].
    -> (q int8 OHWI, scale fp32 [cout]).  |w - q * scale| <= scale / 2 element-wise."""
    w32 = np.asarray(w, np.float32)
    amax = np.abs(w32).reshape(w32.shape[0], -1).max(axis=1)
    scale = np.where(amax > 0, amax / 127.0, 1.0).astype(np.float32)
    q = np.clip(np.rint(w32 / scale[:, None, None, None]), -127, 127).astype(np.int8)
    return q, scale


def build_blob_int8(specs: List[arch.ConvSpec], tensors: List[Tuple[np.ndarray, np.ndarray]], nc: int, nk: int,
                    backbone: int = arch.BACKBONE_C2F) -> bytes:
    """.irmw with dtype 2 (BASELINE configs[4]: int8 weights): per layer int8 OHWI weights followed (4-byte aligned) by
    the fp32 per-output-channel scales; biases stay fp32.  Half the bytes of the fp16 blob on disk, over the RCCL
    broadcast and in the host->device upload; the engine expands w = fp16(q * scale) once at load into the MFMA fragment
    layout, so the kernels and their fp16-multiply / fp32-accumulate arithmetic are the fp16 model's."""
    off = _align(HEADER_SIZE + LAYER_SIZE * len(specs))
    table, chunks = [], []
    for sp, (w, b) in zip(specs, tensors):
        q, scale = quantize_int8(w)
        w_off = off
        off = _align(off + _align(sp.n_weights, 4) + 4 * sp.cout)
        b_off = off
        off = _align(off + 4 * sp.cout)
        table.append(struct.pack(LAYER_FMT, sp.name.encode(), sp.cin, sp.cout, sp.k, sp.stride, sp.act, sp.groups if sp.groups > 1 else 0, w_off, b_off))
        chunks.append((w_off, q.tobytes()))
        chunks.append((w_off + _align(sp.n_weights, 4), scale.tobytes()))
        chunks.append((b_off, np.asarray(b, np.float32).tobytes()))
    buf = bytearray(off)
    buf[:HEADER_SIZE] = struct.pack(HEADER_FMT, MAGIC, VERSION, nc, nk, arch.REG_MAX, len(specs), DTYPE_INT8, backbone)
    p = HEADER_SIZE
    for t in table:
        buf[p:p + LAYER_SIZE] = t
        p += LAYER_SIZE
    for o, data in chunks:
        buf[o:o + len(data)] = data
    return bytes(buf)


def quantize_blob_int8(blob: bytes) -> bytes:
    """fp16 .irmw -> int8 .irmw (same layers, biases, head)."""
    hdr, layers = parse_blob(blob)
    return build_blob_int8([sp for sp, _, _ in layers], [(w, b) for _, w, b in layers], hdr["nc"], hdr["nk"], hdr["backbone"])


def load_calib(backbone: int = arch.BACKBONE_C2F) -> Optional[dict]:
    path = _CALIB_PATH_SHUFFLE if backbone == arch.BACKBONE_SHUFFLE else _CALIB_PATH
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


# Corner offsets (in units of the level stride) of the synthetic keypoint head's
# bias, in the order PnP consumes them: left-bottom, left-top, right-top,
# right-bottom (reference src/pnp_solver.cpp:41-44).  Image y points down.
KPT_BASE = ((-1.5, 0.6), (-1.5, -0.6), (1.5, -0.6), (1.5, 0.6))


def synthetic_tensors(seed: int = 0, nc: int = arch.NUM_CLASSES, nk: int = arch.NUM_KPT_CH,
                      calib: Optional[dict] = None, use_calib: bool = True, backbone: int = arch.BACKBONE_C2F):
    """Seeded weights.  `calib` = {"gain": {name: g}, "cls_bias": {level: b}}
    (written by tests/golden/make_calib.py); absent entries use analytic
    defaults (SiLU second moment 0.356 -> gain 1.68)."""
    if calib is None and use_calib:
        calib = load_calib(backbone)
    gains: Dict[str, float] = (calib or {}).get("gain", {})
    cls_bias: Dict[str, float] = (calib or {}).get("cls_bias", {})
    rng = np.random.Generator(np.random.PCG64(seed))
    specs = arch.conv_specs(nc, nk, backbone)
    tensors = []
    for sp in specs:
        fan_in = sp.cin * sp.k * sp.k
        default_gain = 4.0 if sp.name == "model.0.conv" else (1.0 if sp.groups > 1 else 1.68)   # depthwise: no activation after it
        g = float(gains.get(sp.name, default_gain))
        w = rng.standard_normal((sp.cout, sp.k, sp.k, sp.cin), dtype=np.float32)
        b = rng.standard_normal(sp.cout, dtype=np.float32)
        parts = sp.name.split(".")
        is_final = parts[1] == "22" and parts[4] == "2"
        if not is_final:
            w *= g / np.sqrt(fan_in)
            b *= 0.1
        else:
            branch, lvl = parts[2], parts[3]
            if branch == "cv2":      # DFL logits: decaying bias -> ~2-bin expectation
                w *= g / np.sqrt(fan_in)
                j = np.arange(sp.cout, dtype=np.float32) % arch.REG_MAX
                b = b * 0.3 - 0.4 * j
            elif branch == "cv3":    # class logits: N(bias, ~1)
                w *= g / np.sqrt(fan_in)
                b = b * 0.2 + float(cls_bias.get(lvl, -4.2))
            else:                    # keypoints: armor-like quad around the anchor
                w *= 0.15 * g / np.sqrt(fan_in)
                base = np.array(KPT_BASE, np.float32).reshape(-1)[:sp.cout]
                b = (0.25 + base / 2.0 + 0.02 * b).astype(np.float32)
        tensors.append((w.astype(np.float16), b.astype(np.float32)))
    return specs, tensors


def synthetic_blob(seed: int = 0, nc: int = arch.NUM_CLASSES, nk: int = arch.NUM_KPT_CH,
                   calib: Optional[dict] = None, use_calib: bool = True, backbone: int = arch.BACKBONE_C2F) -> bytes:
    specs, tensors = synthetic_tensors(seed, nc, nk, calib, use_calib, backbone)
    return build_blob(specs, tensors, nc, nk, backbone)


def model_blob_path(onnx_file_path: str) -> str:
    """`<dir>/<stem>.onnx` -> `<dir>/<stem>.irmw`, the counterpart of the
    reference's `.onnx` -> `.engine` rule (src/yolo_engine.cpp:28-31)."""
    return os.path.splitext(onnx_file_path)[0] + ".irmw"
