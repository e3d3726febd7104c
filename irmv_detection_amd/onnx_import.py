"""ONNX -> .irmw converter.

The reference's on-disk model is a TensorRT engine built from `models/yolov7.onnx`
with trtexec (reference src/yolo_engine.cpp:28-40, README.md).  This build's
engine loads `<stem>.irmw` instead; this module produces it from the same ONNX
file.  No `onnx` package exists in the image, so the few protobuf fields needed
are decoded by hand (ModelProto.graph = 7; GraphProto.initializer = 5;
TensorProto: dims = 1, data_type = 2, float_data = 4, name = 8, raw_data = 9).

Expected input: an Ultralytics YOLOv8n / YOLOv8n-pose export (BatchNorm already
folded into the convs, so every conv initializer pair is `<layer>.weight`,
`<layer>.bias`, OIHW fp32 or fp16).  Layer names must be the Ultralytics ones
listed by `arch.conv_specs` (`model.0.conv`, `model.2.m.0.cv1.conv`, ...,
`model.22.cv2.0.2`); the fixed DFL conv (`model.22.dfl.conv`) is ignored.

    python -m irmv_detection_amd.onnx_import models/yolov7.onnx      # writes models/yolov7.irmw
"""
from __future__ import annotations

import struct
import sys
from typing import Dict, Tuple

import numpy as np

from . import arch, weights


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field number, wire type, value) of one protobuf message."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, val


_DTYPES = {1: np.float32, 10: np.float16, 11: np.float64}


def _tensor(buf: bytes):
    dims, dtype, name, raw, floats = [], 1, "", None, []
    for fno, wt, val in _fields(buf):
        if fno == 1:
            if wt == 0:
                dims.append(val)
            else:                                    # packed repeated int64
                p = 0
                while p < len(val):
                    d, p = _varint(val, p)
                    dims.append(d)
        elif fno == 2:
            dtype = val
        elif fno == 8:
            name = val.decode()
        elif fno == 9:
            raw = val
        elif fno == 4:
            floats.append(val)
    if dtype not in _DTYPES:
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, _DTYPES[dtype])
    elif floats:
        arr = np.concatenate([np.frombuffer(f, np.float32) for f in floats])
    else:
        return name, None
    return name, arr.reshape(dims) if dims else arr


def read_initializers(onnx_bytes: bytes) -> Dict[str, np.ndarray]:
    graph = None
    for fno, wt, val in _fields(onnx_bytes):
        if fno == 7 and wt == 2:
            graph = val
    if graph is None:
        raise ValueError("no GraphProto in the ONNX file")
    out = {}
    for fno, wt, val in _fields(graph):
        if fno == 5 and wt == 2:
            name, arr = _tensor(val)
            if arr is not None:
                out[name] = arr
    return out


def _lookup(init: Dict[str, np.ndarray], layer: str):
    """Ultralytics names: Conv modules keep their conv under `.conv`; the head's plain Conv2d do not."""
    for stem in (layer, layer + ".conv", layer.replace(".conv", "") + ".conv"):
        w, b = init.get(stem + ".weight"), init.get(stem + ".bias")
        if w is not None and b is not None:
            return w, b
    return None, None


def convert(onnx_bytes: bytes) -> bytes:
    init = read_initializers(onnx_bytes)
    # head sizes from the final convs of level 0
    wc, _ = _lookup(init, "model.22.cv3.0.2")
    if wc is None:
        raise ValueError("model.22.cv3.0.2 not found: not an Ultralytics YOLOv8 detect/pose export")
    nc = int(wc.shape[0])
    wk, _ = _lookup(init, "model.22.cv4.0.2")
    nk = int(wk.shape[0]) if wk is not None else 0
    if nk not in (0, 8):
        raise ValueError(f"keypoint head has {nk} outputs; this engine supports 4 keypoints x (x, y) = 8")
    # the ShuffleNetV2-backbone variant (arch.BACKBONE_SHUFFLE) is recognised by its first depthwise conv; its layer names
    # are this build's (the external repository the reference's README points to is not available offline)
    backbone = arch.BACKBONE_SHUFFLE if _lookup(init, "model.2.b1.dw")[0] is not None else arch.BACKBONE_C2F
    specs = arch.conv_specs(nc, nk, backbone)
    tensors = []
    for sp in specs:
        w, b = _lookup(init, sp.name)
        if w is None:
            raise ValueError(f"initializers of {sp.name} not found")
        if tuple(w.shape) != (sp.cout, sp.cin, sp.k, sp.k) or b.shape != (sp.cout,):
            raise ValueError(f"{sp.name}: shape {tuple(w.shape)} does not match the layer table ({sp.cout}, {sp.cin}, {sp.k}, {sp.k})")
        tensors.append((np.ascontiguousarray(w.astype(np.float32).transpose(0, 2, 3, 1)).astype(np.float16),   # OIHW -> OHWI
                        np.ascontiguousarray(b.astype(np.float32))))
    return weights.build_blob(specs, tensors, nc, nk, backbone)


# ---- minimal writer, used by the tests to make an ONNX file to import -------------
def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _enc_field(fno: int, payload: bytes) -> bytes:
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def write_initializer_only_onnx(tensors: Dict[str, np.ndarray]) -> bytes:
    """An ONNX ModelProto that carries nothing but named fp32 initializers."""
    graph = b""
    for name, arr in tensors.items():
        arr = np.ascontiguousarray(arr, np.float32)
        t = b"".join(_enc_varint((1 << 3) | 0) + _enc_varint(int(d)) for d in arr.shape)
        t += _enc_varint((2 << 3) | 0) + _enc_varint(1)
        t += _enc_field(8, name.encode()) + _enc_field(9, arr.tobytes())
        graph += _enc_field(5, t)
    return _enc_varint((1 << 3) | 0) + _enc_varint(8) + _enc_field(7, graph)


def main(argv):
    args = [a for a in argv[1:] if a != "--int8"]
    if len(args) != 1:
        print(__doc__)
        return 2
    src = args[0]
    with open(src, "rb") as f:
        blob = convert(f.read())
    if "--int8" in argv:      # per-output-channel int8 weights (BASELINE configs[4]); biases stay fp32
        blob = weights.quantize_blob_int8(blob)
    dst = weights.model_blob_path(src)
    with open(dst, "wb") as f:
        f.write(blob)
    print(f"wrote {dst} ({len(blob)} bytes)")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
