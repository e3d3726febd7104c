"""ONNX -> .irmw converter (row f4): hand-rolled protobuf reader round trip."""
import numpy as np
import pytest

from irmv_detection_amd import arch, onnx_import, weights


def _as_onnx(specs, tensors, conv_suffix=True):
    init = {}
    for sp, (w, b) in zip(specs, tensors):
        is_plain = sp.name.startswith("model.22.") and sp.name.endswith(".2")
        stem = sp.name if (is_plain or sp.name.endswith(".conv") or not conv_suffix) else sp.name + ".conv"
        init[stem + ".weight"] = w.astype(np.float32).transpose(0, 3, 1, 2)       # OHWI -> OIHW
        init[stem + ".bias"] = b
    init["model.22.dfl.conv.weight"] = np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1)   # ignored
    return onnx_import.write_initializer_only_onnx(init)


def test_roundtrip_reproduces_the_blob(blob):
    specs, tensors = weights.synthetic_tensors(0)
    out = onnx_import.convert(_as_onnx(specs, tensors))
    assert out == blob                                   # fp16 weights survive fp32 ONNX exactly
    out2 = onnx_import.convert(_as_onnx(specs, tensors, conv_suffix=False))
    assert out2 == blob


def test_bbox_only_model_and_errors():
    specs, tensors = weights.synthetic_tensors(0, nk=0)
    out = onnx_import.convert(_as_onnx(specs, tensors))
    hdr, layers = weights.parse_blob(out)
    assert hdr["nk"] == 0 and hdr["nc"] == 14 and len(layers) == 63
    with pytest.raises(ValueError):
        onnx_import.convert(b"\x08\x08")                 # no graph
    bad = dict(zip((s.name for s in specs), tensors))
    init = {"model.22.cv3.0.2.weight": np.zeros((14, 64, 1, 1), np.float32), "model.22.cv3.0.2.bias": np.zeros(14, np.float32)}
    with pytest.raises(ValueError, match="model.0.conv"):
        onnx_import.convert(onnx_import.write_initializer_only_onnx(init))


def test_cli_writes_sibling_irmw(tmp_path, blob):
    specs, tensors = weights.synthetic_tensors(0)
    p = tmp_path / "yolov7.onnx"
    p.write_bytes(_as_onnx(specs, tensors))
    assert onnx_import.main(["prog", str(p)]) == 0
    assert (tmp_path / "yolov7.irmw").read_bytes() == blob


def test_shuffle_backbone_is_recognised_and_round_trips():
    from irmv_detection_amd import arch
    specs, tensors = weights.synthetic_tensors(0, backbone=arch.BACKBONE_SHUFFLE)
    out = onnx_import.convert(_as_onnx(specs, tensors))
    assert out == weights.synthetic_blob(0, backbone=arch.BACKBONE_SHUFFLE)
    assert weights.parse_blob(out)[0]["backbone"] == arch.BACKBONE_SHUFFLE
