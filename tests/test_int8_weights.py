"""Row f4 / BASELINE configs[4]: per-output-channel int8 weight blobs (.irmw dtype 2).

The reference lists INT8 as not done (README.md:30) and ships no model, so there is nothing of the reference's to pin
these against: parity unpinned.  What is checked: the quantiser's error bound, the container round trip, that the oracle
computes with exactly w = fp16(q * scale), and (GPU) that the engine does too -- bit for bit the engine fed the
dequantised fp16 blob -- at 640 and at configs[4]'s 416 x 416 input."""
import numpy as np
import pytest

from irmv_detection_amd import frames, onnx_import, weights
from oracle import oracle


@pytest.fixture(scope="module")
def int8_blob(blob):
    return weights.quantize_blob_int8(blob)


def _dequantised_fp16_blob(int8_blob):
    hdr, layers = weights.parse_blob(int8_blob)
    return weights.build_blob([sp for sp, _, _ in layers], [(w, b) for _, w, b in layers], hdr["nc"], hdr["nk"])


def test_quantiser_error_bound_and_range():
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((48, 3, 3, 32)) * rng.uniform(0.01, 2.0, (48, 1, 1, 1))).astype(np.float16)
    w[7] = 0
    q, s = weights.quantize_int8(w)
    assert q.dtype == np.int8 and q.min() >= -127 and q.max() <= 127 and s.shape == (48,) and (s > 0).all()
    err = np.abs(w.astype(np.float32) - q.astype(np.float32) * s[:, None, None, None])
    assert (err <= 0.5 * s[:, None, None, None] * (1 + 1e-6)).all()
    assert (np.abs(q).reshape(48, -1).max(1)[np.arange(48) != 7] == 127).all()      # every channel uses the full range
    assert not q[7].any()


def test_int8_blob_container_round_trip(blob, int8_blob):
    assert len(int8_blob) < 0.56 * len(blob)
    h16, l16 = weights.parse_blob(blob)
    h8, l8 = weights.parse_blob(int8_blob)
    assert h8["dtype"] == weights.DTYPE_INT8 and (h8["nc"], h8["nk"], h8["n_layers"]) == (h16["nc"], h16["nk"], h16["n_layers"])
    for (sp, w, b), (sp8, w8, b8) in zip(l16, l8):
        assert sp == sp8 and np.array_equal(b, b8)
        q, s = weights.quantize_int8(w)
        assert np.array_equal(w8, (q.astype(np.float32) * s[:, None, None, None]).astype(np.float16))
        assert np.abs(w.astype(np.float32) - w8.astype(np.float32)).max() <= 0.5 * s.max() + 1e-3
    with pytest.raises(ValueError):
        weights.parse_blob(int8_blob[:24] + b"\x07\x00\x00\x00" + int8_blob[28:])     # unknown dtype


def test_oracle_computes_with_the_dequantised_fp16_weights(int8_blob, frame0):
    x = oracle.preprocess(frame0, 640)
    h8 = oracle.Net(int8_blob).forward(x)
    h16 = oracle.Net(_dequantised_fp16_blob(int8_blob)).forward(x)
    assert np.array_equal(h8, h16)


def test_importer_int8_flag(tmp_path, blob, int8_blob):
    from test_onnx_import import _as_onnx
    specs, tensors = weights.synthetic_tensors(0)
    p = tmp_path / "m.onnx"
    p.write_bytes(_as_onnx(specs, tensors))
    assert onnx_import.main(["prog", "--int8", str(p)]) == 0
    assert (tmp_path / "m.irmw").read_bytes() == int8_blob


@pytest.mark.gpu
@pytest.mark.parametrize("net", [640, 416])
def test_engine_int8_blob_equals_dequantised_fp16_blob(int8_blob, net):
    from irmv_detection_amd.engine import YoloEngine
    f = frames.synthetic_frame(5)
    heads = []
    for b in (int8_blob, _dequantised_fp16_blob(int8_blob)):
        with YoloEngine(None, (1280, 1024), weights_blob=b, net_size=net) as e:
            e.get_src_image_buffer()[:] = f
            e.detect()
            heads.append((e.read_head(0).copy(), e.read_raw(0)))
    assert np.array_equal(heads[0][0], heads[1][0])
    assert heads[0][1]["num_dets"] == heads[1][1]["num_dets"] and np.array_equal(heads[0][1]["boxes"], heads[1][1]["boxes"])
    ho = oracle.Net(int8_blob).forward(oracle.preprocess(f, net))
    assert heads[0][0].shape == ho.shape == ((net // 8) ** 2 + (net // 16) ** 2 + (net // 32) ** 2, 86)
    assert np.abs(heads[0][0] - ho).max() <= 4e-2      # HEAD_TOL of tests/test_gpu_engine.py
    d = oracle.decode_nms(heads[0][0], net, 14, 8)
    assert d["num_dets"] == heads[0][1]["num_dets"] and np.array_equal(d["boxes"], heads[0][1]["boxes"])


@pytest.mark.gpu
def test_imported_model_reproduces_the_source_head(tmp_path, blob, frame0):
    """onnx_import -> <stem>.irmw -> engine == the engine fed the source blob (the model-file convention of
    src/yolo_engine.cpp:28-40 end to end)."""
    from irmv_detection_amd.engine import YoloEngine
    from test_onnx_import import _as_onnx
    specs, tensors = weights.synthetic_tensors(0)
    p = tmp_path / "yolov7.onnx"
    p.write_bytes(_as_onnx(specs, tensors))
    assert onnx_import.main(["prog", str(p)]) == 0
    heads = []
    for kw in (dict(onnx_file_path=str(p)), dict(onnx_file_path=None, weights_blob=blob)):
        path = kw.pop("onnx_file_path")
        with YoloEngine(path, (1280, 1024), **kw) as e:
            e.get_src_image_buffer()[:] = frame0
            e.detect()
            heads.append(e.read_head(0).copy())
    assert np.array_equal(heads[0], heads[1])
