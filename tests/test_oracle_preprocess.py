"""Oracle preprocess (reference src/yolo_engine.cpp:179-200) against golden vectors
and an independent float64 numpy statement of the same bilinear rule."""
import json
import zlib

import numpy as np
import pytest

from conftest import golden_path
from oracle import oracle


def numpy_bilinear(src, net, rotate, swap):
    """Half-pixel-centre bilinear stretch in float64 (no coefficient quantisation)."""
    if rotate:
        src = src[::-1, ::-1]
    if swap:
        src = src[..., ::-1]
    sh, sw, _ = src.shape
    fy = (np.arange(net) + 0.5) * sh / net - 0.5
    fx = (np.arange(net) + 0.5) * sw / net - 0.5
    y0 = np.clip(np.floor(fy).astype(int), 0, sh - 1); y1 = np.clip(y0 + 1, 0, sh - 1)
    x0 = np.clip(np.floor(fx).astype(int), 0, sw - 1); x1 = np.clip(x0 + 1, 0, sw - 1)
    wy = np.clip(fy - np.floor(fy), 0, 1); wy[fy < 0] = 0; wy[fy >= sh - 1] = 0
    wx = np.clip(fx - np.floor(fx), 0, 1); wx[fx < 0] = 0; wx[fx >= sw - 1] = 0
    s = src.astype(np.float64)
    top = s[y0][:, x0] * (1 - wx)[None, :, None] + s[y0][:, x1] * wx[None, :, None]
    bot = s[y1][:, x0] * (1 - wx)[None, :, None] + s[y1][:, x1] * wx[None, :, None]
    return top * (1 - wy)[:, None, None] + bot * wy[:, None, None]


def test_golden_pre_cases():
    g = np.load(golden_path("pre_cases.npz"))
    for i in range(8):
        mode, rot, swap = (int(v) for v in g[f"cfg{i}"])
        x, u8 = oracle.preprocess(g[f"src{i}"], 32, mode, bool(rot), bool(swap), want_u8=True)
        assert np.array_equal(u8, g[f"out{i}"]), i
        # K3 + K4 of the reference chain: v/255, HWC -> CHW
        assert np.array_equal(x, (u8.astype(np.float32) / np.float32(255)).transpose(2, 0, 1))


@pytest.mark.parametrize("rotate,swap", [(True, False), (False, False), (True, True)])
def test_matches_float_bilinear_within_one_lsb(rotate, swap):
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (103, 161, 3), dtype=np.uint8)   # 1.61x / 1.6x like 1280x1024 -> 640
    _, u8 = oracle.preprocess(src, 64, 0, rotate, swap, want_u8=True)
    ref = numpy_bilinear(src, 64, rotate, swap)
    # 11-bit coefficients: the fixed-point result is the rounded float result +-1 LSB
    assert np.abs(u8.astype(np.float64) - ref).max() <= 1.0
    assert (u8 == np.floor(ref + 0.5)).mean() > 0.97


def test_identity_scale_is_exact_copy():
    rng = np.random.default_rng(4)
    src = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    _, u8 = oracle.preprocess(src, 64, 0, False, False, want_u8=True)
    assert np.array_equal(u8, src)
    _, u8r = oracle.preprocess(src, 64, 0, True, False, want_u8=True)
    assert np.array_equal(u8r, src[::-1, ::-1])
    assert np.array_equal(oracle.rotate180(src), src[::-1, ::-1])


def test_rotation_commutes_with_resize():
    # nppiMirror then nppiResize == resize of the mirrored frame (size-independent property)
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)
    a = oracle.preprocess(src, 64, 0, True, False)
    b = oracle.preprocess(np.ascontiguousarray(src[::-1, ::-1]), 64, 0, False, False)
    assert np.array_equal(a, b)


def test_letterbox_geometry():
    src = np.full((1024, 1280, 3), 200, np.uint8)
    _, u8 = oracle.preprocess(src, 640, oracle.RESIZE_LETTERBOX, True, False, want_u8=True)
    assert (u8[:64] == 114).all() and (u8[576:] == 114).all() and (u8[64:576] == 200).all()


def test_rm_test_jpg_golden(rm_test_image):
    meta = json.load(open(golden_path("rm_test_pre.json")))
    assert list(rm_test_image.shape) == meta["src_shape"] == [1024, 1280, 3]
    if zlib.crc32(rm_test_image.tobytes()) != meta["src_crc32"]:
        pytest.skip("JPEG decoder differs from the one that produced the golden CRC")
    x, u8 = oracle.preprocess(rm_test_image, 640, 0, True, False, want_u8=True)
    assert zlib.crc32(u8.tobytes()) == meta["u8_crc32"]
    assert zlib.crc32(x.astype(np.float16).tobytes()) == meta["fp16_chw_crc32"]


def test_u8_to_unit_fp16_by_reciprocal_is_exact():
    """The HIP kernels compute (half)(q * (1/255)) instead of the oracle's (half)(q / 255.0f): identical for all 256 inputs."""
    q = np.arange(256, dtype=np.float32)
    assert np.array_equal((q / np.float32(255.0)).astype(np.float16),
                          (q * (np.float32(1.0) / np.float32(255.0))).astype(np.float16))
