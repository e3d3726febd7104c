"""Oracle network (orc_net.c) against torch-CPU and against the committed block goldens."""
import numpy as np
import torch

from conftest import golden_path
from torch_ref import TorchNet


def test_blocks_match_golden_and_torch(blob, onet):
    g = np.load(golden_path("net_blocks.npz"))
    x = g["x"].astype(np.float32)
    tn = TorchNet(blob)
    head_t = tn.forward(torch.from_numpy(x)).numpy()
    head = onet.forward(x)
    assert np.array_equal(head, g["head"])                       # oracle is pinned bit for bit
    # stated tolerance: CPU fp32 vs torch-CPU <= 1e-4 relative to the tensor's scale
    assert np.abs(head - head_t).max() <= 1e-4 * max(1.0, np.abs(head_t).max())
    for tap in ("0", "2", "4", "9", "15", "21"):
        _, t = onet.forward(x, tap=tap)
        assert np.array_equal(t, g["tap_" + tap]), tap
        tt = tn.taps[tap][0].permute(1, 2, 0).numpy()
        assert np.abs(t - tt).max() <= 1e-4 * max(1.0, np.abs(tt).max()), tap


def test_single_conv_layers_vs_torch(blob, onet):
    tn = TorchNet(blob)
    rng = np.random.default_rng(0)
    for name, H in (("model.1.conv", 32), ("model.2.m.0.cv1", 16), ("model.9.cv2", 8), ("model.22.cv3.1.2", 8),
                    ("model.22.cv4.0.1", 16), ("model.7.conv", 16)):
        sp, w, b = tn.p[name]
        x = rng.standard_normal((H, H, sp.cin)).astype(np.float32)
        y = onet.conv_layer(name, x, sp.cout, sp.stride)
        yt = tn.conv(name, torch.from_numpy(x).permute(2, 0, 1)[None])[0].permute(1, 2, 0).numpy()
        assert y.shape == yt.shape
        assert np.abs(y - yt).max() <= 1e-4 * max(1.0, np.abs(yt).max()), name


def test_fp16_emulation_is_close_to_fp32(onet):
    g = np.load(golden_path("net_blocks.npz"))
    x = g["x"].astype(np.float32)
    h32, h16 = onet.forward(x), onet.forward(x, emulate_fp16=True)
    assert np.abs(h32 - h16).max() < 5e-2 and not np.array_equal(h32, h16)


def test_full_size_summary_golden(onet, frame0):
    from oracle import oracle
    g = np.load(golden_path("full_640.npz"))
    head = onet.forward(oracle.preprocess(frame0, 640))
    flat = head.reshape(-1)
    assert np.array_equal(flat[g["top_idx"]], g["top_val"])
    assert np.allclose(head.mean(0), g["col_mean"], atol=1e-5) and np.allclose(head.std(0), g["col_std"], atol=1e-4)
    d = oracle.decode_nms(head, 640, 14, 8)
    assert d["n_candidates"] == int(g["n_candidates"])
    assert np.array_equal(d["anchors"], g["anchors"]) and np.array_equal(d["classes"], g["classes"])
    assert np.array_equal(d["boxes"], g["boxes"]) and np.array_equal(d["scores"], g["scores"])
    assert np.array_equal(d["kpts"], g["kpts"])
