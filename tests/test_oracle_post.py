"""Oracle decode + NMS + parse_output (EfficientNMS semantics, SURVEY.md App. B;
reference src/yolo_engine.cpp:202-220) against numpy restatements and goldens."""
import json
import math

import numpy as np

from conftest import golden_path
from oracle import oracle


def np_iou(a, b):
    ix = max(0.0, min(a[2], b[2]) - max(a[0], b[0])); iy = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = ix * iy
    u = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / u if u > 0 else float("nan")


def test_expf_is_accurate():
    xs = np.concatenate([np.linspace(-87, 88, 3001), np.linspace(-2, 2, 2001)]).astype(np.float32)
    got = np.array([oracle.expf(float(x)) for x in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert (np.abs(got - ref) / ref).max() < 3e-7
    assert oracle.expf(0.0) == 1.0


def test_nms_golden_cases():
    cases = json.load(open(golden_path("nms_cases.json")))
    assert set(cases) >= {"empty", "one", "overlap_same_class", "overlap_diff_class", "more_than_max_det", "ties_identical_boxes"}
    for name, c in cases.items():
        keep = oracle.nms_sorted(np.array(c["boxes"], np.float32).reshape(-1, 4), np.array(c["classes"], np.int32), c["thr"], c["max_det"])
        assert list(keep) == c["keep"], name
    assert len(cases["more_than_max_det"]["keep"]) == 100
    assert cases["overlap_diff_class"]["keep"][:3] == [0, 1, 2]      # other classes never suppress
    assert cases["ties_identical_boxes"]["keep"] == [0, 3, 4]


def _random_head(rng, A=8400, nc=14, nk=8, hot=0.004):
    head = np.zeros((A, 64 + nc + nk), np.float32)
    head[:, :64] = rng.standard_normal((A, 64)) - 0.4 * (np.arange(64) % 16)
    cls = rng.standard_normal((A, nc)) - 6.0
    mask = rng.random((A, nc)) < hot
    cls[mask] = rng.uniform(-1.0, 4.0, mask.sum())
    head[:, 64:64 + nc] = cls
    head[:, 64 + nc:] = 0.25 + 0.3 * rng.standard_normal((A, nk))
    return head


def np_decode(head, net=640, nc=14):
    """float64 numpy restatement of SURVEY.md App. A.2"""
    boxes = np.zeros((len(head), 4))
    a = 0
    for s in (8, 16, 32):
        w = net // s
        for iy in range(w):
            for ix in range(w):
                if a >= len(head):
                    return boxes
                l = head[a, :64].astype(np.float64).reshape(4, 16)
                p = np.exp(l - l.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
                d = (p * np.arange(16)).sum(1)
                boxes[a] = [(ix + .5 - d[0]) * s, (iy + .5 - d[1]) * s, (ix + .5 + d[2]) * s, (iy + .5 + d[3]) * s]
                a += 1
    return boxes


def test_decode_matches_numpy_and_order_rules():
    rng = np.random.default_rng(1)
    head = _random_head(rng)
    boxes, keys = oracle.decode_candidates(head, 640, 14, 8, 0.25)
    ref = np_decode(head[:400])                       # first 400 anchors are enough (pure-Python loop)
    assert np.abs(boxes[:400] - ref).max() < 2e-3
    lt = math.log(0.25 / 0.75)
    cls = head[:, 64:78]
    assert len(keys) == int((cls > np.float32(lt)).sum())
    ids = (0xFFFFFFFF - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
    logits = cls.reshape(-1)[ids]
    assert (np.diff(logits) <= 0).all()               # score-descending
    ties = np.diff(logits) == 0
    assert (np.diff(ids)[ties] > 0).all()             # ties: lower anchor, then lower class, first


def test_decode_nms_matches_bruteforce_walk():
    rng = np.random.default_rng(2)
    for hot, thr, iou_thr, max_det in ((0.004, 0.25, 0.45, 100), (0.02, 0.4, 0.6, 30), (0.0005, 0.25, 0.45, 100)):
        head = _random_head(rng, hot=hot)
        d = oracle.decode_nms(head, 640, 14, 8, thr, iou_thr, max_det)
        boxes, keys = oracle.decode_candidates(head, 640, 14, 8, thr)
        ids = (0xFFFFFFFF - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
        anc, cls = ids // 14, ids % 14
        keep = []
        for i in range(len(ids)):
            if len(keep) >= max_det:
                break
            if all(not (cls[j] == cls[i] and np_iou(boxes[anc[j]].astype(np.float64), boxes[anc[i]].astype(np.float64)) > iou_thr) for j in keep):
                keep.append(i)
        assert d["num_dets"] == len(keep)
        assert list(d["anchors"]) == list(anc[keep]) and list(d["classes"]) == list(cls[keep])
        assert (np.diff(d["scores"]) <= 0).all()
        logit = head[d["anchors"], 64 + d["classes"]].astype(np.float64)
        assert np.abs(d["scores"] - 1 / (1 + np.exp(-logit))).max() < 1e-6
        # idempotence: NMS of the survivors keeps them all (size-independent property)
        again = oracle.nms_sorted(d["boxes"], d["classes"], iou_thr, max_det)
        assert list(again) == list(range(d["num_dets"]))


def test_empty_and_pre_nms_cap():
    head = _random_head(np.random.default_rng(3), hot=0.0)
    head[:, 64:78] = -20
    d = oracle.decode_nms(head, 640, 14, 8)
    assert d["num_dets"] == 0 and d["n_candidates"] == 0
    head = _random_head(np.random.default_rng(4), hot=0.05)
    full = oracle.decode_nms(head, 640, 14, 8, pre_nms_cap=8192)
    capped = oracle.decode_nms(head, 640, 14, 8, pre_nms_cap=50)
    assert full["n_candidates"] == capped["n_candidates"] > 4000
    assert capped["num_dets"] <= 50 and list(capped["anchors"]) == list(full["anchors"][:capped["num_dets"]])


def test_keypoint_decode_rule():
    # k = (2 v + (anchor - 0.5)) * stride
    head = np.zeros((8400, 86), np.float32); head[:, 64:78] = -20
    a = 6400 + 5 * 40 + 7                              # P4 anchor (ix=7, iy=5), stride 16
    head[a, 64 + 3] = 5.0
    head[a, 78:] = [0.5, -0.25, 1.0, 0.0, 0.25, 0.75, -1.0, 2.0]
    d = oracle.decode_nms(head, 640, 14, 8)
    assert d["num_dets"] == 1 and d["anchors"][0] == a and d["classes"][0] == 3
    exp = [(2 * 0.5 + 7) * 16, (2 * -0.25 + 5) * 16, (2 * 1.0 + 7) * 16, (2 * 0 + 5) * 16,
           (2 * .25 + 7) * 16, (2 * .75 + 5) * 16, (2 * -1 + 7) * 16, (2 * 2 + 5) * 16]
    assert np.allclose(d["kpts"][0], exp)


def test_parse_output_scaling():
    # reference src/yolo_engine.cpp:155-156,211-214: x * W/640, y * H/640
    b = np.array([[10, 20, 110, 220], [0, 0, 640, 640]], np.float32)
    out = oracle.parse_output(b, 1280, 1024, 640, 0)
    assert np.array_equal(out, b * np.array([2.0, 1.6, 2.0, 1.6], np.float32))
    lb = oracle.parse_output(b, 1280, 1024, 640, 1)       # letterbox: r = 0.5, pad_y = 64
    assert np.allclose(lb, (b - np.array([0, 64, 0, 64], np.float32)) * 2.0)


def test_margin_fixtures_hold_under_fp16_noise(blob, onet):
    """tests/golden/margin_cases.json (made by make_margin.py from the fp32 oracle): the fp32 oracle reproduces the
    committed survivors, and the fp16-EMULATING oracle -- a differently rounded head -- returns the identical
    survivor set, i.e. the margins do what they are built for."""
    from irmv_detection_amd import frames
    m = json.load(open(golden_path("margin_cases.json")))
    assert len(m["cases"]) >= 4
    for c in m["cases"][:3]:
        x = oracle.preprocess(frames.synthetic_frame(c["frame"]), 640)
        for emu in (False, True):
            d = oracle.decode_nms(onet.forward(x, emulate_fp16=emu), 640, 14, 8, c["score_thr"], c["iou_thr"], m["max_det"], m["pre_nms_cap"])
            assert d["n_candidates"] == c["n_candidates"]
            assert d["anchors"].tolist() == c["anchors"] and d["classes"].tolist() == c["classes"], (c["frame"], emu)
            assert 0 < d["num_dets"] < d["n_candidates"]                  # something is suppressed: not vacuous
        assert np.array_equal(d["anchors"], np.array(c["anchors"]))
