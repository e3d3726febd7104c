"""GPU parity tests: the HIP path, called through the C ABI (ctypes mirror of the
reference's YoloEngine / PnPSolver interface), against the CPU oracle.

Stated tolerances (DESIGN.md "Parity"):
  preprocess            bit-exact (integer taps; fp16 of v/255 is unique)
  activations / head    fp16 storage, fp32 accumulate, vs the fp32 oracle, stated on frames nobody picked
                        (test_head_error_over_unchosen_frames: synthetic frames 0 .. 63 + the reference's rm_test.jpg):
                        every frame's max |d| <= HEAD_TOL = 4e-2 on logits of magnitude ~20; at least 90 % of the frames
                        <= HEAD_P90 = 3e-2 (SURVEY 8c's figure); median frame <= 2.5e-2.  Measured in round 5: max 0.0348
                        (frame 17), p90 0.0279, median 0.0209 -- the accumulated rounding of ~25 fp16 activation tensors
                        (relative rms error 3e-4 behind model.0, 1e-3 behind model.15: tests/head_sweep.py), which is also
                        where the oracle's own fp16-emulating mode sits (0.029 over 11 frames).  Against that
                        fp16-EMULATING mode the bound is EMU_TOL = 6e-2: two fp16 pipelines with different accumulation
                        orders (measured 0.045)
  decode / NMS / kpts   bit-exact on identical head tensors (survivor set AND order);
                        end to end vs the fp32 oracle: survivor set IDENTICAL on the margin fixtures
                        (tests/golden/margin_cases.json: every decode / NMS decision clear of fp16 noise), boxes
                        within BOX_TOL(stride) px, scores within 5e-3, keypoints within 0.75 px
  PnP                   fp64: |d rvec|, |d tvec| <= 1e-6

Box tolerance per stride: a head-logit error e moves a DFL side (the expectation over 16 bins) by a multiple of e
that grows with the spread of the bin distribution; in pixels that is x stride.  Measured against the fp32 oracle over
~20 frames and several builds (the K order of a layer changes its rounding): 0.30 / 0.58 / 1.35 px at strides 8 / 16 /
32 = 0.04 bins.  SURVEY's 0.5 px therefore holds at stride 8, is marginal at 16 and cannot hold at 32 with fp16
activations; the bar is max(0.5 px, 0.05 bins) = 0.5 / 0.8 / 1.6 px.  Keypoints (2 * v * stride, |dv| <~ 3e-3): 0.75 px.
"""
import json
import zlib

import numpy as np
import pytest

from conftest import D_REF, K_REF, ROOT, golden_path
from irmv_detection_amd import capi, frames
from irmv_detection_amd.engine import Armor, ArmorClass, Light, PnPSolver, YoloEngine, bbox
from oracle import oracle

pytestmark = pytest.mark.gpu

HEAD_TOL = 4e-2     # any frame's max |d head| vs the fp32 oracle
HEAD_P90 = 3e-2     # ... and what 90 % of the frames stay under (SURVEY 8c's 3e-2 is not a bound over all frames: two of 65 exceed it)
HEAD_MEDIAN = 2.5e-2
EMU_TOL = 6e-2      # vs the oracle's fp16-emulating mode (another fp16 pipeline)
KPT_TOL = 0.75
SCORE_TOL = 5e-3


def BOX_TOL(stride):
    """Unconstrained frames only (a DFL expectation moves by up to ~0.04 bins under a 3e-2 logit error: 1.3 px at stride 32)."""
    return max(0.5, 0.05 * stride)


MARGIN_BOX_TOL = 0.5    # SURVEY 8c's own bar, asserted where it holds: the margin fixtures (measured 0.11 / 0.38 / 0.36 px at strides 8 / 16 / 32)


def _report_head(tag, err):
    """Every parity test prints the head's max |d| vs the fp32 oracle: the headroom under HEAD_TOL is tracked per round in the test log."""
    print(f"head max|d| vs fp32 oracle [{tag}]: {err:.4f} (tolerance {HEAD_TOL})")
    return err


def _stride(anchor):
    return 8 if anchor < 6400 else (16 if anchor < 8000 else 32)


@pytest.fixture(scope="module")
def eng(blob):
    e = YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=3)
    yield e
    e.close()


def _load(eng, slot, img):
    eng.get_src_image_buffer(slot)[:] = img


def _raw_tuple(r):
    return (r["num_dets"], r["boxes"].copy(), r["scores"].copy(), r["classes"].copy(), r["anchors"].copy(), r["kpts"].copy())


# ---------------------------------------------------------------- preprocess
@pytest.mark.parametrize("size,mode,rot,swap", [
    ((1280, 1024), 0, True, False),      # the reference configuration (src/yolo_engine.cpp:179-200)
    ((1280, 1024), 0, False, True),
    ((1280, 1024), 1, True, False),      # letterbox (north-star variant)
    ((640, 640), 0, True, False),        # BASELINE configs[1]: identity scale
    ((641, 479), 0, True, False),        # rows not 16-byte aligned -> scalar staging path
    ((333, 1000), 1, False, False),
    ((1276, 1280), 1, True, False),      # 2 : 1 behind one pad column (the fused front's direct tiles with de = 0: test_fused_kernels_are_bitwise_identical)
    ((1276, 1280), 1, False, True),
])
def test_preprocess_bit_exact(blob, size, mode, rot, swap):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (size[1], size[0], 3), dtype=np.uint8)
    with YoloEngine(None, size, weights_blob=blob, resize_mode=mode, rotate180=rot, swap_rb=swap) as e:
        _load(e, 0, img)
        e.detect()
        got = e.read_input(0)
    exp = oracle.preprocess(img, 640, mode, rot, swap).astype(np.float16).astype(np.float32)
    assert np.array_equal(got, exp)


def test_preprocess_rm_test_jpg_golden(eng, rm_test_image):
    meta = json.load(open(golden_path("rm_test_pre.json")))
    if zlib.crc32(rm_test_image.tobytes()) != meta["src_crc32"]:
        pytest.skip("JPEG decoder differs from the one that produced the golden CRC")
    _load(eng, 0, rm_test_image)
    eng.detect(0)
    assert zlib.crc32(eng.read_input(0).astype(np.float16).tobytes()) == meta["fp16_chw_crc32"]


def test_rotated_image_matches_reference_semantics(eng, frame0):
    # get_rotated_image() == the frame after nppiMirror both axes (src/yolo_engine.cpp:182-184)
    _load(eng, 1, frame0)
    assert np.array_equal(eng.get_rotated_image(1), frame0[::-1, ::-1])


# ---------------------------------------------------------------- network
def test_network_taps_and_head_vs_oracle(eng, onet, frame0):
    _load(eng, 0, frame0)
    eng.detect(0)
    x = oracle.preprocess(frame0, 640)
    for tap in ("0", "1", "2", "3", "4", "6", "8", "9", "12", "15", "16", "18", "21", "22.cv2.0.1", "22.cv4.2.1"):
        _, t_o = onet.forward(x, emulate_fp16=True, tap=tap)
        t_g = eng.read_tap(tap, 0)
        assert t_g.shape == t_o.shape, tap
        assert np.abs(t_g - t_o).max() <= EMU_TOL, tap     # fp16-emulating oracle: see the module docstring
        assert np.abs(t_g - t_o).mean() <= 2e-3, tap
    h_g = eng.read_head(0)
    assert _report_head("frame 0, 640 net", float(np.abs(h_g - onet.forward(x)).max())) <= HEAD_TOL   # vs the fp32 oracle: the contract
    assert np.abs(h_g - onet.forward(x, emulate_fp16=True)).max() <= EMU_TOL     # vs another fp16 pipeline


def test_network_on_golden_block_input(blob, onet):
    """64x64 crop fixture: per-block activations against the committed goldens."""
    g = np.load(golden_path("net_blocks.npz"))
    x = g["x"].astype(np.float32)                                 # [3,64,64], fp16-representable
    img = np.clip(np.rint(x.transpose(1, 2, 0) * 255), 0, 255).astype(np.uint8)
    with YoloEngine(None, (64, 64), weights_blob=blob, net_size=64, rotate180=False) as e:
        _load(e, 0, img)
        e.detect()
        xin = e.read_input(0)
        head = e.read_head(0)
        taps = {t: e.read_tap(t, 0) for t in ("0", "2", "4", "9", "15", "21")}
    assert np.array_equal(xin, (img.astype(np.float32) / np.float32(255)).astype(np.float16).astype(np.float32).transpose(2, 0, 1))
    head_o = onet.forward(xin)
    assert _report_head("golden 64 x 64 crop", float(np.abs(head - head_o).max())) <= HEAD_TOL
    for t, v in taps.items():
        _, to = onet.forward(xin, tap=t)
        assert np.abs(v - to).max() <= HEAD_TOL, t
    # if the u8 round trip reproduced the fixture input exactly, the goldens themselves apply
    exact = np.array_equal(xin, x)
    print(f"golden branch: {'committed net_blocks.npz head applied' if exact else 'fixture input not reproduced by the u8 round trip: oracle-on-the-fly only'}")
    if exact:
        assert np.abs(head - g["head"]).max() <= HEAD_TOL


def test_activation_range_under_the_log2e_scale(blob):
    """Activations are stored as log2(e) * a (irmv_common.hpp): the fp16 range ends at |a| = 65504 / log2(e) = 45 403 instead of
    65 504.  Pinned on both sides of that edge with a model.0 whose channels 0 / 1 are constants (zero weights, bias 4.0e4 /
    5.0e4): 4.0e4 comes back finite and equal to the fp16-emulating oracle's value to one fp16 step of the stored number
    (32 * ln 2); 5.0e4 -- representable in a plain fp16 pipeline, which is what the oracle's emulating mode is -- overflows
    to +inf here (documented deviation; nothing saturates, nothing faults, the step completes)."""
    from irmv_detection_amd import weights
    hdr, layers = weights.parse_blob(blob)
    specs, tensors = [], []
    for sp, w, b in layers:
        w, b = w.copy(), b.copy()
        if sp.name == "model.0.conv":
            w[0:2] = 0
            b[0], b[1] = 4.0e4, 5.0e4
        specs.append(sp); tensors.append((w, b))
    hot = weights.build_blob(specs, tensors, hdr["nc"], hdr["nk"], hdr["backbone"])
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    with YoloEngine(None, (64, 64), weights_blob=hot, net_size=64, rotate180=False) as e:
        _load(e, 0, img)
        e.detect()                                        # inf / NaN run through every later layer, NMS and PnP: no fault
        t0 = e.read_tap("0", 0)
        e.detect()
    x = (img.astype(np.float32) / np.float32(255)).astype(np.float16).astype(np.float32).transpose(2, 0, 1)
    _, o0 = oracle.Net(hot).forward(x, emulate_fp16=True, tap="0")
    assert np.isfinite(t0[..., 0]).all() and np.abs(t0[..., 0] - o0[..., 0]).max() <= 32.0 * 0.7, (t0[0, 0, 0], o0[0, 0, 0])
    assert np.isfinite(o0[..., 1]).all() and o0[0, 0, 1] > 4.9e4      # the plain fp16 pipeline still holds 5e4 ...
    assert np.isposinf(t0[..., 1]).all()                                # ... the scaled one does not: +inf, not NaN, not a clamp
    assert np.isfinite(t0[..., 2:]).all() and np.abs(t0[..., 2:] - o0[..., 2:]).max() <= EMU_TOL


def test_head_error_over_unchosen_frames(blob, onet, rm_test_image, capsys):
    """The head tolerance as a statement over frames nobody picked: synthetic frames 0 .. 63 (every seed in order) and the
    reference's own test/rm_test.jpg, through detect() on the reference node's engine shape.  Prints max / p99 / median per
    Detect level and branch (the distribution goes to the test log) and asserts the bound DESIGN section 5 states."""
    import head_sweep
    def it():
        for fi in range(64):
            yield fi, frames.synthetic_frame(fi)
        yield "rm_test", rm_test_image
    lines = []
    with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
        res = head_sweep.sweep(e, onet, it(), oracle, log=lines.append)
    mx, p99, med = head_sweep.report(res, log=lines.append)
    with capsys.disabled():
        print("\n" + "\n".join(lines[-11:]))
    per = np.array([m for _, m in res["per_frame"]])
    assert len(per) == 65
    assert mx <= HEAD_TOL, sorted(res["per_frame"], key=lambda t: -t[1])[:5]
    assert np.quantile(per, 0.9) <= HEAD_P90
    assert med <= HEAD_MEDIAN
    # nothing vacuous: most frames carry structure (frames without an armor give a near-constant head: |d| ~ 4e-4)
    assert (per > 5e-3).sum() >= 55
    # the keypoint branch, whose logits feed pixel coordinates directly (2 * v * stride), stays an order of magnitude tighter
    assert max(max(res["cells"][(li, "kpt")]) for li in range(3)) <= 1e-2


# ---------------------------------------------------------------- decode / NMS / keypoints
def _assert_post_exact(eng, head, slot=0, **kw):
    eng.write_head(head, slot)
    eng.run_post(slot, 1)
    raw = eng.read_raw(slot)
    exp = oracle.decode_nms(head, 640, 14, 8, kw.get("score_thr", 0.25), kw.get("iou_thr", 0.45),
                            kw.get("max_det", 100), kw.get("pre_nms_cap", 4096))
    assert raw["n_candidates"] == exp["n_candidates"]
    assert raw["num_dets"] == exp["num_dets"]
    assert np.array_equal(raw["anchors"], exp["anchors"]) and np.array_equal(raw["classes"], exp["classes"])
    assert np.array_equal(raw["boxes"], exp["boxes"])              # bit-exact fp32 decode
    assert np.array_equal(raw["scores"], exp["scores"])
    assert np.array_equal(raw["kpts"], exp["kpts"])
    n = raw["num_dets"]
    assert not raw["boxes_padded"][n:].any() and not raw["scores_padded"][n:].any()   # zero padded like EfficientNMS
    return raw


def _synthetic_head(rng, hot, A=8400):
    head = np.zeros((A, 86), np.float32)
    head[:, :64] = rng.standard_normal((A, 64)) - 0.4 * (np.arange(64) % 16)
    cls = rng.standard_normal((A, 14)) - 6.0
    mask = rng.random((A, 14)) < hot
    cls[mask] = rng.uniform(-1.0, 4.0, mask.sum())
    head[:, 64:78] = cls
    head[:, 78:] = 0.25 + 0.3 * rng.standard_normal((A, 8))
    return head.astype(np.float32)


def test_post_exact_on_oracle_heads(eng, onet):
    for fi in (0, 1, 2):
        head = onet.forward(oracle.preprocess(frames.synthetic_frame(fi), 640))
        raw = _assert_post_exact(eng, head)
        assert (np.diff(raw["scores"]) <= 0).all()


@pytest.mark.parametrize("hot,expect", [(0.0, "empty"), (0.00002, "few"), (0.004, "typical"), (0.0055, "crowded"), (0.008, "crowded"),
                                        (0.03, "more_than_pre_nms_cap_is_fine")])
def test_post_exact_on_synthetic_heads(eng, hot, expect):
    rng = np.random.default_rng(int(hot * 1e6) + 1)
    head = _synthetic_head(rng, hot)
    if expect == "empty":
        head[:, 64:78] = -20.0
    raw = _assert_post_exact(eng, head)
    if expect == "empty":
        assert raw["num_dets"] == 0 and raw["n_candidates"] == 0
    if expect == "crowded":                              # 513 .. 1024 candidates: the matrix one block of rows at a time
        assert 512 < raw["n_candidates"] <= 1024
    if expect == "more_than_pre_nms_cap_is_fine":
        assert 2500 < raw["n_candidates"] <= capi.CAND_CAP


def test_post_ties_and_identical_boxes(eng):
    head = np.zeros((8400, 86), np.float32)
    head[:, 64:78] = -20.0
    # identical logits on several anchors/classes: order must be lower anchor, then lower class
    for a, c in ((100, 5), (100, 2), (50, 9), (7000, 0), (8399, 13)):
        head[a, 64 + c] = 1.5
    head[51, 64 + 9] = 1.5          # neighbour of anchor 50 with an identical box shape -> suppressed (same class)
    raw = _assert_post_exact(eng, head)
    assert list(raw["anchors"][:2]) == [50, 100] and raw["classes"][1] == 2


def _class_walk_heads():
    """Heads of up to 512 candidates (and three above, for the first walk on the head of the list) that lean on every part of
    nms_pnp_kernel's class-major path: the widest class in one to eight 64-bit words, chains of suppression inside a class,
    equal scores, several classes on one anchor, classes without candidates."""
    out = {}
    rng = np.random.default_rng(404)
    out["fourteen classes, ~ 400"] = _synthetic_head(rng, 0.0034)
    out["fourteen classes, ~ 120"] = _synthetic_head(rng, 0.001)
    one = np.zeros((8400, 86), np.float32)                   # ONE class, 500 candidates on neighbouring stride-8 anchors: wide,
    one[:, 64:78] = -20.0                                    # heavily overlapping boxes -> rows of eight words, long chains
    one[:, :64] = 0.05 * rng.standard_normal((8400, 64))
    one[:, 78:] = 0.25 + 0.3 * rng.standard_normal((8400, 8))
    one[:500, 64 + 6] = 1.0 + rng.permutation(500).astype(np.float32) * 2e-3
    out["one class, 500 overlapping"] = one
    two = one.copy()                                         # two interleaved classes of 250 (130 + 120 words apart) + singletons
    two[:, 64:78] = -20.0
    two[0:500:2, 64 + 1] = 1.0 + rng.permutation(250).astype(np.float32) * 2e-3
    two[1:500:2, 64 + 12] = 1.0 + rng.permutation(250).astype(np.float32) * 2e-3
    two[6000:6010, 64 + 13] = 2.0
    two[8399, 64 + 0] = 0.5
    out["two classes interleaved + equal scores"] = two
    ties = one.copy()                                        # every candidate the SAME logit: order = anchor, then class
    ties[:, 64:78] = -20.0
    ties[:300, 64 + 4] = 1.25
    ties[100:200, 64 + 9] = 1.25                             # ... and a second class on a hundred of those anchors
    out["all scores equal, two classes per anchor"] = ties
    lone = one.copy()
    lone[:, 64:78] = -20.0
    lone[4242, 64 + 7] = 3.0
    out["one candidate"] = lone
    iso = np.zeros((8400, 86), np.float32)                   # 300 candidates with point-like boxes (DFL mass on bin 0): nobody suppresses
    iso[:, 64:78] = -20.0                                    # anybody -> 300 survivors, cut at max_det (256: the last records reach into
    iso[:, 0:64:16] = 12.0                                   # the LDS that held the keypoint logits)
    iso[:, 78:] = 0.25 + 0.3 * rng.standard_normal((8400, 8))
    pick = rng.choice(8400, 300, replace=False)
    iso[pick, 64 + rng.integers(0, 14, 300)] = rng.uniform(0.0, 3.0, 300).astype(np.float32)
    out["300 isolated candidates"] = iso
    out["cluster of 700 in front (first walk starts over)"] = _clustered_head(rng, 700, 500)
    out["cluster of 400 in front of 800"] = _clustered_head(rng, 400, 800)
    return out


@pytest.mark.parametrize("cap,md", [(4096, 100), (150, 100), (4096, 7), (2048, 256)])
def test_class_walk_path_on_crafted_heads(blob, monkeypatch, cap, md):
    """run_post takes its candidates from scan_decode_kernel, which also decodes their boxes -- nms_pnp_kernel then skips its
    class-major path (the one every whole step takes: sort, decode in score order into LDS, per-class rows and walks, cap in
    score order).  IRMV_POST_KEYS_ONLY=1 makes run_post's NMS decode its own boxes as a whole step's does, so the crafted heads
    reach that path: each against the oracle, bit for bit, at four (pre_nms_cap, max_det) pairs -- a cap below the candidate
    count, a max_det the first class fills, the largest max_det (records reach into the keypoints' LDS) -- and against the
    round-3 walk (IRMV_NMS_CLASSWALK=0)."""
    monkeypatch.setenv("IRMV_POST_KEYS_ONLY", "1")
    got = {}
    for cw in ("1", "0"):
        monkeypatch.setenv("IRMV_NMS_CLASSWALK", cw)
        with YoloEngine(None, (1280, 1024), weights_blob=blob, pre_nms_cap=cap, max_det=md) as e:
            for name, head in _class_walk_heads().items():
                raw = _assert_post_exact(e, head, max_det=md, pre_nms_cap=cap)
                got[(cw, name)] = _raw_tuple(raw)
                if name == "one class, 500 overlapping":
                    assert raw["n_candidates"] == 500 and 1 < raw["num_dets"] < 500
                if name == "fourteen classes, ~ 400":
                    assert 300 < raw["n_candidates"] <= 512
                if name == "300 isolated candidates":
                    assert raw["n_candidates"] == 300 and raw["num_dets"] == min(md, 300 if cap >= 300 else cap)
    for name in _class_walk_heads():
        a, b = got[("1", name)], got[("0", name)]
        assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:])), name


def test_post_max_det_and_thresholds(blob):
    rng = np.random.default_rng(11)
    head = _synthetic_head(rng, 0.01)
    with YoloEngine(None, (1280, 1024), weights_blob=blob, max_det=17, score_thr=0.4, iou_thr=0.6, pre_nms_cap=300) as e:
        raw = _assert_post_exact(e, head, max_det=17, score_thr=0.4, iou_thr=0.6, pre_nms_cap=300)
        assert raw["num_dets"] == 17


def test_more_candidates_than_the_lds_sort_holds(eng):
    """> 8192 (anchor, class) pairs above threshold: the kernel radix-selects exactly the pre_nms_cap best
    keys before sorting -- same survivors as the oracle's sort-everything-then-cut."""
    head = np.zeros((8400, 86), np.float32)
    head[:, 64:78] = 3.0                                # all 117 600 pairs pass, all tied: order = anchor, then class
    raw = _assert_post_exact(eng, head)
    assert raw["n_candidates"] == 8400 * 14 and raw["num_dets"] > 0
    rng = np.random.default_rng(21)
    head = _synthetic_head(rng, 0.15)                   # ~17 000 distinct scores
    raw = _assert_post_exact(eng, head)
    assert raw["n_candidates"] > capi.CAND_CAP
    eng.results(0)                                      # no overflow error any more


@pytest.mark.parametrize("cap", [100, 300, 512, 513, 800, 1024, 1025, 2048])
def test_more_candidates_than_the_lds_sort_holds_small_pre_nms_cap(blob, cap):
    """> 8192 candidates AND pre_nms_cap <= 512: the radix-selected keys go through the rank sort, which must sort the
    SELECTED keys (round 1 re-read the first K raw keys there).  513 / 2048: the bitonic branch after a select."""
    rng = np.random.default_rng(33)
    head = _synthetic_head(rng, 0.15)
    with YoloEngine(None, (1280, 1024), weights_blob=blob, pre_nms_cap=cap) as e:
        raw = _assert_post_exact(e, head, pre_nms_cap=cap)
        assert raw["n_candidates"] > capi.CAND_CAP and raw["num_dets"] > 0


# ---------------------------------------------------------------- end to end / API
def test_detect_api_end_to_end(eng, onet, frame0):
    """detect() as the reference tests use it (test/yolo_test.cpp:27-36)."""
    buf = eng.get_src_image_buffer(2)
    buf[:] = frame0
    bboxes = eng.detect(2)
    assert all(isinstance(b, bbox) and isinstance(b.class_id, ArmorClass) for b in bboxes)
    assert eng.get_profiling_time() > 0
    # what the GPU produced == oracle post-processing of the GPU's own head, scaled by parse_output
    raw = eng.read_raw(2)
    exp = oracle.decode_nms(eng.read_head(2), 640, 14, 8)
    assert len(bboxes) == exp["num_dets"] == raw["num_dets"]
    xy = oracle.parse_output(exp["boxes"], 1280, 1024, 640, 0)
    assert np.array_equal(np.array([b.xyxy for b in bboxes], np.float32), xy)
    assert [int(b.class_id) for b in bboxes] == list(exp["classes"])
    # and against the fp32 oracle end to end: same detections up to fp16 noise
    ref = oracle.decode_nms(onet.forward(oracle.preprocess(frame0, 640)), 640, 14, 8)
    got = {(int(a), int(c)): i for i, (a, c) in enumerate(zip(raw["anchors"], raw["classes"]))}
    want = {(int(a), int(c)): i for i, (a, c) in enumerate(zip(ref["anchors"], ref["classes"]))}
    # frame0 at the default thresholds is NOT a margin fixture (max_det binds, candidates sit on both thresholds):
    # here only the shared survivors are compared; set identity is asserted in test_margin_fixtures_*
    common = set(got) & set(want)
    assert len(common) >= 0.9 * max(len(want), 1)
    for k in common:
        assert np.abs(raw["boxes"][got[k]] - ref["boxes"][want[k]]).max() <= BOX_TOL(_stride(k[0]))
        assert abs(raw["scores"][got[k]] - ref["scores"][want[k]]) <= SCORE_TOL
        assert np.abs(raw["kpts"][got[k]] - ref["kpts"][want[k]]).max() <= KPT_TOL


def test_margin_fixtures_survivor_set_identical_end_to_end(blob):
    """on frames whose every decode / NMS decision is clear of fp16 noise (tests/golden/make_margin.py) the
    HIP path, frame in -> detections out, returns EXACTLY the fp32 oracle's survivors, in the oracle's order."""
    m = json.load(open(golden_path("margin_cases.json")))
    assert len(m["cases"]) >= 4
    worst = {8: [0.0, 0.0], 16: [0.0, 0.0], 32: [0.0, 0.0]}
    for c in m["cases"]:
        with YoloEngine(None, (1280, 1024), weights_blob=blob, score_thr=c["score_thr"], iou_thr=c["iou_thr"],
                        max_det=m["max_det"], pre_nms_cap=m["pre_nms_cap"]) as e:
            _load(e, 0, frames.synthetic_frame(c["frame"]))
            e.detect(0)
            raw = e.read_raw(0)
        assert raw["n_candidates"] == c["n_candidates"], c["frame"]
        assert list(zip(raw["anchors"].tolist(), raw["classes"].tolist())) == list(zip(c["anchors"], c["classes"])), c["frame"]
        for i, a in enumerate(c["anchors"]):
            db = float(np.abs(raw["boxes"][i] - np.array(c["boxes"][i], np.float32)).max())
            dk = float(np.abs(raw["kpts"][i] - np.array(c["kpts"][i], np.float32)).max())
            w = worst[_stride(a)]
            w[0], w[1] = max(w[0], db), max(w[1], dk)
            assert db <= MARGIN_BOX_TOL, (c["frame"], a, db)    # SURVEY's 0.5 px, at every stride
            assert dk <= KPT_TOL, (c["frame"], a, dk)
            assert abs(float(raw["scores"][i]) - c["scores"][i]) <= SCORE_TOL
    print("margin fixtures: max |d box|, |d kpt| px per stride:", worst)


def test_fused_pnp_matches_oracle(eng, frame0):
    _load(eng, 0, frame0)
    armors = eng.detect_armors(0)
    assert len(armors) > 0
    n_ok = 0
    for a in armors:
        o = oracle.solve_pnp_ippe(K_REF, D_REF, a.image_points(), 0)
        assert o["ok"] == a.pnp_ok
        if a.pnp_ok:
            n_ok += 1
            assert np.abs(o["rvec"] - a.rvec).max() <= 1e-6 and np.abs(o["tvec"] - a.tvec).max() <= 1e-6
            q = oracle.rvec_to_quat(a.rvec)
            assert min(np.abs(q - a.quat_xyzw).max(), np.abs(q + a.quat_xyzw).max()) <= 1e-6
    assert n_ok > 0


def test_batched_step_equals_per_slot_detect(eng):
    imgs = [frames.synthetic_frame(10 + i) for i in range(3)]
    single = []
    for s, im in enumerate(imgs):
        _load(eng, s, im)
        eng.detect(s)
        single.append((eng.read_head(s).copy(), eng.read_raw(s)))
    eng.submit(0, 3, h2d=True)
    eng.wait()
    for s in range(3):
        assert np.array_equal(eng.read_head(s), single[s][0])          # bitwise: batch rides the GEMM M axis
        r = eng.read_raw(s)
        assert r["num_dets"] == single[s][1]["num_dets"] and np.array_equal(r["boxes"], single[s][1]["boxes"])
    assert not np.array_equal(single[0][0], single[1][0])


def test_repeated_detect_is_deterministic_and_not_in_place(eng, frame0):
    # the reference mirrors in place, so a second detect() on an un-refreshed buffer un-rotates
    # (SURVEY.md App. E.4); here the rotation is folded into sampling and the slot is never modified
    _load(eng, 0, frame0)
    a = eng.detect(0)
    b = eng.detect(0)
    assert a == b and np.array_equal(eng.get_src_image_buffer(0), frame0)


def test_visualize_bboxes(eng, frame0):
    img = frame0.copy()
    eng.visualize_bboxes(img, [bbox((100.0, 100.0, 200.0, 180.0), 0.9, ArmorClass.B3), bbox((300.0, 300.0, 400.0, 380.0), 0.9, ArmorClass.R1)])
    assert (img[100, 100:201] == (0, 0, 255)).all() and (img[300, 300:401] == (255, 0, 0)).all()
    # the class label (src/yolo_engine.cpp:238-241): text origin = the box's top-left corner, glyphs 15 x 21 px above it.
    # 'B' starts with a filled top row of four cells; 'R1': the '1' glyph's first row is the single cell of column 2
    assert (img[79:82, 100:112] == (0, 0, 255)).all() and (img[79:82, 112:115] == frame0[79:82, 112:115]).all()
    assert (img[279:282, 300:312] == (255, 0, 0)).all() and (img[279:282, 324:327] == (255, 0, 0)).all()
    assert (img[279:282, 318:324] == frame0[279:282, 318:324]).all()
    small = np.zeros((10, 10, 3), np.uint8)
    eng.visualize_bboxes(small, [])                        # size mismatch: message + return (src/yolo_engine.cpp:225-228)
    assert not small.any()


def test_model_file_convention(tmp_path, blob):
    # "<stem>.onnx" -> sibling "<stem>.irmw" (counterpart of src/yolo_engine.cpp:28-40)
    p = tmp_path / "yolov7.irmw"
    p.write_bytes(blob)
    with YoloEngine(str(tmp_path / "yolov7.onnx"), (1280, 1024)) as e:
        assert e.num_anchors == 8400 and e.head_channels == 86
    with pytest.raises(capi.IrmvError) as ei:
        YoloEngine(str(tmp_path / "missing.onnx"), (1280, 1024))
    assert ei.value.code == capi.ERR_MODEL
    bad = bytearray(blob); bad[0:4] = b"XXXX"
    with pytest.raises(capi.IrmvError):
        YoloEngine(None, (1280, 1024), weights_blob=bytes(bad))


# ---------------------------------------------------------------- PnPSolver
def test_pnp_solver_api_vs_oracle_and_goldens():
    cases = json.load(open(golden_path("pnp_cases.json")))
    solver = PnPSolver(K_REF, list(D_REF))
    nodist = PnPSolver(K_REF, [0, 0, 0, 0, 0])
    for c in cases:
        s = solver if c["tag"] == "ref" else nodist
        ok, r, t = s.solve_batch(np.array(c["img_pts"], np.float32), c["size"])
        assert ok[0] == 1
        e1 = max(np.abs(r[0] - c["rvec"]).max(), np.abs(t[0] - c["tvec"]).max())
        e2 = max(np.abs(r[0] - c["rvec2"]).max(), np.abs(t[0] - c["tvec2"]).max())
        # fronto-parallel cases have two equally good solutions; order is then numerically arbitrary
        assert e1 <= 1e-6 or (abs(c["err"][0] - c["err"][1]) < 1e-7 and e2 <= 1e-6)
    # reference call shape: solvePnP(armor) with left/right light top/bottom (src/pnp_solver.cpp:36-52)
    c = cases[2]
    p = np.array(c["img_pts"]).reshape(4, 2)
    armor = Armor(left_light=Light(bottom=tuple(p[0]), top=tuple(p[1])), right_light=Light(top=tuple(p[2]), bottom=tuple(p[3])))
    ok, rvec, tvec = solver.solvePnP(armor)
    assert ok and np.abs(rvec - c["rvec"]).max() <= 1e-6 and np.abs(tvec - c["tvec"]).max() <= 1e-6
    ok, _, _ = solver.solve_batch(np.array([100, 100] * 4, np.float32))
    assert ok[0] == 0                                                     # degenerate -> false, like cv::solvePnP
    assert abs(solver.calculateDistanceToCenter((345.943891 + 3, 284.057302 + 4)) - 5.0) < 1e-4
    solver.close(); nodist.close()


def test_pnp_batch_random_quads_vs_oracle():
    rng = np.random.default_rng(5)
    solver = PnPSolver(K_REF, list(D_REF))
    base = np.array([[-40, 15], [-40, -15], [40, -15], [40, 15]], np.float32)
    pts = (base[None] * rng.uniform(0.5, 3, (256, 1, 1)) + rng.uniform(150, 500, (256, 1, 2)) + rng.normal(0, 3, (256, 4, 2))).astype(np.float32)
    ok, r, t = solver.solve_batch(pts.reshape(256, 8))
    for i in range(256):
        o = oracle.solve_pnp_ippe(K_REF, D_REF, pts[i], 0)
        assert bool(ok[i]) == o["ok"]
        if o["ok"] and abs(o["err"][0] - o["err"][1]) > 1e-7:
            assert np.abs(r[i] - o["rvec"]).max() <= 1e-6 and np.abs(t[i] - o["tvec"]).max() <= 1e-6
    solver.close()


def test_tile_choice_is_bitwise_neutral(blob, frame0):
    """The autotuner may pick different (MT, NT) tiles for different batch sizes; every tile walks K in
    the same order, so heads must agree bit for bit between a 1-slot and a 4-slot engine."""
    heads = []
    for S in (1, 4):
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=S) as e:
            for s in range(S):
                e.get_src_image_buffer(s)[:] = frame0
            e.submit(0, S)
            e.wait()
            heads.append(e.read_head(S - 1).copy())
            if S > 1:
                e.detect(0)                       # single-frame step on the batched engine: its own tile set
                heads.append(e.read_head(0).copy())
    assert np.array_equal(heads[0], heads[1]) and np.array_equal(heads[0], heads[2])


@pytest.mark.parametrize("net", [640, 416])
def test_pointwise_kernel_is_bitwise_the_direct_kernel(blob, monkeypatch, net):
    """The persistent 1x1 kernel (weights in LDS, pixel tiles software-pipelined; k_conv.hip) against the direct kernel it
    may replace: same operands, same k order -> same bits, batched (partial last tiles included: 3 frames) and single-frame; at a 416 net the
    26 x 26 and 13 x 13 maps leave odd numbers of 16-pixel units and partial units."""
    imgs = [frames.synthetic_frame(40 + i) for i in range(3)]
    heads = {}
    # IRMV_FORCE_PWN: the multi-block form (one 8-wave workgroup runs a pixel tile against 2 / 4 output-channel blocks whose
    # weights all sit in LDS; the input is read once instead of once per block) wherever it is offered
    for mode in ("IRMV_FORCE_PW", "IRMV_FORCE_PWN", "IRMV_NO_PW"):
        for m in ("IRMV_FORCE_PW", "IRMV_FORCE_PWN", "IRMV_NO_PW", "IRMV_NO_PWN"):
            monkeypatch.delenv(m, raising=False)
        monkeypatch.setenv(mode, "1")
        if mode == "IRMV_FORCE_PW":
            monkeypatch.setenv("IRMV_NO_PWN", "1")
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=3, num_streams=1, net_size=net) as e:
            names = [st["name"] for st in e.profile(0, 3)] + [st["name"] for st in e.profile(0, 1)]
            assert any(n == "conv1x1s1_pw" for n in names) == (mode == "IRMV_FORCE_PW"), (mode, names)
            assert (sum(n.startswith("conv1x1s1_pw_n") for n in names) >= 10) == (mode == "IRMV_FORCE_PWN"), (mode, names)
            if mode == "IRMV_FORCE_PWN":
                assert any(n == "conv1x1s1_pw_n4" for n in names) and any(n == "conv1x1s1_pw_n2" for n in names), names
            for s, im in enumerate(imgs):
                _load(e, s, im)
            e.submit(0, 3); e.wait()
            batched = [e.read_head(s).copy() for s in range(3)]
            taps = [e.read_tap(t, 2).copy() for t in ("6", "8", "9", "12", "18", "21")]
            e.detect(1)
            heads[mode] = batched + [e.read_head(1).copy()] + taps
    for mode in ("IRMV_FORCE_PW", "IRMV_FORCE_PWN"):
        for a, b in zip(heads[mode], heads["IRMV_NO_PW"]):
            assert np.array_equal(a, b), mode
    assert np.array_equal(heads["IRMV_FORCE_PW"][1], heads["IRMV_FORCE_PW"][3])


def test_chunk_major_and_eight_wave_tiles_are_bitwise_the_plain_ones(blob, monkeypatch):
    """LDS 3x3 family: chunk-major order over a workgroup's images (a chunk's weights staged once, one accumulator set per
    image; k_conv.hip CM) and the stride-2 layers' 8-wave workgroup (NWV = 8) against the image-major 4-wave kernels they
    may replace: every output sees the same MFMA sequence -> same bits.  Six frames in one sub-batch: groups of four leave
    a short last group (two images), groups of two three full ones."""
    imgs = [frames.synthetic_frame(60 + i) for i in range(6)]
    heads = {}
    modes = ("IRMV_FORCE_CM", "IRMV_FORCE_W8", "IRMV_FORCE_NT8", "IRMV_NO_CM")   # (NT8: the stride-2 layers with >= 128 output channels on one 128-channel workgroup)
    for mode in modes:
        for m in modes + ("IRMV_NO_W8",):
            monkeypatch.delenv(m, raising=False)
        monkeypatch.setenv(mode, "1")
        if mode == "IRMV_NO_CM":
            monkeypatch.setenv("IRMV_NO_W8", "1")
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=6, num_streams=1) as e:
            names = [st["name"] for st in e.profile(0, 6)]
            n_cm, n_w8 = sum("_cm" in n for n in names), sum("_w8" in n for n in names)
            if mode == "IRMV_FORCE_CM":
                assert n_cm >= 10 and any("_i4_cm" in n for n in names) and any("_cm+1x1" in n for n in names), names
            elif mode == "IRMV_FORCE_W8":
                assert n_w8 == 5, names                      # the five stride-2 convs
            elif mode == "IRMV_FORCE_NT8":
                assert sum("_nt8" in n for n in names) == 3, names   # model.5 / 7 / 19
            else:
                assert n_cm == 0 and n_w8 == 0, names
            for s, im in enumerate(imgs):
                _load(e, s, im)
            e.submit(0, 6); e.wait()
            heads[mode] = [e.read_head(s).copy() for s in range(6)]
            heads[mode + "_taps"] = {t: e.read_tap(t, 5).copy() for t in ("3", "5", "7", "16", "19", "21")}   # the stride-2 convs' outputs, last slot
    for mode in modes[:3]:
        for a, b in zip(heads[mode], heads["IRMV_NO_CM"]):
            assert np.array_equal(a, b), mode
        for k, v in heads[mode + "_taps"].items():
            assert np.array_equal(v, heads["IRMV_NO_CM_taps"][k]), (mode, k)


@pytest.mark.parametrize("net", [640, 416])
def test_resident_weight_kernels_are_bitwise_the_chunked_ones(blob, monkeypatch, net):
    """The Cin = Cout = 64 3x3 layers on the weights-resident kernel (k_conv.hip WR: all 18 k-steps of an image between one
    pair of barriers, the layer's 72 KiB of weights staged once per workgroup) -- as two ping-pong groups of four waves
    where the map tiles into 8 x 16 blocks (80 x 80), in lockstep on row runs elsewhere (40 x 40, 20 x 20; 52 / 26 / 13 at a
    416 net) -- against the chunked 4-wave kernels: same K order (chunk, tap) -> same bits.  Seven frames, three per
    workgroup: two full image groups and a short one; the fused Detect finals (N2 = 1 with candidate emission, N2 = 4) and
    the C2f shortcut convs ride along."""
    imgs = [frames.synthetic_frame(90 + i) for i in range(7)]
    out = {}
    for mode, val in (("IRMV_FORCE_WRES", "3"), ("IRMV_FORCE_WRES", "-3"), ("IRMV_NO_WRES", "1")):
        for m in ("IRMV_FORCE_WRES", "IRMV_NO_WRES"):
            monkeypatch.delenv(m, raising=False)
        monkeypatch.setenv(mode, val)
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=7, num_streams=1, net_size=net) as e:
            names = [st["name"] for st in e.profile(0, 7)]
            n_wr, n_pp = sum("_wres" in n for n in names), sum("_wres_pp" in n for n in names)
            if mode == "IRMV_NO_WRES":
                assert n_wr == 0, names
            elif net == 640:
                assert n_wr >= 14 and (n_pp == 4) == (val == "3"), names   # ping-pong: the four 80 x 80 Detect convs
            else:
                assert n_wr >= 8 and n_pp == 0, names
            for s, im in enumerate(imgs):
                _load(e, s, im)
            e.submit(0, 7); e.wait()
            key = mode + val
            out[key] = [e.read_head(s).copy() for s in range(7)] + [e.read_tap(t, 6).copy() for t in ("6", "12", "18")]
            out[key + "_dets"] = [[(int(d.armor_class), d.confidence, d.bbox_xyxy) for d in e.results(s)] for s in range(7)]
    for key in ("IRMV_FORCE_WRES3", "IRMV_FORCE_WRES-3"):
        for a, b in zip(out[key], out["IRMV_NO_WRES1"]):
            assert np.array_equal(a, b), key
        assert out[key + "_dets"] == out["IRMV_NO_WRES1_dets"], key


@pytest.mark.parametrize("switch", ["IRMV_INLINE_COPIES=1", "IRMV_ZERO_COPY_RESULTS=0", "IRMV_SPLIT_SCAN=0", "IRMV_EMIT_SCAN=0",
                                    "IRMV_FUSED_HEAD=0", "IRMV_MERGE_HEAD0=0", "IRMV_GROUP_HEAD=0", "IRMV_NO_PF2=1", "IRMV_NO_DEEP=1",
                                    "IRMV_FRONT_FASTX=0", "IRMV_FRONT_DIRECT=0", "IRMV_FRONT_TILE8=0",
                                    "IRMV_STREAMS=1", "IRMV_AUTOTUNE=0", "IRMV_GROUP_FORCE=1", "IRMV_NUMA=0", "IRMV_GRAPH_UPLOAD=0", "IRMV_XCD_IMAGES=0", "IRMV_NO_NT8=1", "IRMV_NMS_CLASSWALK=0", "IRMV_NO_PF4=1", "IRMV_WRES_STAGGER=0", "IRMV_BNECK64=0", "IRMV_KPT3=0", "IRMV_UPLOAD_KERNEL=0", "IRMV_SYNC_LAUNCH=graph", "IRMV_SYNC_LAUNCH=eager"])
def test_every_remaining_switch_is_bitwise_the_default(blob, monkeypatch, switch):
    """The environment switches that select between implementations of the same arithmetic (where the copies ride, where the
    results land, where candidates are found, which launches are merged): heads and detections of a batched step, of
    pipelined single-frame steps and of detect() are those of the default configuration, bit for bit."""
    imgs = [frames.synthetic_frame(120 + i) for i in range(4)]

    def run():
        out = []
        for slots in (4, 1):     # a batched engine (two streams) and the reference node's single-frame shape
            with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=slots) as e:
                for s in range(slots):
                    _load(e, s, imgs[s])
                e.submit(0, slots); e.wait()
                out += [e.read_head(s).copy() for s in range(slots)]
                out += [_raw_tuple(e.read_raw(s)) for s in range(slots)]
                for s in range(slots):
                    e.submit(s, 1, async_upload=True)
                for s in range(slots):
                    e.wait_slots(s, 1)
                out += [_raw_tuple(e.read_raw(s)) for s in range(slots)]
                e.detect(0)
                out.append(_raw_tuple(e.read_raw(0)))
        return out

    name, val = switch.split("=")
    monkeypatch.delenv(name, raising=False)
    want = run()
    monkeypatch.setenv(name, val)
    got = run()
    assert len(want) == len(got)
    for a, b in zip(want, got):
        if isinstance(a, tuple):
            assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:])), switch
        else:
            assert np.array_equal(a, b), switch


def test_eight_wave_tiles_on_maps_that_do_not_tile(blob, monkeypatch):
    """The stride-2 layers' 8-wave workgroup on a 416 net: 52 x 52, 26 x 26 and 13 x 13 outputs take the row-run block
    scheme (no 2-D block divides them), last blocks are partial, the batch's last image group is short (three frames)."""
    imgs = [frames.synthetic_frame(80 + i) for i in range(3)]
    heads = {}
    for mode in ("IRMV_FORCE_W8", "IRMV_FORCE_NT8", "IRMV_NO_W8"):
        for m in ("IRMV_FORCE_W8", "IRMV_FORCE_NT8", "IRMV_NO_W8", "IRMV_NO_CM"):
            monkeypatch.delenv(m, raising=False)
        monkeypatch.setenv(mode, "1")
        if mode == "IRMV_NO_W8":
            monkeypatch.setenv("IRMV_NO_CM", "1")
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=3, num_streams=1, net_size=416) as e:
            names = [st["name"] for st in e.profile(0, 3)]
            if mode == "IRMV_FORCE_NT8":
                assert sum("_nt8" in n for n in names) == 3, names      # the 128-channel workgroup on model.5 / 7 / 19
            else:
                assert (sum("_w8" in n for n in names) == 5) == (mode == "IRMV_FORCE_W8"), (mode, names)
            for s, im in enumerate(imgs):
                _load(e, s, im)
            e.submit(0, 3); e.wait()
            heads[mode] = [e.read_head(s).copy() for s in range(3)] + [e.read_tap(t, 2).copy() for t in ("3", "5", "7", "16", "19")]
    for mode in ("IRMV_FORCE_W8", "IRMV_FORCE_NT8"):
        for a, b in zip(heads[mode], heads["IRMV_NO_W8"]):
            assert np.array_equal(a, b), mode


def test_four_steps_ahead_staging_is_bitwise_the_others(blob, monkeypatch):
    """The LDS family's staging four (image, chunk) steps ahead (`_p4`: the 128-channel layers of a lone frame, all of a
    workgroup's steps in flight) on every layer that has the tile, for single-frame and four-frame steps at 640 (20 x 20 and
    40 x 40 maps, stride 1 and 2) and 416 (13 x 13 / 26 x 26: row runs with partial last blocks): heads and detections are
    those of an engine without it (IRMV_NO_PF4=1), bit for bit."""
    for net in (640, 416):
        outs = {}
        for mode in ("IRMV_FORCE_PF4", "IRMV_NO_PF4"):
            monkeypatch.delenv("IRMV_FORCE_PF4", raising=False)
            monkeypatch.delenv("IRMV_NO_PF4", raising=False)
            monkeypatch.setenv(mode, "1")
            with YoloEngine(None, (1280, 1024), weights_blob=blob, net_size=net, num_slots=4) as e:
                n1 = [st["name"] for st in e.profile(0, 1)]
                n4 = [st["name"] for st in e.profile(0, 4)]
                if mode == "IRMV_FORCE_PF4":
                    assert sum(n.endswith("_p4") for n in n1) >= 6 and sum(n.endswith("_p4") for n in n4) >= 6, (n1, n4)
                else:
                    assert not any("_p4" in n for n in n1 + n4)
                for s in range(4):
                    _load(e, s, frames.synthetic_frame(40 + s))
                e.submit(0, 4); e.wait()
                res = [e.read_head(s).copy() for s in range(4)] + [_raw_tuple(e.read_raw(s)) for s in range(4)]
                e.detect(2)
                res += [e.read_head(2).copy(), _raw_tuple(e.read_raw(2))]
                outs[mode] = res
        for a, b in zip(outs["IRMV_FORCE_PF4"], outs["IRMV_NO_PF4"]):
            if isinstance(a, tuple):
                assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:])), net
            else:
                assert np.array_equal(a, b), net


def test_merged_head_first_stage_is_bitwise_the_separate_convs(blob, frame0, monkeypatch):
    """Single-frame engines run the three first-stage Detect convs of a level as one conv (weights concatenated along
    cout, second-stage convs reading channel slices): same bits as the separate convs, for pose and bbox-only models."""
    from irmv_detection_amd import weights
    for b in (blob, weights.synthetic_blob(0, nk=0)):
        heads = []
        for mode in ("1", "0"):
            monkeypatch.setenv("IRMV_MERGE_HEAD0", mode)
            with YoloEngine(None, (1280, 1024), weights_blob=b, point_source=capi.POINTS_AUTO) as e:
                layers = [st["layer"] for st in e.profile(0, 1)]
                assert any(l.startswith("model.22.s0.") for l in layers) == (mode == "1")
                assert len(layers) < 60 if mode == "1" else True
                _load(e, 0, frame0)
                e.detect()
                heads.append((e.read_head(0).copy(), e.read_tap("22.cv2.0.1", 0).copy(), e.read_tap("22.cv3.2.1", 0).copy()))
        for x, y in zip(*heads):
            assert np.array_equal(x, y)
    monkeypatch.delenv("IRMV_MERGE_HEAD0")
    with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=8) as e:      # batched engines keep the separate convs
        assert not any(st["layer"].startswith("model.22.s0.") for st in e.profile(0, 4))


def test_single_frame_bottleneck_kernels_are_bitwise_the_layers(blob, monkeypatch):
    """Steps of ONE frame run the 64-channel Bottlenecks of model.6 / 12 / 18 as one launch each, the block's last one together
    with its cv2 (k_bneck.hip: 14 launches -> 4).  Same rounding points and K order as the per-layer kernels: block outputs,
    head and detections are theirs bit for bit -- 640 net (40 x 40 maps: 25 whole tiles) and 416 net (26 x 26: partial tiles
    on two sides), pose and bbox-only models, detect() and pipelined single-frame steps; a batched step of the same engine
    keeps the layers and gives the same bits; read-backs of the blocks' internal tensors still work."""
    from irmv_detection_amd import weights
    img = frames.synthetic_frame(5)
    for b, net in ((blob, 640), (blob, 416), (weights.synthetic_blob(0, nk=0), 640)):
        outs = []
        for mode in ("1", "0"):
            monkeypatch.setenv("IRMV_BNECK64", mode)
            with YoloEngine(None, (1280, 1024), weights_blob=b, net_size=net, num_slots=3, point_source=capi.POINTS_AUTO) as e:
                names = [st["name"] for st in e.profile(0, 1)]
                assert (sum(n == "bneck64_a" for n in names), sum(n == "bneck64_b" for n in names)) == ((1, 3) if mode == "1" else (0, 0)), names
                layers = [st["layer"] for st in e.profile(0, 1)]
                assert any(l == "model.12.m.0.cv1" for l in layers) == (mode == "0")
                assert not any(n.startswith("bneck64") for n in (st["name"] for st in e.profile(0, 3)))   # a batched step keeps the layers
                _load(e, 0, img)
                e.detect(0)
                got = [e.read_head(0).copy()] + [e.read_tap(t, 0).copy() for t in ("6", "12", "18")] + [_raw_tuple(e.read_raw(0))]
                got += [e.read_tap(t, 0).copy() for t in ("model.6.cat", "model.12.cat", "model.18.tmp")]   # internal tensors: materialised by the layer ops
                for s in range(3):
                    _load(e, s, img)
                e.submit(0, 3); e.wait()                 # the batched step (layer kernels) ...
                got.append(e.read_head(2).copy())
                assert np.array_equal(got[0], got[-1])   # ... equals the single-frame step of the same engine
                e.submit(1, 1, async_upload=True); e.wait_slots(1, 1)
                got.append(_raw_tuple(e.read_raw(1)))
                outs.append(got)
        for x, y in zip(*outs):
            if isinstance(x, tuple):
                assert x[0] == y[0] and all(np.array_equal(p, q) for p, q in zip(x[1:], y[1:])), net
            else:
                assert np.array_equal(x, y), net
    monkeypatch.delenv("IRMV_BNECK64")


def test_a_stream_share_submitted_alone_is_its_sub_batch_of_the_whole_step(blob):
    """A multi-slot step is one captured graph per compute stream; a submit of exactly one stream's share of the slots is that
    sub-batch (same graph, same stream): feeding the two shares of a six-slot engine separately, in either order and with the
    second one late, gives the whole step's heads and detections (scripts/probes/phase_probe.py feeds a benchmark this way)."""
    with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=6) as e:
        for s in range(6):
            _load(e, s, frames.synthetic_frame(40 + s))
        e.submit(0, 6); e.wait()
        want = [e.read_head(s).copy() for s in range(6)] + [_raw_tuple(e.read_raw(s)) for s in range(6)]
        for order in ((0, 3), (3, 0)):
            e.submit(order[0], 3); e.wait_slots(order[0], 3)
            e.submit(order[1], 3); e.wait()
            got = [e.read_head(s).copy() for s in range(6)] + [_raw_tuple(e.read_raw(s)) for s in range(6)]
            for a, b in zip(want, got):
                if isinstance(a, tuple):
                    assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
                else:
                    assert np.array_equal(a, b)
        names = [st["name"] for st in e.profile(3, 3)]
        assert len(names) > 10


def test_keypoint_branch_kernel_is_bitwise_the_layers(blob, monkeypatch):
    """Engines that keep the Detect branches' convs apart (every batched engine) run a level's keypoint branch -- 3x3 (Cin -> 16),
    3x3 (16 -> 16), final 1x1 -- as one launch (k_kpt.hip: 6 launches -> 3).  Same rounding points and K order as the layer
    kernels: head, detections and keypoints are theirs bit for bit at a 640 net (80 / 40 / 20 maps: whole 10 x 10 tiles) and a
    416 net (52 / 26 / 13: partial tiles on two sides), in a batched step and a step of one frame of the same engine;
    read-backs of the branch's internal tensors still run the layers.  A 320 net has a 10 x 10 top level: one tile per image."""
    for net in (640, 416, 320):
        outs = []
        for mode in ("1", "0"):
            monkeypatch.setenv("IRMV_KPT3", mode)
            with YoloEngine(None, (1280, 1024), weights_blob=blob, net_size=net, num_slots=6, point_source=capi.POINTS_AUTO) as e:
                prof = e.profile(0, 6)
                names = [st["name"] for st in prof]
                assert sorted(n for n in names if n.startswith("kpt3_c")) == (["kpt3_c128", "kpt3_c256", "kpt3_c64"] if mode == "1" else []), names
                assert any(st["layer"] == "model.22.cv4.0.0" for st in prof) == (mode == "0")
                for s in range(6):
                    _load(e, s, frames.synthetic_frame(5 + s))
                e.submit(0, 6); e.wait()                     # (six slots: two streams of three -- a batched engine; up to four slots are single-frame engines)
                got = [e.read_head(s).copy() for s in range(6)] + [_raw_tuple(e.read_raw(s)) for s in range(6)]
                got += [e.read_tap(t, 1).copy() for t in ("22.cv4.0.0", "22.cv4.1.1", "22.cv4.2.0")]   # internal tensors: materialised by the layer ops
                e.detect(1)                                  # a step of one frame on the same engine ...
                got.append(e.read_head(1).copy())
                assert np.array_equal(got[1], got[-1])       # ... equals the batched step
                outs.append(got)
        for x, y in zip(*outs):
            if isinstance(x, tuple):
                assert x[0] == y[0] and all(np.array_equal(p, q) for p, q in zip(x[1:], y[1:])), net
            else:
                assert np.array_equal(x, y), net
        assert np.abs(outs[0][0][:, 80:88]).max() > 0        # (the keypoint channels are populated)
    # A workgroup keeps its tile for several consecutive images once a launch has >= 2048 tile-images (80 x 80 level: from 32
    # frames per stream), and deals image groups to XCDs in eights: 74 slots = two streams of 37 frames -> two images per
    # workgroup, 19 groups (a last group of one image, three groups behind the whole eights); with and without the XCD order.
    heads = {}
    for mode, xcd in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("IRMV_KPT3", mode)
        monkeypatch.setenv("IRMV_XCD_IMAGES", xcd)
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=74, point_source=capi.POINTS_AUTO) as e:
            for s in range(74):
                _load(e, s, frames.synthetic_frame(200 + (s % 9)))
            e.submit(0, 74); e.wait()
            heads[(mode, xcd)] = [e.read_head(s)[:, 80:88].copy() for s in range(74)]
    for s in range(74):
        assert np.array_equal(heads[("1", "1")][s], heads[("0", "1")][s]), s
        assert np.array_equal(heads[("1", "0")][s], heads[("0", "1")][s]), s
        assert np.array_equal(heads[("1", "1")][s], heads[("1", "1")][s % 9]), s     # (slots s and s % 9 hold the same frame)
    monkeypatch.delenv("IRMV_KPT3")
    monkeypatch.delenv("IRMV_XCD_IMAGES")


def test_grouped_detect_launches_are_bitwise_the_separate_convs(blob, monkeypatch):
    """Single-frame engines run the independent Detect-branch convs of the three levels as one launch per stage
    (conv3x3_lds_multi / conv_mfma_multi): same kernels on the same operands, so same bits -- head tensor, the second-stage
    activations, candidates and detections -- for pose and bbox-only models, at 640 and 416."""
    from irmv_detection_amd import weights
    for b, net in ((blob, 640), (weights.synthetic_blob(0, nk=0), 640), (blob, 416)):
        outs = []
        for mode in ("1", "0"):
            monkeypatch.setenv("IRMV_GROUP_HEAD", mode)
            with YoloEngine(None, (1280, 1024), weights_blob=b, net_size=net, point_source=capi.POINTS_AUTO) as e:
                names = [st["name"] for st in e.profile(0, 1)]
                assert any(n.startswith("head_s0_lds") for n in names) == (mode == "1")
                assert any(n.startswith("head_s1+1x1_lds") for n in names) == (mode == "1")
                if mode == "1":
                    assert len(names) <= 46          # 40 with all four groups; the keypoint pair forms only when it times faster
                _load(e, 0, frames.synthetic_frame(3))
                e.detect()
                raw = e.read_raw(0)
                outs.append((e.read_head(0).copy(), e.read_tap("22.cv2.1.1", 0).copy(), e.read_tap("22.cv3.0.1", 0).copy(),
                             raw["boxes"].copy(), raw["scores"].copy(), raw["anchors"].copy(), np.array([raw["n_candidates"], raw["num_dets"]])))
        for i, (x, y) in enumerate(zip(*outs)):
            assert np.array_equal(x, y), (net, i, names)
        assert net != 640 or outs[0][-1][1] > 0          # (frame 3 has candidates at 640; none at 416 with the 640-calibrated weights)
    monkeypatch.delenv("IRMV_GROUP_HEAD")


def test_a_garbage_candidate_counter_costs_one_step_and_no_fault(blob):
    """Round 1's GPU memory fault was a kernel trusting a per-frame counter that held garbage.  The counter exists again
    (candidates are appended by the class-branch conv epilogues): whatever it holds when a step starts, that step stays
    inside its buffers and resets it, and the following step is correct -- single-frame engines and batched steps."""
    for slots in (1, 4):
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=slots) as e:
            for s in range(slots):
                _load(e, s, frames.synthetic_frame(s))
            e.submit(0, slots)
            e.wait()
            want = [e.read_raw(s) for s in range(slots)]
            assert all(w["num_dets"] > 0 for w in want)
            for value in (0x7FFFFFFF, -7, 50000, 8400 * 14 + 5, 1):
                e.debug_poke_candidate_counts(value)
                e.submit(0, slots)            # may report anything -- but must neither fault nor leave the counter dirty
                e.wait()
                e.submit(0, slots)
                e.wait()
                for s in range(slots):
                    got = e.read_raw(s)
                    assert got["n_candidates"] == want[s]["n_candidates"] and got["num_dets"] == want[s]["num_dets"], (slots, value, s)
                    assert np.array_equal(got["boxes"], want[s]["boxes"]) and np.array_equal(got["anchors"], want[s]["anchors"])
                    assert np.array_equal(got["kpts"], want[s]["kpts"])


def _bench_tune_cache(tmp_path, monkeypatch):
    """Seed the autotuner exactly as bench.py does (its own copy of profiles/*_tune_cache.txt)."""
    import os, shutil
    from conftest import ROOT
    import bench
    seed = bench.tune_cache_seed()
    mine = tmp_path / "tune.txt"
    shutil.copyfile(seed, mine)
    monkeypatch.setenv("IRMV_TUNE_CACHE", str(mine))
    monkeypatch.setenv("IRMV_TUNE_WARN", "1")


def test_benchmarked_configuration_is_bitwise_the_single_slot_engine(blob, onet, tmp_path, monkeypatch):
    """bench.py's configuration -- 256 slots replayed as two concurrent 128-frame hipGraphs with the tiles of the
    committed tune cache (images-per-workgroup pipelines, resident-weight and multi-block kernels, 128-frame grids of the
    front / C2f / SPPF / decode kernels) --
    under assertions: every slot's head and detections bitwise equal a 1-slot engine's detect() on the same frame, and
    the first / last slots of every graph within tolerance of the fp32 oracle."""
    B = 256
    imgs = [frames.synthetic_frame(200 + i) for i in range(B)]
    _bench_tune_cache(tmp_path, monkeypatch)
    with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=B) as e:
        assert e.num_streams == 2
        names = [st["name"] for st in e.profile(0, B // 2)]
        assert any("_i4" in n or "_i2" in n for n in names), names       # several images per workgroup are in play
        assert any("_wres" in n for n in names) and any("pw_n2" in n for n in names), names   # ... and the round-3 kernels
        # the committed table's special tiles are REPLAYED, not silently re-tuned (a cached tile that fails validation is)
        import bench
        flags = [int(l.split()[-2]) for l in open(bench.tune_cache_seed()) if "|n128|" in l]
        assert any(f & (64 | 128) for f in flags) == any("_cm" in n for n in names), names
        assert any(f & 256 for f in flags) == any("_w8" in n for n in names), names
        for s in range(B):
            _load(e, s, imgs[s])
        e.submit(0, B, h2d=True)
        e.wait()
        heads = [e.read_head(s).copy() for s in range(B)]
        raws = [e.read_raw(s) for s in range(B)]
        # second replay from HBM-resident frames, as the timed loop does
        e.submit(0, B, h2d=False)
        e.wait()
        for s in (0, 63, 127, 128, 191, 255):
            assert np.array_equal(e.read_head(s), heads[s])
    monkeypatch.delenv("IRMV_TUNE_CACHE")
    with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=1) as one:
        for s in range(B):
            _load(one, 0, imgs[s])
            one.detect(0)
            assert np.array_equal(one.read_head(0), heads[s]), s
            r = one.read_raw(0)
            assert r["num_dets"] == raws[s]["num_dets"] and np.array_equal(r["anchors"], raws[s]["anchors"]), s
            assert np.array_equal(r["boxes"], raws[s]["boxes"]) and np.array_equal(r["scores"], raws[s]["scores"]), s
            assert np.array_equal(r["kpts"], raws[s]["kpts"]), s
    assert not np.array_equal(heads[0], heads[1])
    for s in (0, 63, 127, 128, 191, 255):
        ho = onet.forward(oracle.preprocess(imgs[s], 640))
        assert _report_head(f"benchmarked configuration, slot {s}", float(np.abs(heads[s] - ho).max())) <= HEAD_TOL, s
        exp = oracle.decode_nms(heads[s], 640, 14, 8)
        assert raws[s]["num_dets"] == exp["num_dets"] and np.array_equal(raws[s]["anchors"], exp["anchors"])
        assert np.array_equal(raws[s]["boxes"], exp["boxes"])


def test_synchronous_launch_form_is_reported_and_forced(blob, frame0, monkeypatch):
    """A synchronous single-frame step has two launch forms (one hipGraph replay / kernel by kernel behind the upload): the engine
    times both at creation and reports its choice (irmv_engine_sync_launch); IRMV_SYNC_LAUNCH forces one.  Same detections."""
    got = {}
    for forced in ("graph", "eager", None):
        if forced:
            monkeypatch.setenv("IRMV_SYNC_LAUNCH", forced)
        else:
            monkeypatch.delenv("IRMV_SYNC_LAUNCH", raising=False)
        with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=3) as e:
            assert e.sync_launch in ("graph", "eager")
            if forced:
                assert e.sync_launch == forced
            _load(e, 1, frame0)
            e.detect(1)
            e.detect(1)
            got[forced] = (_raw_tuple(e.read_raw(1)), e.read_head(1).copy())
    for k in ("eager", None):
        assert got[k][0][0] == got["graph"][0][0] and all(np.array_equal(x, y) for x, y in zip(got[k][0][1:], got["graph"][0][1:]))
        assert np.array_equal(got[k][1], got["graph"][1])


def test_two_engines_with_independent_lifetimes(blob, frame0):
    """The reference node owns three engines (src/irm_detector.cpp:35-38): create A, create and destroy B, then A must
    still replay its graphs (upload included) and give the same bits."""
    a = YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=4)
    for s in range(4):
        _load(a, s, frames.synthetic_frame(s))
    a.submit(0, 4); a.wait()
    h0 = a.read_head(3).copy()
    b = YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=1)
    _load(b, 0, frame0)
    b.detect()
    b.close()
    a.submit(0, 4, h2d=True); a.wait()
    assert np.array_equal(a.read_head(3), h0)
    a.submit(0, 4, h2d=False); a.wait()
    assert np.array_equal(a.read_head(3), h0)
    a.close()


def test_pipelined_slots_overlap_and_match_detect(blob):
    """a13: slot n+1 is submitted (upload on the copy stream) while slot n is in flight; wait_slots() takes each result
    as it completes.  Same bits as synchronous detect()."""
    imgs = [frames.synthetic_frame(300 + i) for i in range(6)]
    with YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=3) as e:
        want = []
        for im in imgs:
            _load(e, 0, im)
            e.detect(0)
            want.append(e.read_raw(0))
        got = []
        _load(e, 0, imgs[0]); e.submit(0, 1, async_upload=True)
        for i in range(6):
            if i + 1 < 6:
                _load(e, (i + 1) % 3, imgs[i + 1])
                e.submit((i + 1) % 3, 1, async_upload=True)   # uploads while slot i % 3 is in flight
            e.wait_slots(i % 3, 1)
            got.append(e.read_raw(i % 3))
        e.wait()
    for w, g in zip(want, got):
        assert w["num_dets"] == g["num_dets"] and np.array_equal(w["boxes"], g["boxes"]) and np.array_equal(w["scores"], g["scores"])


@pytest.mark.parametrize("size,net,mode,rot,swap,fused", [
    ((1280, 1024), 640, 0, True, False, True),     # reference configuration
    ((1280, 1024), 640, 0, False, True, True),
    ((1280, 1024), 640, 1, True, False, True),     # letterbox: pad rows inside tiles
    ((640, 640), 640, 0, True, False, True),       # BASELINE configs[1]
    ((1280, 1024), 416, 0, True, False, True),     # 104 x 104 output: partial tiles
    ((1920, 1200), 640, 1, True, False, True),     # 3x down-scale: 87 KB source region per tile, one workgroup per CU
    ((1024, 768), 640, 0, True, False, True),
    ((1600, 1200), 640, 1, True, False, True),
    ((644, 480), 640, 0, False, False, True),      # up-scale in x: a tile's source region is narrower than the tile
    ((1280, 720), 640, 1, False, True, True),      # letterbox bands above and below
    ((800, 600), 416, 1, True, False, True),
    ((1276, 1280), 640, 1, True, False, True),     # 2 : 1 columns behind ONE pad column: the direct tiles' column pairs start inside the tile (step -2)
    ((1276, 1280), 640, 1, False, True, True),     # ... and step +2
    ((832, 832), 416, 0, True, False, True),       # 2 : 1 at a 416 net: the 8 x 16 direct tiles with a partial last tile column (104 = 6.5 x 16)
    ((832, 600), 416, 1, False, True, True),       # ... with letterbox bands (rows 58 .. 357 have a source)
    ((641, 479), 640, 0, True, False, False),      # width not a multiple of 4: falls back to the three kernels
    ((4096, 3000), 640, 0, True, False, False),    # tile's source region larger than the LDS stage: falls back
])
def test_fused_kernels_are_bitwise_identical(blob, monkeypatch, size, net, mode, rot, swap, fused):
    """preprocess + model.0 + model.1 in one kernel (k_front.hip) and model.2 in one kernel (k_c2f.hip) == the
    seven separate kernels, bit for bit; likewise the Detect branches' final 1x1 convs computed in the epilogue of the
    preceding 3x3 (k_conv.hip, N2 > 0)."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (size[1], size[0], 3), dtype=np.uint8)
    img[: size[1] // 2] = frames.synthetic_frame(3, size[0], size[1])[: size[1] // 2]
    got, kernels = [], []
    for env in ("1", "0"):
        monkeypatch.setenv("IRMV_FUSED_FRONT", env)
        monkeypatch.setenv("IRMV_FUSED_C2F", env)
        monkeypatch.setenv("IRMV_FUSED_HEAD", env)
        with YoloEngine(None, size, weights_blob=blob, net_size=net, resize_mode=mode, rotate180=rot, swap_rb=swap) as e:
            names = [st["name"] for st in e.profile(0, 1)]
            kernels.append([(st["layer"], st["name"]) for st in e.profile(0, 1) if "s2" in st["name"] or "c2f" in st["name"]])
            assert ("front_fused" in names) == (env == "1" and fused)
            assert ("c2f2_fused" in names) == (env == "1")
            assert ("c2f32_ab" in names and "c2f32_a" in names and "c2f32_b" in names) == (env == "1")   # model.15; model.4
            assert any("+1x1" in n for n in names) == (env == "1")              # Detect finals inside the 3x3 epilogue (single launches or the grouped one)
            _load(e, 0, img)
            e.detect()
            got.append((e.read_tap("1", 0).copy(), e.read_head(0).copy(), e.read_input(0).copy(), e.read_tap("0", 0).copy(),
                        e.read_tap("2", 0).copy(), e.read_tap("model.2.cat", 0).copy(), e.read_tap("22.cv2.0.1", 0).copy(),
                        e.read_tap("22.cv3.2.1", 0).copy(), e.read_tap("4", 0).copy(), e.read_tap("15", 0).copy(),
                        e.read_tap("model.4.cat", 0).copy(), e.read_tap("model.15.cat", 0).copy(), e.read_tap("model.4.tmp", 0).copy()))
    labels = ("1", "head", "input", "0", "2", "model.2.cat", "22.cv2.0.1", "22.cv3.2.1", "4", "15", "model.4.cat", "model.15.cat", "model.4.tmp")
    differ = [lb for lb, a, b in zip(labels, got[0], got[1]) if not np.array_equal(a, b)]
    where = ""
    for lb, a, b in zip(labels, got[0], got[1]):
        if lb == "head" and not np.array_equal(a, b):
            d = np.abs(a - b)
            where = f"head rows {np.nonzero(d.max(1) > 0)[0][:12].tolist()} cols {np.nonzero(d.max(0) > 0)[0][:24].tolist()} max {d.max()}"
    assert not differ, (differ, where, kernels)
    assert np.abs(got[0][0]).max() > 0.1


def test_frame_slots_sit_on_the_devices_numa_node():
    """Multi-GPU host side (SURVEY section 7 "hard parts"): the pinned frame slots are allocated and first touched on the host
    NUMA node closest to the engine's device; IRMV_NUMA=0 leaves placement to the OS.  In a child process WITHOUT torch: the
    test runner's torch puts the library on its bundled ROCm 7.0 runtime, which does not know hipDeviceAttributeHostNumaId
    (the engine then reports node -1 and places nothing -- also checked).  Where the box reports no node, or the page query
    is not permitted, the engine says so and nothing is asserted about pages."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from irmv_detection_amd import frames, weights\n"
            "from irmv_detection_amd.engine import YoloEngine\n"
            "assert 'torch' not in sys.modules\n"
            "with YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=3) as e:\n"
            "    node, placed = e.numa_node, e.numa_placed\n"
            "    pages = [e.src_page_node(s) for s in range(3)]\n"
            "    print(f'numa: device node {node}, placed {placed}, slot pages on nodes {pages}')\n"
            "    assert node >= -1\n"
            "    if node >= 0 and placed and min(pages) >= 0:\n"
            "        assert all(p == node for p in pages), (node, pages)\n"
            "        print('numa: pages verified on the device node')\n"
            "    e.get_src_image_buffer(0)[:] = frames.synthetic_frame(0)\n"
            "    assert len(e.detect(0)) > 0\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr[-2000:])
    assert out.returncode == 0
    # in THIS process the engine must come up whatever runtime serves it
    from irmv_detection_amd import weights as W
    with YoloEngine(None, (1280, 1024), weights_blob=W.synthetic_blob(0), num_slots=1) as e:
        assert e.numa_node >= -1 and (e.numa_placed is False or e.numa_node >= 0)


def test_stride2_implementations_are_bitwise_identical(blob, monkeypatch):
    """The stride-2 layers of the LDS 3x3 family have three implementations the autotuner may pick between -- the LDS kernel,
    the direct kernel walking K chunk-major on the family's weights, its deep-prefetch form (single-frame steps) -- and
    IRMV_FORCE_S2 pins one of them: same taps, same head, bit for bit, at a 416 net (52 / 26 / 13 maps: partial tiles)."""
    img = np.random.default_rng(11).integers(0, 256, (1024, 1280, 3), dtype=np.uint8)
    res = {}
    for kind in ("lds", "ct", "deep"):
        monkeypatch.setenv("IRMV_FORCE_S2", kind)
        with YoloEngine(None, (1280, 1024), weights_blob=blob, net_size=416) as e:
            names = [s["name"] for s in e.profile(0, 1) if "s2" in s["name"]]
            assert len(names) >= 4, names
            if kind == "lds":
                assert all("_lds_" in n for n in names), names
            else:
                assert sum(n.endswith("_deep_ct" if kind == "deep" else "_ct") for n in names) >= 3, (kind, names)   # (model.3's Cin = 32 layer and model.1 have no LDS-family form to stand in for)
            _load(e, 0, img)
            e.detect(0)
            res[kind] = [e.read_head(0).copy()] + [e.read_tap(t, 0).copy() for t in ("3", "5", "7", "16", "19")]
    monkeypatch.delenv("IRMV_FORCE_S2")
    for kind in ("ct", "deep"):
        for a, b in zip(res["lds"], res[kind]):
            assert np.array_equal(a, b), kind


def _clustered_head(rng, n_cluster, n_tail):
    """A head whose best `n_cluster` candidates are one class on neighbouring anchors with wide, heavily overlapping boxes (the
    greedy walk keeps a few dozen of them), followed by `n_tail` weaker candidates of all classes spread over the image: the
    hundredth survivor sits far behind the thousandth candidate."""
    head = np.zeros((8400, 86), np.float32)
    head[:, 64:78] = -20.0
    head[:, :64] = 0.05 * rng.standard_normal((8400, 64))             # DFL ~ uniform: boxes of ~ +-7.5 bins around the anchor
    head[:, 78:] = 0.25 + 0.3 * rng.standard_normal((8400, 8))
    head[:n_cluster, 64 + 3] = 3.0 + rng.permutation(n_cluster).astype(np.float32) * 1e-3   # distinct logits, all above the tail's
    tail = rng.choice(np.arange(n_cluster, 8400), n_tail, replace=False)
    head[tail, 64 + rng.integers(0, 14, n_tail)] = rng.uniform(-1.0, 2.5, n_tail).astype(np.float32)
    return head


@pytest.mark.parametrize("n_cluster,n_tail", [(1600, 600), (1100, 4000), (3000, 2000), (700, 500)])
def test_crowded_frames_prefilter_is_exact_also_when_it_has_to_start_over(blob, monkeypatch, n_cluster, n_tail):
    """nms_pnp_kernel on frames with more than 1024 candidates first walks the best <= 1024 only (exact radix threshold) and
    starts over on the whole list when that walk neither filled max_det nor reached pre_nms_cap.  Heads built so that it
    MUST start over (a cluster of 1100 .. 3000 mutually suppressing candidates in front), one where the cluster is shorter
    than the head of the list, default caps and a small pre_nms_cap: survivors = the oracle's, bit for bit, and the same
    with the prefilter switched off (IRMV_NMS_PREFILTER=0)."""
    rng = np.random.default_rng(n_cluster + n_tail)
    head = _clustered_head(rng, n_cluster, n_tail)
    got = {}
    for pf in ("1", "0"):
        monkeypatch.setenv("IRMV_NMS_PREFILTER", pf)
        for cap, md in ((4096, 100), (900, 60), (2048, 256)):
            with YoloEngine(None, (1280, 1024), weights_blob=blob, pre_nms_cap=cap, max_det=md) as e:
                raw = _assert_post_exact(e, head, max_det=md, pre_nms_cap=cap)
                got[(pf, cap, md)] = _raw_tuple(raw)
                if cap == 4096 and md == 100:
                    assert raw["n_candidates"] > 1100 and 20 < raw["num_dets"] <= 100
    for cap, md in ((4096, 100), (900, 60), (2048, 256)):
        a, b = got[("1", cap, md)], got[("0", cap, md)]
        assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))


def test_crowded_camera_frames_end_to_end_prefilter_on_and_off(blob, monkeypatch):
    """The same through the whole step (candidates emitted by the class-branch conv epilogues, boxes decoded inside the NMS
    kernel for the selected candidates only): 640 x 640 crops of the synthetic camera frames carry 2 300 .. 4 900 candidates
    (BASELINE configs[1]'s frames, bench.py `config1`); detections = the oracle's on the engine's own head, prefilter on and off."""
    out = {}
    for pf in ("1", "0"):
        monkeypatch.setenv("IRMV_NMS_PREFILTER", pf)
        with YoloEngine(None, (640, 640), weights_blob=blob, num_slots=1) as e:
            for fi in (1, 2, 7):
                img = np.ascontiguousarray(frames.synthetic_frame(fi)[:640, :640])
                _load(e, 0, img)
                e.detect(0)
                raw, head = e.read_raw(0), e.read_head(0)
                exp = oracle.decode_nms(head, 640, 14, 8)
                assert raw["n_candidates"] == exp["n_candidates"] and raw["n_candidates"] > 2000, (fi, raw["n_candidates"])
                assert raw["num_dets"] == exp["num_dets"] and np.array_equal(raw["anchors"], exp["anchors"]) and np.array_equal(raw["classes"], exp["classes"])
                assert np.array_equal(raw["boxes"], exp["boxes"]) and np.array_equal(raw["scores"], exp["scores"]) and np.array_equal(raw["kpts"], exp["kpts"])
                out[(pf, fi)] = _raw_tuple(raw)
    for fi in (1, 2, 7):
        a, b = out[("1", fi)], out[("0", fi)]
        assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))


@pytest.mark.parametrize("switch", ["IRMV_SPLIT_SCAN=0", "IRMV_EMIT_SCAN=0"])
def test_crowded_frames_with_the_other_candidate_sources(blob, monkeypatch, switch):
    """The first-walk-on-the-head-of-the-list path of nms_pnp_kernel takes its keys from three places: the class-branch conv
    epilogues (default), scan_decode_kernel (IRMV_EMIT_SCAN=0, and every run_post) and the kernel's own scan
    (IRMV_SPLIT_SCAN=0: no counters, keys also mirrored to the global list the start-over reloads from).  A head that forces
    the start-over and a 4 900-candidate camera crop, each against the oracle."""
    name, val = switch.split("=")
    monkeypatch.setenv(name, val)
    head = _clustered_head(np.random.default_rng(5), 1600, 600)
    with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
        raw = _assert_post_exact(e, head)
        assert raw["n_candidates"] > 1600
    with YoloEngine(None, (640, 640), weights_blob=blob) as e:
        _load(e, 0, np.ascontiguousarray(frames.synthetic_frame(1)[:640, :640]))
        e.detect(0)
        raw, hd = e.read_raw(0), e.read_head(0)
        exp = oracle.decode_nms(hd, 640, 14, 8)
        assert raw["n_candidates"] == exp["n_candidates"] > 4000
        assert raw["num_dets"] == exp["num_dets"] and np.array_equal(raw["anchors"], exp["anchors"]) and np.array_equal(raw["boxes"], exp["boxes"])
        assert np.array_equal(raw["scores"], exp["scores"]) and np.array_equal(raw["kpts"], exp["kpts"])
