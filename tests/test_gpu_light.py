"""GPU parity of the classical armor-point extraction through the C ABI:
irmv_engine_extract_armors and the bbox-only-model step, against oracle/orc_light.c."""
import numpy as np
import pytest

from conftest import D_REF, K_REF
from irmv_detection_amd import capi, frames, weights
from irmv_detection_amd.engine import YoloEngine, bbox, ArmorClass
from oracle import oracle

pytestmark = pytest.mark.gpu


def _compare(armors, rot, boxes, params=None, tol=1e-4):
    n_valid = 0
    for a, b in zip(armors, boxes):
        assert not a.no_answer, b
        o = oracle.extract_armor(rot, b, params)
        assert a.valid == o["ok"], (b, a.valid, o)
        assert a.n_lights == o["n_lights"]
        if o["ok"]:
            n_valid += 1
            assert int(a.size) == o["size"]
            assert np.abs(a.image_points() - o["pts"]).max() <= tol
            p = oracle.solve_pnp_ippe(K_REF, D_REF, o["pts"], 0)
            assert a.pnp_ok == p["ok"]
            if p["ok"] and abs(p["err"][0] - p["err"][1]) > 1e-7:
                assert np.abs(a.rvec - p["rvec"]).max() <= 1e-6 and np.abs(a.tvec - p["tvec"]).max() <= 1e-6
    return n_valid


def test_extract_armors_on_synthetic_frames(blob):
    rng = np.random.default_rng(3)
    with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
        total_valid = 0
        for fi in (1, 2, 5, 7):
            frame = frames.synthetic_frame(fi)
            e.get_src_image_buffer()[:] = frame
            rot = oracle.rotate180(frame)
            # boxes around the bright structures plus random ones (also partly / fully outside the frame)
            ys, xs = np.where(rot.max(2) >= 200)
            boxes = []
            for _ in range(60):
                k = rng.integers(0, len(xs))
                w, h = rng.uniform(60, 420), rng.uniform(60, 300)
                cx, cy = xs[k] + rng.uniform(-80, 80), ys[k] + rng.uniform(-40, 40)
                boxes.append((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2))
            for _ in range(30):
                x, y = rng.uniform(-100, 1300), rng.uniform(-100, 1050)
                boxes.append((x, y, x + rng.uniform(1, 300), y + rng.uniform(1, 200)))
            boxes.append((10.2, 10.0, 10.9, 60.0))          # zero-width after truncation
            boxes.append((0, 0, 1280, 1024))                # the whole frame as one ROI
            boxes = np.array(boxes, np.float32)
            total_valid += _compare(e.extract_armors(boxes), rot, boxes)
        assert total_valid >= 10                             # the comparison is not vacuous


def test_scratch_exhaustion_is_reported_not_guessed(blob):
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (1024, 1280, 3), dtype=np.uint8)       # thousands of one-pixel contours
    with YoloEngine(None, (1280, 1024), weights_blob=blob, rotate180=False) as e:
        e.get_src_image_buffer()[:] = noise
        boxes = np.array([(0, 0, 1280, 1024), (100, 100, 140, 130)] + [(0, 0, 1280, 1024)] * 9, np.float32)
        arm = e.extract_armors(boxes)
        assert arm[0].no_answer and not arm[0].valid             # > 1024 contours in one ROI
        o = oracle.extract_armor(noise, boxes[1])
        assert not arm[1].no_answer and arm[1].valid == o["ok"] and arm[1].n_lights == o["n_lights"]
        assert all(a.no_answer for a in arm[2:])
        # the label pool (8 frame areas) is handed out in detection order: seven whole-frame ROIs and a small one
        # fit, the rest are dropped -- deterministically
        frame = frames.synthetic_frame(1)
        e.get_src_image_buffer()[:] = frame
        arm = e.extract_armors(boxes)
        assert [a.no_answer for a in arm] == [False] * 8 + [True] * 3
        _compare(arm[:8], frame, boxes[:8])


def test_parameters_and_unrotated_mode(blob):
    frame = frames.synthetic_frame(1)
    ys, xs = np.where(frame.max(2) >= 200)
    boxes = np.array([(xs.min() - 20, ys.min() - 20, xs.min() + 380, ys.min() + 260), (300, 300, 700, 620)], np.float32)
    with YoloEngine(None, (1280, 1024), weights_blob=blob, rotate180=False, binary_threshold=120, light_max_angle=10.0,
                    light_min_ratio=0.05, armor_center_distances=(0.5, 2.0, 2.0, 6.0)) as e:
        e.get_src_image_buffer()[:] = frame
        P = oracle.light_params(binary_threshold=120, light_max_angle=10.0, light_min_ratio=0.05, armor_min_small_center_distance=0.5,
                                armor_max_small_center_distance=2.0, armor_min_large_center_distance=2.0, armor_max_large_center_distance=6.0)
        _compare(e.extract_armors(boxes), frame, boxes, P)


def test_bbox_only_model_uses_classical_points(frame0):
    """A model without a keypoint head (what the reference ships): detect() -> GPU light extraction -> PnP."""
    blob = weights.synthetic_blob(0, nk=0)
    with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
        assert e.head_channels == 64 + 14
        e.get_src_image_buffer()[:] = frame0
        armors = e.detect_armors()
        rot = oracle.rotate180(frame0)
        assert len(armors) > 0
        boxes = np.array([a.bbox_xyxy for a in armors], np.float32)
        _compare(armors, rot, boxes)
        # the standalone entry gives the same answer as the in-step extraction
        again = e.extract_armors(boxes)
        for a, b in zip(armors, again):
            assert a.valid == b.valid and a.n_lights == b.n_lights and np.array_equal(a.image_points(), b.image_points())
    with pytest.raises(capi.IrmvError):
        YoloEngine(None, (1280, 1024), weights_blob=blob, point_source=capi.POINTS_KEYPOINT_HEAD)


def test_classical_points_can_be_forced_on_a_pose_model(blob, frame0):
    with YoloEngine(None, (1280, 1024), weights_blob=blob, point_source=capi.POINTS_CLASSICAL, num_slots=2) as e:
        for s in range(2):
            e.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
        e.submit(0, 2)
        e.wait()
        for s in range(2):
            arm = e.results(s)
            rot = oracle.rotate180(frames.synthetic_frame(s))
            _compare(arm, rot, np.array([a.bbox_xyxy for a in arm], np.float32))


def test_rm_test_jpg_armor_golden_on_gpu(blob, rm_test_image):
    """The armor in the reference's test image, through irmv_engine_extract_armors: the committed golden, bit for bit,
    and a plausible pose (a large armor ~1.3 m in front of the reference camera)."""
    import json
    from conftest import golden_path
    cases = json.load(open(golden_path("light_cases.json")))
    boxes = np.array([c["box"] for c in cases], np.float32)
    with YoloEngine(None, (1280, 1024), weights_blob=blob, rotate180=False, armor_size=capi.ARMOR_LARGE) as e:
        e.get_src_image_buffer()[:] = rm_test_image
        arm = e.extract_armors(boxes)
    for a, c in zip(arm, cases):
        assert not a.no_answer and a.valid == c["ok"] and a.n_lights == c["n_lights"], c["box"]
        if c["ok"]:
            assert int(a.size) == c["size"] and np.array_equal(a.image_points().ravel(), np.array(c["pts"], np.float32))
            assert a.pnp_ok and 0.5 < a.tvec[2] < 5.0 and abs(a.tvec[0]) < 1.0 and abs(a.tvec[1]) < 1.0
