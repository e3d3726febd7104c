"""Conv table, FLOP/param accounting and the .irmw blob (host logic, CPU)."""
import hashlib

import numpy as np

from irmv_detection_amd import arch, frames, weights


def test_param_and_flop_counts_match_survey():
    # 63 convs, 3 008 362 params, 8.096 GFLOP (bbox head only)
    L = arch.conv_specs(14, 0)
    assert len(L) == 63
    assert sum(s.n_params for s in L) == 3_008_362
    assert abs(arch.flops_per_frame(640, 14, 0) - 8.0956416e9) < 1
    # with the 4-keypoint head: 72 convs, 3 080 290 params, 8.343 GFLOP
    L = arch.conv_specs(14, 8)
    assert len(L) == 72
    assert sum(s.n_params for s in L) == 3_080_290
    assert abs(arch.flops_per_frame(640, 14, 8) - 8.3429376e9) < 1
    # nc = 80 sanity check against the public YOLOv8n figures (8.7 GFLOPs / 3.2 M)
    assert sum(s.n_params for s in arch.conv_specs(80, 0)) == 3_151_888
    assert arch.num_anchors(640) == 8400 and arch.num_anchors(416) == 3549


def test_class_table_matches_reference_enum():
    # reference include/irmv_detection/armor.hpp:7
    assert arch.ARMOR_CLASS_NAMES == ("B1", "B2", "B3", "B4", "B5", "BO", "BS", "R1", "R2", "R3", "R4", "R5", "RO", "RS", "UNKNOWN")
    from irmv_detection_amd.engine import ArmorClass
    assert [c.name for c in ArmorClass] == list(arch.ARMOR_CLASS_NAMES)


def test_blob_roundtrip_and_determinism(blob):
    hdr, layers = weights.parse_blob(blob)
    assert hdr == dict(nc=14, nk=8, reg_max=16, n_layers=72, dtype=weights.DTYPE_FP16, backbone=0)
    specs, tensors = weights.synthetic_tensors(0)
    for (sp, w, b), sp2, (w2, b2) in zip(layers, specs, tensors):
        assert sp == sp2 and np.array_equal(w, w2) and np.array_equal(b, b2)
    assert weights.synthetic_blob(0) == blob                 # same seed -> same bytes
    assert weights.synthetic_blob(1) != blob
    assert np.isfinite(np.frombuffer(blob, np.float16, 1000, 8192).astype(np.float32)).all()


def test_blob_hash_is_stable_across_machines(blob):
    # golden vectors under tests/golden/ were produced with exactly these bytes
    assert hashlib.sha256(blob).hexdigest() == open(__file__.replace("test_arch_weights.py", "golden/weights_seed0.sha256")).read().strip()


def test_model_path_rule():
    # reference src/yolo_engine.cpp:28-31: "<stem>.onnx" -> sibling compiled model
    assert weights.model_blob_path("/a/b/yolov7.onnx") == "/a/b/yolov7.irmw"


def test_synthetic_frames_are_deterministic():
    a, b = frames.synthetic_frame(3), frames.synthetic_frame(3)
    assert a.shape == (1024, 1280, 3) and a.dtype == np.uint8 and np.array_equal(a, b)
    assert not np.array_equal(a, frames.synthetic_frame(4))
    assert frames.synthetic_batch(0, 2, 64, 48).shape == (2, 48, 64, 3)
