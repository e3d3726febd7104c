"""The C++ facade (include/irmv_detection/*.hpp): code written against the
reference's class names compiles against it (CPU), the triple buffer hand-off is
correct (CPU), and the reference's test flow runs on the GPU with the same results
as the Python mirror (same C ABI underneath)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from irmv_detection_amd import _build

INC = os.path.join(ROOT, "include")
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "_bin")


ORACLE = os.path.join(ROOT, "oracle")


def _compile(src, out, link_hip, link_oracle=False):
    os.makedirs(BIN, exist_ok=True)
    cmd = ["g++", "-std=c++20", "-O2", "-pthread", "-Wall", "-I", INC, os.path.join(CPP, src), "-o", out]
    if link_hip:
        _build.build()
        cmd += ["-L", _build.LIB_DIR, "-lirmv_hip", f"-Wl,-rpath,{_build.LIB_DIR}", "-Wl,-rpath-link,/opt/rocm/lib"]
    if link_oracle:   # the checker, test binaries only
        from oracle import oracle
        oracle.build()
        cmd += ["-I", ORACLE, "-L", ORACLE, "-l:liboracle.so", f"-Wl,-rpath,{ORACLE}"]
    subprocess.check_call(cmd)
    return out


def test_triple_buffer_handoff_terminates_and_never_tears():
    exe = _compile("triple_buffer_test.cpp", os.path.join(BIN, "triple_buffer_test"), False)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "torn 0" in out.stdout and "none_pending 1" in out.stdout


def test_reference_style_code_compiles_against_the_facade():
    # built here so the binaries travel to the GPU box with the tree
    exe = _compile("yolo_test.cpp", os.path.join(BIN, "yolo_test"), True)
    assert os.path.exists(exe)
    exe = _compile("camera_stream_test.cpp", os.path.join(BIN, "camera_stream_test"), True)
    assert os.path.exists(exe)
    exe = _compile("irm_detector_core_test.cpp", os.path.join(BIN, "irm_detector_core_test"), True, link_oracle=True)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_reference_test_flow_on_gpu(tmp_path, blob, frame0):
    exe = os.path.join(BIN, "yolo_test")
    if not os.path.exists(exe):
        exe = _compile("yolo_test.cpp", exe, True)
    (tmp_path / "yolov7.irmw").write_bytes(blob)
    frame0.tofile(tmp_path / "frame.bin")
    # 30 runs x 10 iterations, the reference's own count (test/yolo_test.cpp:76-92)
    out = subprocess.run([exe, str(tmp_path / "yolov7.onnx"), str(tmp_path / "frame.bin"), "30"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    from irmv_detection_amd.engine import YoloEngine
    with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
        e.get_src_image_buffer()[:] = frame0
        bb = e.detect()
        arm = e.detect_armors()
    assert int(re.search(r"bboxes (\d+)", txt).group(1)) == len(bb)
    first = [float(v) for v in re.search(r"bbox 0 (\S+) (\S+) (\S+) (\S+) (\S+)", txt).groups()]
    assert np.allclose(first[:4], bb[0].xyxy, atol=1e-4) and abs(first[4] - bb[0].score) < 1e-5
    assert "rotated_ok 1" in txt
    m = re.search(r"classical armors (\d+) of (\d+) boxes, poses (\d+)", txt)
    boxes_xyxy = np.array([b.xyxy for b in bb], np.float32)
    with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
        e.get_src_image_buffer()[:] = frame0
        n_valid = sum(a.valid for a in e.extract_armors(boxes_xyxy))
    assert int(m.group(1)) == int(m.group(3)) == n_valid and int(m.group(2)) == len(bb)
    m = re.search(r"pnp ok (\d) fused_ok (\d) worst_diff (\S+) tvec (\S+) (\S+) (\S+)", txt)
    assert m.group(1) == m.group(2) == "1" and float(m.group(3)) < 1e-9        # standalone PnPSolver == fused PnP
    assert np.allclose([float(m.group(i)) for i in (4, 5, 6)], arm[0].tvec, atol=1e-8)
    assert re.search(r"detect_ms avg (\S+) max (\S+)", txt) and float(re.search(r"max (\S+) min", txt).group(1)) < 30.0


@pytest.mark.gpu
def test_camera_stream_330fps_through_the_triple_buffer(tmp_path, blob):
    """BASELINE configs[2]: paced 330 FPS synthetic camera -> pinned slots -> async H2D -> detect."""
    from irmv_detection_amd import frames
    exe = os.path.join(BIN, "camera_stream_test")
    if not os.path.exists(exe):
        exe = _compile("camera_stream_test.cpp", exe, True)
    (tmp_path / "yolov7.irmw").write_bytes(blob)
    frames.synthetic_batch(0, 8).tofile(tmp_path / "frames.bin")
    out = subprocess.run([exe, str(tmp_path / "yolov7.onnx"), str(tmp_path / "frames.bin"), "8", "2.0"], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"paced330 producer_fps (\S+) consumer_fps (\S+) lat_ms mean (\S+) p99 (\S+)", out.stdout)
    assert abs(float(m.group(1)) - 330) < 33 and float(m.group(2)) > 0.9 * float(m.group(1)) and float(m.group(4)) < 10.0
    m = re.search(r"unpaced producer_fps (\S+) consumer_fps (\S+)", out.stdout)
    assert float(m.group(2)) > 330        # un-paced, the consumer keeps up with far more than the camera rate
    m = re.search(r"unpaced_pipelined producer_fps (\S+) consumer_fps (\S+)", out.stdout)
    assert float(m.group(2)) > 330        # and with one frame's kernels running under the next frame's upload


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["kpt", "classical"])
def test_irm_detector_core_vs_oracle(tmp_path, blob, rm_test_image, kind):
    """the ROS-free core of the node (detect -> four points -> PnP -> quaternion -> message fields) against
    the oracle, for a keypoint-head model and for a bbox-only model (the reference's kind: classical extraction)."""
    from irmv_detection_amd import frames, weights
    exe = os.path.join(BIN, "irm_detector_core_test")
    if not os.path.exists(exe):
        exe = _compile("irm_detector_core_test.cpp", exe, True, link_oracle=True)
    (tmp_path / "yolov7.irmw").write_bytes(blob if kind == "kpt" else weights.synthetic_blob(0, nk=0))
    fr = [frames.synthetic_frame(i) for i in range(4)] + [np.ascontiguousarray(rm_test_image)]
    np.stack(fr).tofile(tmp_path / "frames.bin")
    out = subprocess.run([exe, str(tmp_path / "yolov7.onnx"), str(tmp_path / "frames.bin"), str(len(fr)), kind],
                         capture_output=True, text=True, timeout=600)
    print(out.stdout[-3000:])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    m = re.search(r"bboxes (\d+) armors (\d+) fails 0", out.stdout)
    assert m and int(m.group(1)) > 0
    if kind == "kpt":
        assert int(m.group(2)) > 0


@pytest.mark.gpu
def test_reference_test_flow_second_model_int8_416(tmp_path, frame0):
    """BASELINE configs[4] through the C++ facade: the reference's test program on the ShuffleNetV2-backbone variant with an
    int8 weight blob at a 416 x 416 network input (IRMV_NET_SIZE): same detections as the Python binding on that model."""
    from irmv_detection_amd import arch, weights
    from irmv_detection_amd.engine import YoloEngine
    exe = os.path.join(BIN, "yolo_test")
    if not os.path.exists(exe):
        exe = _compile("yolo_test.cpp", exe, True)
    blob = weights.quantize_blob_int8(weights.synthetic_blob(0, backbone=arch.BACKBONE_SHUFFLE))
    (tmp_path / "yolov7.irmw").write_bytes(blob)
    frame0.tofile(tmp_path / "frame.bin")
    env = dict(os.environ, IRMV_NET_SIZE="416")
    out = subprocess.run([exe, str(tmp_path / "yolov7.onnx"), str(tmp_path / "frame.bin"), "3"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    with YoloEngine(None, (1280, 1024), weights_blob=blob, net_size=416) as e:
        e.get_src_image_buffer()[:] = frame0
        bb = e.detect()
    assert len(bb) > 0
    assert int(re.search(r"bboxes (\d+)", out.stdout).group(1)) == len(bb)
    first = [float(v) for v in re.search(r"bbox 0 (\S+) (\S+) (\S+) (\S+) (\S+)", out.stdout).groups()]
    assert np.allclose(first[:4], bb[0].xyxy, atol=1e-4) and abs(first[4] - bb[0].score) < 1e-5
