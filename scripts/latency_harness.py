"""Runs tests/cpp/yolo_test (the reference's yolo_engine_benchmark shape: 100 warm-ups, 30 runs x 10 iterations, each =
memcpy of the 3.93 MB frame into the engine's input slot + detect()) and prints its detect_ms line."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from irmv_detection_amd import frames, weights
import test_cpp_facade as T
exe = os.path.join(T.BIN, "yolo_test")
if not os.path.exists(exe):
    exe = T._compile("yolo_test.cpp", exe, True)
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "yolov7.irmw"), "wb").write(weights.synthetic_blob(0))
    frames.synthetic_frame(0).tofile(os.path.join(d, "frame.bin"))
    out = subprocess.run([exe, os.path.join(d, "yolov7.onnx"), os.path.join(d, "frame.bin"), "30"], capture_output=True, text=True, timeout=600)
    print(out.stdout[-600:], out.stderr[-300:])
