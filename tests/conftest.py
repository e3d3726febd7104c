"""pytest configuration: `gpu` marker, import paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The CPU oracle is OpenMP code.  A GPU box shows every host CPU but grants a share of them: a team as large as the visible
# CPU count, spinning at its barriers, turns a 64 x 64 forward pass into a minute.  Before liboracle.so (libgomp) loads:
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, len(os.sched_getaffinity(0))))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def blob():
    from irmv_detection_amd import weights
    return weights.synthetic_blob(0)


@pytest.fixture(scope="session")
def onet(blob):
    from oracle import oracle
    oracle.build()
    return oracle.Net(blob)


@pytest.fixture(scope="session")
def frame0():
    from irmv_detection_amd import frames
    return frames.synthetic_frame(0)


@pytest.fixture(scope="session")
def rm_test_image():
    """The reference's only image fixture (reference test/rm_test.jpg, 1280x1024)."""
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GOLDEN, "rm_test.jpg")).convert("RGB"))


def golden_path(name):
    return os.path.join(GOLDEN, name)


# reference config/camera_info.yaml:7,12
K_REF = np.array([957.669211, 0, 345.943891, 0, 969.127115, 284.057302, 0, 0, 1.0])
D_REF = np.array([-0.405274, 0.126058, -0.026939, -0.006503, 0.0])
