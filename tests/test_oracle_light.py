"""Oracle for the classical light extraction (row f1; reference src/irm_detector.cpp:292-355,
include/irmv_detection/armor.hpp) against independent numpy/scipy statements."""
import numpy as np
from scipy import ndimage
from scipy.spatial import ConvexHull

from irmv_detection_amd import frames
from oracle import oracle


def test_rectangle_contours_in_opencv_order():
    b = np.zeros((12, 14), np.uint8)
    b[2:9, 3:6] = 255
    b[4:10, 9:11] = 255
    b[0, 0] = 255
    cs = oracle.find_external_contours(b)
    # last found first; each rectangle: top-left, bottom-left, bottom-right, top-right (cv::findContours' walk)
    assert [c.tolist() for c in cs] == [[[9, 4], [9, 9], [10, 9], [10, 4]], [[3, 2], [3, 8], [5, 8], [5, 2]], [[0, 0]]]


def test_external_only_and_8_connectivity():
    r = np.zeros((11, 11), np.uint8)
    r[1:10, 1:10] = 255
    r[3:8, 3:8] = 0
    r[5, 5] = 255                                  # a dot inside the ring's hole is not an external contour
    assert len(oracle.find_external_contours(r)) == 1
    d = np.zeros((6, 6), np.uint8)
    d[1, 1] = d[2, 2] = d[3, 3] = 255              # diagonal chain = ONE 8-connected component
    assert len(oracle.find_external_contours(d)) == 1


def test_contour_count_matches_connected_components_on_random_blobs():
    rng = np.random.default_rng(0)
    for _ in range(20):
        img = (ndimage.gaussian_filter(rng.random((40, 60)), 2.0) > 0.52).astype(np.uint8) * 255
        lab, n = ndimage.label(img, structure=np.ones((3, 3)))
        holes = ndimage.binary_fill_holes(img > 0)          # components nested in holes are not external
        lab2, n_ext = ndimage.label(holes, structure=np.ones((3, 3)))
        cs = oracle.find_external_contours(img)
        assert len(cs) == n_ext
        for c in cs:                                         # every contour point is a foreground pixel
            assert (img[c[:, 1], c[:, 0]] > 0).all()


def test_min_area_rect_against_bruteforce():
    rng = np.random.default_rng(1)
    for _ in range(30):
        pts = rng.integers(0, 60, (int(rng.integers(5, 40)), 2)).astype(np.int16)
        c = oracle.min_area_rect(pts).astype(np.float64)
        area = np.linalg.norm(c[1] - c[0]) * np.linalg.norm(c[2] - c[1])
        # brute force over hull edges in float64 numpy
        hull = pts[ConvexHull(pts.astype(float)).vertices].astype(float)
        best = np.inf
        for i in range(len(hull)):
            u = hull[(i + 1) % len(hull)] - hull[i]
            u /= np.linalg.norm(u)
            s = (hull - hull[i]) @ u
            t = (hull - hull[i]) @ np.array([-u[1], u[0]])
            best = min(best, (s.max() - s.min()) * (t.max() - t.min()))
        assert abs(area - best) <= 1e-3 * max(best, 1.0)
        # all points inside the rectangle (small tolerance)
        e0, e1 = c[1] - c[0], c[3] - c[0]
        for p in pts.astype(float):
            a, b_ = (p - c[0]) @ e0 / max(e0 @ e0, 1e-12), (p - c[0]) @ e1 / max(e1 @ e1, 1e-12)
            assert -1e-4 <= a <= 1 + 1e-4 and -1e-4 <= b_ <= 1 + 1e-4


def _bar_frame():
    """Two tilted bright bars on a dark frame = one armor (the geometry frames.synthetic_frame draws)."""
    img = np.full((240, 320, 3), 30, np.uint8)
    yy, xx = np.mgrid[0:240, 0:320]
    for cx in (110.0, 210.0):
        t = np.deg2rad(8.0)
        u = (xx - cx) * np.cos(t) + (yy - 120.0) * np.sin(t)
        v = -(xx - cx) * np.sin(t) + (yy - 120.0) * np.cos(t)
        img[(np.abs(u) <= 5) & (np.abs(v) <= 30)] = (250, 240, 230)
    return img


def test_extract_armor_on_a_synthetic_armor():
    img = _bar_frame()
    r = oracle.extract_armor(img, (60, 60, 260, 180))
    assert r["ok"] and r["n_lights"] == 2 and r["size"] == 0      # centre distance 100 / length ~60 = 1.67 -> small
    lb, lt, rt, rb = r["pts"]
    assert lb[0] < rb[0] and lt[0] < rt[0] and lt[1] < lb[1] and rt[1] < rb[1]
    # bar ends: the centre line of a bar tilted by 8 degrees, half length 30
    assert abs(np.hypot(*(lt - lb)) - 60) < 3 and abs(np.hypot(*(rt - rb)) - 60) < 3
    assert abs((lb[0] + lt[0]) / 2 - 110) < 1.5 and abs((rb[0] + rt[0]) / 2 - 210) < 1.5
    # threshold above the bars' gray level: nothing
    assert not oracle.extract_armor(img, (60, 60, 260, 180), oracle.light_params(binary_threshold=252))["ok"]
    # only one bar in the box
    assert oracle.extract_armor(img, (60, 60, 160, 180))["n_lights"] == 1
    # degenerate / outside boxes
    assert not oracle.extract_armor(img, (400, 10, 500, 60))["ok"] and not oracle.extract_armor(img, (50, 50, 50.4, 90))["ok"]


def test_gray_and_roi_conventions():
    # gray = (c0*3735 + c1*19235 + c2*9798 + 2^14) >> 15 with the frame read as BGR (src/irm_detector.cpp:309)
    img = np.zeros((40, 40, 3), np.uint8)
    img[10:30, 10:14] = (0, 255, 0)         # gray 150 -> NOT above the threshold of 150
    img[10:30, 24:28] = (0, 255, 3)         # gray 151 -> above
    assert oracle.extract_armor(img, (0, 0, 40, 40))["n_lights"] <= 1
    cs = oracle.find_external_contours(((img.astype(int) @ np.array([3735, 19235, 9798]) + (1 << 14)) >> 15 > 150).astype(np.uint8))
    assert len(cs) == 1


def test_rm_test_jpg_armor_golden(rm_test_image):
    """The reference's own test image holds one armor (two light bars at x 647-654 / 759-767, y 375-404): committed
    golden of the classical extraction on 15 boxes around it (tests/golden/make_golden.py::light_cases)."""
    import json
    from conftest import golden_path
    cases = json.load(open(golden_path("light_cases.json")))
    assert sum(c["ok"] for c in cases) == 5
    for c in cases:
        o = oracle.extract_armor(rm_test_image, np.array(c["box"], np.float32))
        assert o["ok"] == c["ok"] and o["n_lights"] == c["n_lights"], c["box"]
        if c["ok"]:
            assert int(o["size"]) == c["size"] and np.array_equal(np.asarray(o["pts"], np.float32).ravel(), np.array(c["pts"], np.float32))
