"""Deterministic synthetic camera frames.

The reference's only image fixture is test/rm_test.jpg (1280x1024); its camera
layer (src/camera.cpp, src/mv_camera.cpp) is out of scope, so benchmarks and
tests feed seeded frames of the camera's shape instead: dim uniform background
(below the reference's binary threshold of 150, src/irm_detector.cpp:152) with
0-8 "armors", each a pair of bright, slightly tilted vertical light bars.
"""
from __future__ import annotations

import numpy as np

FRAME_SEED_BASE = 0xC0FFEE


def synthetic_frame(frame_idx: int = 0, width: int = 1280, height: int = 1024) -> np.ndarray:
    """-> uint8 [height, width, 3], already in model channel order (the
    reference expects the producer to deposit RGB, test/yolo_test.cpp:25)."""
    rng = np.random.Generator(np.random.PCG64(FRAME_SEED_BASE + int(frame_idx)))
    img = rng.integers(0, 96, size=(height, width, 3), dtype=np.uint8)
    n_armors = int(rng.integers(0, 9))
    for _ in range(n_armors):
        bar_h = float(rng.uniform(40, 120))
        bar_w = float(rng.uniform(6, 14))
        tilt = np.deg2rad(float(rng.uniform(-15, 15)))
        dist = float(rng.uniform(1.5, 3.0)) * bar_h
        cx = float(rng.uniform(0.15, 0.85)) * width
        cy = float(rng.uniform(0.15, 0.85)) * height
        color = rng.integers(200, 256, size=3)
        for side in (-0.5, 0.5):
            bx, by = cx + side * dist, cy
            r = int(np.ceil(0.5 * np.hypot(bar_h, bar_w))) + 1
            x0, x1 = max(int(bx) - r, 0), min(int(bx) + r + 1, width)
            y0, y1 = max(int(by) - r, 0), min(int(by) + r + 1, height)
            if x0 >= x1 or y0 >= y1:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1]
            dx, dy = xx - bx, yy - by
            u = dx * np.cos(tilt) + dy * np.sin(tilt)       # across the bar
            v = -dx * np.sin(tilt) + dy * np.cos(tilt)      # along the bar
            mask = (np.abs(u) <= bar_w / 2) & (np.abs(v) <= bar_h / 2)
            img[y0:y1, x0:x1][mask] = color
    return img


def synthetic_batch(start_idx: int, count: int, width: int = 1280, height: int = 1024) -> np.ndarray:
    return np.stack([synthetic_frame(start_idx + i, width, height) for i in range(count)])
