#!/usr/bin/env python3
"""How far apart should the two concurrently replayed graphs of a batched step run?

bench.py submits a 256-frame step as two 128-frame graphs on two streams that start together: both streams then execute
the SAME kernel at the same time all the way down (two MFMA-bound launches side by side, then two HBM-bound ones).  This probe
feeds the two shares separately and starts the second one `offset` of a graph's duration after the first; both streams then
free-run through K replays.  Prints frames/s per offset (and bench.py's own submit order for reference).

    python3 scripts/probes/phase_probe.py [--steps 100] [--offsets 0,0.25,0.5,0.75]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from irmv_detection_amd import frames as F, weights   # noqa: E402
from irmv_detection_amd.engine import YoloEngine      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--offsets", default="0,0.25,0.5,0.75")
    ap.add_argument("--slots", type=int, default=256)
    a = ap.parse_args()
    B, H = a.slots, a.slots // 2
    with YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=B) as e:
        for s in range(B):
            e.get_src_image_buffer(s)[:] = F.synthetic_frame(s % 16)
        e.submit(0, B, h2d=True); e.wait()
        for _ in range(10):
            e.submit(0, B, h2d=False)
        e.wait()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            e.submit(0, B, h2d=False)
        e.wait()
        dt = time.perf_counter() - t0
        print(f"bench.py order (whole steps, streams start together): {a.steps * B / dt:9.0f} FPS  ({dt / a.steps * 1e3:.3f} ms per {B} frames)", flush=True)
        t0 = time.perf_counter()
        for _ in range(20):
            e.submit(0, H, h2d=False)
        e.wait()
        solo = (time.perf_counter() - t0) / 20
        print(f"one {H}-frame graph alone: {solo * 1e3:.3f} ms", flush=True)
        for off in [float(x) for x in a.offsets.split(",")]:
            for rep in range(2):
                t0 = time.perf_counter()
                e.submit(0, H, h2d=False)
                if off > 0:
                    t1 = time.perf_counter()
                    while time.perf_counter() - t1 < off * solo:
                        pass
                e.submit(H, H, h2d=False)
                for _ in range(a.steps - 1):
                    e.submit(0, H, h2d=False)
                    e.submit(H, H, h2d=False)
                e.wait()
                dt = time.perf_counter() - t0
                print(f"shares fed separately, second stream {off:4.2f} of a graph behind: {a.steps * B / dt:9.0f} FPS  ({dt / a.steps * 1e3:.3f} ms per {B} frames)", flush=True)


if __name__ == "__main__":
    main()
