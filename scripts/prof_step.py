"""Lean driver for rocprofv3: N steps of the hot path, eager or hipGraph, no torch.

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 scripts/prof_step.py --mode graph
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine

ap = argparse.ArgumentParser()
ap.add_argument("--mode", choices=["graph", "eager", "graph_h2d"], default="graph")
ap.add_argument("--frames-per-step", type=int, default=16)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--torch", choices=["no", "init", "devblob"], default="no")
args = ap.parse_args()
B = args.frames_per_step
blob = weights.synthetic_blob(0)
kw = dict(weights_blob=blob)
if args.torch != "no":
    import torch
    torch.cuda.set_device(0)
    keep = torch.zeros(1024, device="cuda")
    torch.cuda.synchronize()
    if args.torch == "devblob":
        wt = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to("cuda")
        torch.cuda.synchronize()
        kw = dict(weights_device_ptr=wt.data_ptr(), weights_bytes=wt.numel())
    print("torch ready", flush=True)
eng = YoloEngine(None, (1280, 1024), num_slots=B, **kw)
for s in range(B):
    eng.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
print("engine ready", flush=True)
if args.mode == "graph":
    eng.submit(0, B, h2d=True); eng.wait()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.submit(0, B, h2d=False)
    eng.wait()
elif args.mode == "graph_h2d":
    eng.submit(0, B, h2d=True); eng.wait()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.submit(0, B, h2d=True)
    eng.wait()
else:
    eng.submit(0, B, h2d=True); eng.wait()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.profile(0, B)
dt = time.perf_counter() - t0
print(f"{args.mode}: {args.steps} steps of {B} frames in {dt*1e3:.2f} ms -> {args.steps*B/dt:.0f} FPS", flush=True)
eng.close()
