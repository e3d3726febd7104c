"""Which libamdhip64.so.7 / libhsa-runtime64 serve libirmv_hip.so, with and without torch in the process, and what the
host-inclusive rate is in each case (the "torch halves the hand-off rate" question).
    python3 scripts/runtime_probe.py irmv_only | torch_first | irmv_first
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "irmv_only"
if mode == "torch_first":
    import torch
    torch.cuda.set_device(0); _keep = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
from irmv_detection_amd import capi, frames, weights
lib = capi.load()
if mode == "irmv_first":
    import torch
    torch.cuda.set_device(0); _keep = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
paths = sorted({l.split()[-1] for l in open("/proc/self/maps") if ("libamdhip64" in l or "libhsa-runtime" in l or "librccl" in l or "librocprofiler" in l)})
print(f"[{mode}] mapped runtime libraries:")
for p in paths:
    ver = ""
    if "libamdhip64" in p:
        v = ctypes.c_int(0)
        try:
            ctypes.CDLL(p).hipRuntimeGetVersion(ctypes.byref(v)); ver = f"  hipRuntimeGetVersion = {v.value}"
        except Exception as e:  # noqa
            ver = f"  ({e})"
    print("   ", p, ver)
print(f"[{mode}] irmv_version: {lib.irmv_version().decode() if hasattr(lib, 'irmv_version') else '?'}")
from irmv_detection_amd.engine import YoloEngine
B = 64
eng = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=B, num_streams=2)
fr = [frames.synthetic_frame(i) for i in range(8)]
for s in range(B):
    eng.get_src_image_buffer(s)[:] = fr[s % 8]
eng.submit(0, B); eng.wait()
for G in (32, 16):
    groups = [(f, min(G, B - f)) for f in range(0, B, G)]
    for _ in range(2):
        for f, c in groups: eng.submit(f, c, h2d=True, async_upload=True)
    eng.wait()
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        for f, c in groups: eng.submit(f, c, h2d=True, async_upload=True)
    t_sub = time.perf_counter() - t0
    eng.wait()
    dt = time.perf_counter() - t0
    print(f"[{mode}] group {G}: {B*steps/dt:8.0f} FPS host-inclusive ({B*steps*3932160/dt/1e9:5.1f} GB/s), host submit {t_sub/steps*1e3:.2f} ms/step", flush=True)
eng.close()
