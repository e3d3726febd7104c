"""Host-inclusive throughput under different hand-off settings (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('PROBE_TORCH') == '1':
    import torch
    torch.cuda.set_device(0); _keep = torch.zeros(1 << 20, device='cuda'); torch.cuda.synchronize()
    print('torch context up', flush=True)
from irmv_detection_amd import frames, weights
from irmv_detection_amd.engine import YoloEngine
blob = weights.synthetic_blob(0)
B = 128
fr = [frames.synthetic_frame(i) for i in range(8)]
for streams in (2,):
    eng = YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=B, num_streams=streams)
    for s in range(B):
        eng.get_src_image_buffer(s)[:] = fr[s % 8]
    eng.submit(0, B); eng.wait()
    def run(G, async_upload, h2d=True, steps=8):
        groups = [(f, min(G, B - f)) for f in range(0, B, G)]
        for _ in range(2):
            for f, c in groups:
                eng.submit(f, c, h2d=h2d, async_upload=async_upload)
        eng.wait()
        t0 = time.perf_counter()
        for _ in range(steps):
            for f, c in groups:
                eng.submit(f, c, h2d=h2d, async_upload=async_upload)
        t_sub = time.perf_counter() - t0
        eng.wait()
        dt = time.perf_counter() - t0
        print(f"streams {streams} group {G:3d} async_upload {int(async_upload)} h2d {int(h2d)}: {B*steps/dt:8.0f} FPS  ({B*steps*3932160/dt/1e9:5.1f} GB/s)  host submit time {t_sub/steps*1e3:.2f} ms/step", flush=True)
    for G in (64, 16):
        run(G, True)
    if os.environ.get('PROBE_TORCH') == '1':
        for G in (64, 16):
            torch.cuda.synchronize(); run(G, True, steps=10)
    eng.close()
