"""Development smoke: stage-by-stage comparison of the HIP engine with the oracle (GPU box)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irmv_detection_amd import weights, frames
from irmv_detection_amd.engine import YoloEngine, PnPSolver
from oracle import oracle

blob = weights.synthetic_blob(0)
S = int(os.environ.get("SLOTS", "2"))
eng = YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=S)
onet = oracle.Net(blob)
fr = [frames.synthetic_frame(i) for i in range(S)]
for i in range(S):
    eng.get_src_image_buffer(i)[:] = fr[i]
eng.submit(0, S, h2d=True); eng.wait()
for i in range(S):
    x_o = oracle.preprocess(fr[i], 640)
    x_g = eng.read_input(i)
    x_o16 = x_o.astype(np.float16).astype(np.float32)
    print(f"[slot {i}] preprocess exact(fp16): {np.array_equal(x_g, x_o16)} maxdiff {np.abs(x_g-x_o16).max():.3g}")
    h_o, _ = None, None
    for tap in ("0", "1", "2", "4", "9", "12", "15", "21"):
        _, t_o = onet.forward(x_o, emulate_fp16=True, tap=tap)
        t_g = eng.read_tap(tap, i)
        print(f"   tap {tap:>3} shape {t_g.shape} maxabs diff vs emu-oracle {np.abs(t_g-t_o).max():.4g} (rms {np.sqrt((t_o**2).mean()):.3g})")
    h_e = onet.forward(x_o, emulate_fp16=True)
    h_f = onet.forward(x_o, emulate_fp16=False)
    h_g = eng.read_head(i)
    print(f"   head maxabs diff: vs emu {np.abs(h_g-h_e).max():.4g}  vs fp32 {np.abs(h_g-h_f).max():.4g}")
    raw = eng.read_raw(i)
    d_o = oracle.decode_nms(h_g, 640, 14, 8)   # oracle post on the GPU's own head
    same = raw["num_dets"] == d_o["num_dets"] and np.array_equal(raw["anchors"], d_o["anchors"]) and np.array_equal(raw["classes"], d_o["classes"])
    print(f"   post on shared head: gpu dets {raw['num_dets']} cand {raw['n_candidates']} | oracle dets {d_o['num_dets']} cand {d_o['n_candidates']} | survivors identical {same}")
    if raw["num_dets"] == d_o["num_dets"] and raw["num_dets"]:
        print(f"   boxes bitexact {np.array_equal(raw['boxes'], d_o['boxes'])} scores bitexact {np.array_equal(raw['scores'], d_o['scores'])} kpts bitexact {np.array_equal(raw['kpts'], d_o['kpts'])}")
    arm = eng.results(i)
    K = np.array(YoloEngine.__init__.__kwdefaults__["camera_matrix"]); D = np.array(YoloEngine.__init__.__kwdefaults__["dist_coeffs"])
    bad = 0; worst = 0.0
    for a in arm:
        o = oracle.solve_pnp_ippe(K, D, a.image_points(), 0)
        if o["ok"] != a.pnp_ok: bad += 1; continue
        if a.pnp_ok:
            worst = max(worst, np.abs(o["tvec"]-a.tvec).max(), np.abs(o["rvec"]-a.rvec).max())
    print(f"   pnp: {len(arm)} armors, ok-mismatch {bad}, worst |d| {worst:.3g}")
t0=time.time()
for _ in range(20):
    eng.submit(0, S, h2d=False)
eng.wait(); dt=(time.time()-t0)/20
print(f"step (S={S}) {dt*1e3:.3f} ms -> {S/dt:.1f} FPS")
st = eng.profile(0, S)
tot = sum(s["ms"] for s in st)
print(f"eager profile total {tot:.3f} ms over {len(st)} kernels")
agg = {}
for s in st:
    a = agg.setdefault(s["name"], [0, 0.0, 0.0]); a[0]+=1; a[1]+=s["ms"]; a[2]+=s["flops"]
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print(f"   {k:28s} n={v[0]:3d} ms={v[1]:.3f} TF/s={v[2]/max(v[1],1e-9)/1e9:.2f}")
