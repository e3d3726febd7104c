"""One-off GPU-box probes behind one entry point (they were a dozen scripts).  Diagnostics only: nothing here
is imported by the product path, the tests or bench.py.

    python3 scripts/probe.py <mode> [args]

modes:
    tune     every autotuner candidate of every layer with its time (SLOTS=<n>; IRMV_AUTOTUNE_VERBOSE is set for you)
    lat      single-frame step / detect() / pipelined-slot times (TAG=<label> for the output line)
    host     host-inclusive throughput by upload-group size (PROBE_TORCH=1: with a torch HIP context up first)
    runtime  which libamdhip64 / libhsa-runtime serve the library: argument irmv_only | torch_first | irmv_first
    s2       stride-2 layers at a 416 net: LDS kernel vs chunk-major direct kernel vs its deep variant, tap by tap (NET=<size>)
    crash    every hand-off path once, a line before each phase: which one survives `rocprofv3 --kernel-trace`
    repro    the round-1 fault sequence (engine A, engine B created / used / destroyed, A replays with upload): run ONCE per change
    headerr  GPU head vs the fp32 oracle over FRAMES (64) synthetic frames + rm_test.jpg: max / p99 / median per Detect level and branch, the worst three taken apart tap by tap
    light    classical light extraction inside a bbox-only step, and the extract_armors API on realistic ROIs
    harness  tests/cpp/yolo_test: the reference yolo_engine_benchmark shape (100 warm-ups, 30 runs x 10)
    summary  one line per bench JSON: probe.py summary <bench.json> [label]
    dev      stage-by-stage comparison of the HIP engine with the oracle (development smoke; SLOTS=<n>)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the CPU oracle is OpenMP code: a GPU box shows every host CPU but grants a share of them, and a team as large as the visible
# count, spinning at its barriers, takes seconds per forward pass (tests/conftest.py does the same before liboracle.so loads)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, len(os.sched_getaffinity(0))))))


def cmd_tune(ARGS):
    """every autotuner candidate of every layer with its time (SLOTS=<n>; IRMV_AUTOTUNE_VERBOSE is set for you)"""
    import os, sys
    os.environ["IRMV_AUTOTUNE_VERBOSE"] = "1"
    from irmv_detection_amd import weights
    from irmv_detection_amd.engine import YoloEngine
    e = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=int(os.environ.get("SLOTS", "1")))
    e.close()


def cmd_lat(ARGS):
    """single-frame step / detect() / pipelined-slot times (TAG=<label> for the output line)"""
    import os, sys, time
    from irmv_detection_amd import frames, weights
    from irmv_detection_amd.engine import YoloEngine
    eng = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=3)
    for s in range(3):
        eng.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
    for _ in range(20):
        eng.detect(0)
    t0 = time.perf_counter()
    for _ in range(200):
        eng.submit(0, 1, h2d=False); eng.wait()
    t_res = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        eng.detect(0)
    t_det = (time.perf_counter() - t0) / 200
    res = []
    for depth in (1, 2):
        for j in range(depth):
            eng.submit(j, 1, async_upload=True)
        t0 = time.perf_counter()
        n = 300
        for i in range(n):
            eng.submit((i + depth) % 3, 1, async_upload=True)
            eng.wait_slots(i % 3, 1)
        eng.wait()
        res.append((time.perf_counter() - t0) / n)
    print(f"[{os.environ.get('TAG', '')}] streams {eng.num_streams}; step (HBM resident) {t_res*1e3:.4f} ms; detect (H2D inclusive) {t_det*1e3:.4f} ms; pipelined 3 slots: "
          f"2 in flight {res[0]*1e3:.4f} ms/frame = {1/res[0]:.0f} FPS, 3 in flight {res[1]*1e3:.4f} ms/frame = {1/res[1]:.0f} FPS", flush=True)
    eng.close()


def cmd_host(ARGS):
    """host-inclusive throughput by upload-group size (PROBE_TORCH=1: with a torch HIP context up first)"""
    import os, sys, time
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


def cmd_runtime(ARGS):
    """which libamdhip64 / libhsa-runtime serve the library: argument irmv_only | torch_first | irmv_first"""
    import ctypes, os, sys, time
    mode = ARGS[0] if len(ARGS) > 0 else "irmv_only"
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


def cmd_s2(ARGS):
    """stride-2 layers at a 416 net: LDS kernel vs chunk-major direct kernel vs its deep variant, tap by tap (NET=<size>)"""
    import os, sys
    import numpy as np
    from irmv_detection_amd import frames, weights
    from irmv_detection_amd.engine import YoloEngine
    blob = weights.synthetic_blob(0)
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (1024, 1280, 3), dtype=np.uint8)
    res = {}
    for kind in ("lds", "ct", "deep"):
        os.environ["IRMV_FORCE_S2"] = kind
        # IRMV_FORCE_S2 is read once per process (static): run each kind in a child
        import subprocess
        code = f"""
    import os, sys
    sys.path.insert(0, {ROOT!r})
    import numpy as np
    from irmv_detection_amd import weights
    from irmv_detection_amd.engine import YoloEngine
    img = np.random.default_rng(11).integers(0, 256, (1024, 1280, 3), dtype=np.uint8)
    with YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), net_size=int(os.environ.get('NET', '416'))) as e:
        names = [(s['layer'], s['name']) for s in e.profile(0, 1) if 's2' in s['name']]
        e.get_src_image_buffer()[:] = img
        e.detect()
        np.savez('/tmp/s2_{kind}.npz', head=e.read_head(0), **{{'t' + t: e.read_tap(t, 0) for t in ('3', '5', '7', '16', '18', '19', '21')}})
    print(names)
    """
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ))
        print(kind, out.stdout.strip()[-600:], out.stderr.strip()[-300:], flush=True)
        res[kind] = dict(np.load(f"/tmp/s2_{kind}.npz"))
    for k in ("ct", "deep"):
        for name in res["lds"]:
            d = np.abs(res["lds"][name] - res[k][name])
            print(f"lds vs {k}: {name:6s} differs {int((d > 0).sum()):7d} max {d.max():.5f}", flush=True)


def cmd_crash(ARGS):
    """every hand-off path once, a line before each phase: which one survives `rocprofv3 --kernel-trace`"""
    import os, sys, time
    from irmv_detection_amd import frames, weights
    from irmv_detection_amd.engine import YoloEngine
    def say(m): print(f"[{os.environ.get('TAG','')}] {m}", flush=True)
    eng = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0), num_slots=4)
    for s in range(4):
        eng.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
    say("engine up")
    for _ in range(5): eng.detect(0)
    say("detect ok")
    for _ in range(5):
        eng.submit(0, 1, h2d=False); eng.wait()
    say("submit hbm-resident ok")
    for _ in range(5):
        eng.submit(0, 4, h2d=True); eng.wait()
    say("batched inline h2d ok")
    for _ in range(5):
        eng.submit(0, 4, h2d=True, async_upload=True); eng.wait()
    say("batched async upload ok")
    eng.submit(0, 1, async_upload=True)
    for i in range(12):
        eng.submit((i + 1) % 3, 1, async_upload=True)
        eng.wait_slots(i % 3, 1)
    eng.wait()
    say("pipelined async upload ok")
    eng.close()
    say("closed")


def cmd_repro(ARGS):
    """the round-1 fault sequence (engine A, engine B created / used / destroyed, A replays with upload): run ONCE per change"""
    import os, sys
    import numpy as np
    if os.environ.get("IRMV_REPRO_TORCH", "1") == "1":
        import torch
        torch.cuda.set_device(0)
        keep = torch.zeros(1 << 20, device="cuda")
        torch.cuda.synchronize()
        print("torch context up", flush=True)
    from irmv_detection_amd import frames, weights
    from irmv_detection_amd.engine import YoloEngine
    blob = weights.synthetic_blob(0)
    N = int(os.environ.get("IRMV_REPRO_SLOTS", "128"))
    a = YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=N)
    for s in range(N):
        a.get_src_image_buffer(s)[:] = frames.synthetic_frame(s % 8)
    a.submit(0, N); a.wait(); print("A ran", flush=True)
    h0 = a.read_head(N - 1).copy()
    b = YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=1)
    b.get_src_image_buffer(0)[:] = frames.synthetic_frame(0)
    b.detect(); print("B ran", flush=True)
    b.close(); print("B closed", flush=True)
    def report(tag):
        if os.environ.get("IRMV_REPRO_REPORT"):
            nc = [a.read_raw(s)["n_candidates"] for s in (0, N // 2, N - 1)]
            print(f"{tag}: raw candidate counters of slots 0, {N // 2}, {N - 1}: {nc}", flush=True)
    report("after first run")
    for i in range(5):
        a.submit(0, N, h2d=True); a.wait(); print(f"A replayed with upload {i}", flush=True)
        report(f"replay {i}")
    a.submit(0, N, h2d=False); a.wait(); print("A replayed", flush=True)
    assert np.array_equal(a.read_head(N - 1), h0)
    a.close()
    print("repro clean: no fault, bits identical", flush=True)


def cmd_headerr(ARGS):
    """GPU head vs the fp32 oracle over FRAMES synthetic frames (default 64: frames 0 .. 63) + rm_test.jpg: max / p99 / median per Detect level and branch, the three worst frames taken apart tap by tap (tests/head_sweep.py); BOXES=1 adds box / keypoint differences per stride"""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import head_sweep
    from irmv_detection_amd import frames, weights
    from irmv_detection_amd.engine import YoloEngine
    from oracle import oracle
    oracle.build()
    blob = weights.synthetic_blob(0)
    net = oracle.Net(blob)
    n = int(os.environ.get("FRAMES", "64"))
    first = int(os.environ.get("FIRST", "0"))
    def frame_of(label):
        if isinstance(label, int):
            return frames.synthetic_frame(label)
        from PIL import Image
        return np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "rm_test.jpg")).convert("RGB"))
    def it():
        for label in list(range(first, first + n)) + ["rm_test"]:
            yield label, frame_of(label)
    with YoloEngine(None, (1280, 1024), weights_blob=blob) as e:
        res = head_sweep.sweep(e, net, it(), oracle)
        head_sweep.report(res)
        worst = sorted(res["per_frame"], key=lambda t: -t[1])[:int(os.environ.get("ATTRIBUTE", "3"))]
        for label, _ in worst:
            head_sweep.attribute(e, net, label, frame_of(label), oracle)
        if os.environ.get("BOXES") == "1":
            worst_px = {8: [0, 0], 16: [0, 0], 32: [0, 0]}
            for fi in range(first, first + min(n, 16)):
                f = frames.synthetic_frame(fi)
                e.get_src_image_buffer()[:] = f
                e.detect()
                raw = e.read_raw(0)
                ref = oracle.decode_nms(net.forward(oracle.preprocess(f, 640)), 640, 14, 8)
                gi = {(int(a_), int(c)): i for i, (a_, c) in enumerate(zip(raw["anchors"], raw["classes"]))}
                for i, (a_, c) in enumerate(zip(ref["anchors"], ref["classes"])):
                    j = gi.get((int(a_), int(c)))
                    if j is None:
                        continue
                    s = 8 if a_ < 6400 else (16 if a_ < 8000 else 32)
                    worst_px[s][0] = max(worst_px[s][0], float(np.abs(raw["boxes"][j] - ref["boxes"][i]).max()))
                    worst_px[s][1] = max(worst_px[s][1], float(np.abs(raw["kpts"][j] - ref["kpts"][i]).max()))
            print("shared survivors, max |d box| / |d kpt| px per stride:", worst_px)


def cmd_light(ARGS):
    """classical light extraction inside a bbox-only step, and the extract_armors API on realistic ROIs"""
    import os, sys
    import numpy as np
    from irmv_detection_amd import frames, weights
    from irmv_detection_amd.engine import YoloEngine
    B = int(os.environ.get("SLOTS", "32"))
    eng = YoloEngine(None, (1280, 1024), weights_blob=weights.synthetic_blob(0, nk=0), num_slots=B, num_streams=1)
    for s in range(B):
        eng.get_src_image_buffer(s)[:] = frames.synthetic_frame(s)
    eng.submit(0, B); eng.wait()
    nd = [len(eng.results(s)) for s in range(B)]
    runs = [eng.profile(0, B) for _ in range(4)][1:]
    for nm in ("light_extract", "nms_pnp", "decode"):
        t = min(st["ms"] for r in runs for st in r if st["name"] == nm)
        print(f"{nm:14s} {t*1e3:8.1f} us")
    print("detections per frame: mean", np.mean(nd), "max", max(nd))
    areas = []
    for s in range(B):
        for a in eng.results(s):
            x1, y1, x2, y2 = a.bbox_xyxy
            areas.append(max(0, min(x2, 1280) - max(x1, 0)) * max(0, min(y2, 1024) - max(y1, 0)))
    print("ROI area mean", np.mean(areas), "max", np.max(areas))
    import time
    t0 = time.perf_counter()
    for _ in range(30): eng.submit(0, B, h2d=False)
    eng.wait(); dt = (time.perf_counter() - t0) / 30
    print(f"graph step {dt*1e3:.3f} ms -> {B/dt:.0f} FPS")

    # realistic ROIs: boxes around the bright structures of a frame (what a trained detector would hand over)
    import scipy.ndimage as ndi
    from oracle import oracle
    frame = frames.synthetic_frame(1)
    rot = oracle.rotate180(frame)
    lab, n = ndi.label(rot.max(2) >= 200, structure=np.ones((3, 3)))
    sl = ndi.find_objects(lab)
    boxes = np.array([(s[1].start - 60, s[0].start - 20, s[1].stop + 60, s[0].stop + 20) for s in sl][:16], np.float32)
    eng.get_src_image_buffer(0)[:] = frame
    eng.extract_armors(boxes)
    t0 = time.perf_counter()
    for _ in range(20):
        arm = eng.extract_armors(boxes)
    dt = (time.perf_counter() - t0) / 20
    print(f"extract_armors API: {len(boxes)} boxes, mean ROI {np.mean((boxes[:,2]-boxes[:,0])*(boxes[:,3]-boxes[:,1])):.0f} px, {dt*1e3:.3f} ms per call (frame H2D included), "
          f"lights found {[a.n_lights for a in arm]}")


def cmd_harness(ARGS):
    """tests/cpp/yolo_test: the reference yolo_engine_benchmark shape (100 warm-ups, 30 runs x 10)"""
    import os, subprocess, sys, tempfile
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


def cmd_summary(ARGS):
    """one line per bench JSON: probe.py summary <bench.json> [label]"""
    import sys, json
    # usage: bench_summary.py <bench.json> [label]   (a file, never stdin: a forgotten pipe must not hang a GPU run)
    for line in open(ARGS[0]):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        r = d["roofline"]
        print(ARGS[1] if len(ARGS) > 1 else "", "B", d["config"]["frames_per_step_per_gpu"], "FPS", d["value"], "ms/step", d["ms_per_step"],
              "dom", r["kernel"], r["achieved"], r["unit"], "; all conv TF/s", r["all_conv_tflops"], "lat1", d.get("latency_ms_single_frame_h2d_inclusive"),
              "pcie", d.get("fps_pcie_inclusive_1gpu"))


def cmd_dev(ARGS):
    """stage-by-stage comparison of the HIP engine with the oracle (development smoke; SLOTS=<n>)"""
    import sys, os, time
    import numpy as np
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


MODES = {"tune": cmd_tune, "lat": cmd_lat, "host": cmd_host, "runtime": cmd_runtime, "s2": cmd_s2, "crash": cmd_crash, "repro": cmd_repro, "headerr": cmd_headerr, "light": cmd_light, "harness": cmd_harness, "summary": cmd_summary, "dev": cmd_dev}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in MODES:
        print(__doc__)
        sys.exit(2)
    MODES[sys.argv[1]](sys.argv[2:])
