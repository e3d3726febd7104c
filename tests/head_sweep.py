"""Head-tensor error of the HIP engine against the fp32 oracle over frames nobody picked, and where it is made.

Test infrastructure (imports the oracle): used by tests/test_gpu_engine.py::test_head_error_over_unchosen_frames and by
`scripts/probe.py headerr`.  The reference pins nothing here (its tests assert a box count and a latency,
/root/reference/test/yolo_test.cpp:36,106; the network is an FP16 TensorRT plan, src/yolo_engine.cpp:105), so the stated
tolerance of this build is the contract, and this sweep is what it is stated ON.
"""
from __future__ import annotations

import numpy as np

LEVELS = ((0, 6400), (6400, 8000), (8000, 8400))          # anchors of Detect levels 0 / 1 / 2 at a 640 net
BRANCHES = (("box", 0, 64), ("cls", 64, 78), ("kpt", 78, 86))
TAPS = ("0", "1", "2", "3", "4", "5", "6", "7", "8", "9", "12", "15", "16", "18", "19", "21")


def frame_error(eng, net, frame, oracle):
    """One frame through detect(): |head_gpu - head_fp32| as an array, plus both heads."""
    eng.get_src_image_buffer(0)[:] = frame
    eng.detect(0)
    hg = eng.read_head(0)
    h32 = net.forward(oracle.preprocess(frame, 640))
    return np.abs(hg - h32), hg, h32


def sweep(eng, net, frames_iter, oracle, log=print):
    """frames_iter: iterable of (label, frame).  Returns dict(per_frame=[(label, max)], cells={(level, branch): [per-frame max]},
    elem={(level, branch): pooled |d| quantiles}) and logs one line per frame."""
    per_frame, cells, pooled = [], {}, {}
    for label, frame in frames_iter:
        d, _, _ = frame_error(eng, net, frame, oracle)
        per_frame.append((label, float(d.max())))
        worst = None
        for li, (a0, a1) in enumerate(LEVELS):
            for bn, c0, c1 in BRANCHES:
                blk = d[a0:a1, c0:c1]
                m = float(blk.max())
                cells.setdefault((li, bn), []).append(m)
                pooled.setdefault((li, bn), []).append(np.quantile(blk, [0.5, 0.99, 0.9999]))
                if worst is None or m > worst[0]:
                    worst = (m, li, bn)
        log(f"frame {label!s:>8}: max|d| {d.max():.4f}  mean {d.mean():.5f}  worst cell: level {worst[1]} {worst[2]}")
    return dict(per_frame=per_frame, cells=cells, pooled=pooled)


def report(res, log=print):
    """max / p99 / median of the per-frame maxima, overall and per (Detect level, branch); pooled element quantiles beside them."""
    mx = np.array([m for _, m in res["per_frame"]])
    log(f"head |d| vs fp32 oracle over {len(mx)} frames: max {mx.max():.4f}  p99 {np.quantile(mx, 0.99):.4f}  p90 {np.quantile(mx, 0.9):.4f}  "
        f"median {np.median(mx):.4f}  min {mx.min():.4f};  frames over 0.03: {int((mx > 0.03).sum())}")
    log("per (level, branch): per-frame maxima max / p99 / median   |   element |d| median / p99 / p99.99 (mean over frames)")
    for (li, bn), v in sorted(res["cells"].items()):
        v = np.array(v)
        q = np.mean(np.array(res["pooled"][(li, bn)]), axis=0)
        log(f"  level {li} {bn}: {v.max():.4f} / {np.quantile(v, 0.99):.4f} / {np.median(v):.4f}   |   {q[0]:.5f} / {q[1]:.4f} / {q[2]:.4f}")
    return float(mx.max()), float(np.quantile(mx, 0.99)), float(np.median(mx))


def attribute(eng, net, label, frame, oracle, log=print):
    """Where one frame's head error is made: every tap against the fp32 oracle, then the worst Detect cell taken apart --
    the error its final 1x1 INHERITS (fp32 1x1 on the GPU's own cvX.Y.1 tensor), the share of that which is the fp16
    rounding of the 1x1's input alone (fp32 1x1 on the ORACLE's tensor rounded to fp16), and what the GPU's own 1x1 adds."""
    d, hg, h32 = frame_error(eng, net, frame, oracle)
    x = oracle.preprocess(frame, 640)
    log(f"--- attribution, frame {label}: head max|d| {d.max():.4f}")
    for t in TAPS:
        _, to = net.forward(x, tap=t)
        tg = eng.read_tap(t, 0)
        e = np.abs(tg - to)
        log(f"  tap {t:>3} {str(to.shape):>14}: max|d| {e.max():.4f}  rms d {np.sqrt((e ** 2).mean()):.5f}  rms value {np.sqrt((to ** 2).mean()):.3f}  "
            f"rel {np.sqrt((e ** 2).mean()) / max(np.sqrt((to ** 2).mean()), 1e-9):.2e}")
    out = {}
    for li, (a0, a1) in enumerate(LEVELS):
        for br, (bn, c0, c1) in zip(("cv2", "cv3", "cv4"), BRANCHES):
            cell = d[a0:a1, c0:c1]
            S = int(round((a1 - a0) ** 0.5))
            e_in = {}
            for st in ("0", "1"):
                _, to = net.forward(x, tap=f"22.{br}.{li}.{st}")
                try:
                    tg = eng.read_tap(f"22.{br}.{li}.{st}", 0)
                except Exception:   # (an engine that merges the first-stage Detect convs has no stand-alone cvX.Y.0 tensor)
                    e_in[st] = (float("nan"), float("nan"), float(np.sqrt((to ** 2).mean())), None, to)
                    continue
                e_in[st] = (float(np.abs(tg - to).max()), float(np.sqrt(((tg - to) ** 2).mean())), float(np.sqrt((to ** 2).mean())), tg, to)
            cout = c1 - c0
            name = f"model.22.{br}.{li}.2"
            ref = h32[a0:a1, c0:c1].reshape(S, S, cout)
            inherit = net.conv_layer(name, e_in["1"][3], cout, 1)                                   # fp32 1x1 on the GPU's input tensor
            rounded = net.conv_layer(name, e_in["1"][4].astype(np.float16).astype(np.float32), cout, 1)   # fp32 1x1 on round16(oracle input)
            own = hg[a0:a1, c0:c1].reshape(S, S, cout) - inherit
            out[(li, bn)] = (float(cell.max()), float(np.abs(inherit - ref).max()), float(np.abs(rounded - ref).max()), float(np.abs(own).max()))
            log(f"  level {li} {bn}: head max|d| {cell.max():.4f} = inherited by the final 1x1 {np.abs(inherit - ref).max():.4f} "
                f"(input rounding alone would give {np.abs(rounded - ref).max():.4f}) + the 1x1's own arithmetic {np.abs(own).max():.4f};  "
                f"stage-0 tap max|d| {e_in['0'][0]:.4f} (rms {e_in['0'][1]:.5f} on {e_in['0'][2]:.2f}), stage-1 tap {e_in['1'][0]:.4f} (rms {e_in['1'][1]:.5f} on {e_in['1'][2]:.2f})")
    return out
