"""Generate the committed golden vectors under tests/golden/ from the CPU oracle.

The reference's tests assert no numerical output (test/yolo_test.cpp:36,106), so
these vectors are the build's own known-answer layer: produced here by the
oracle, each cross-checked in tests/ against an independent implementation
(torch-CPU, numpy brute force, scipy) and then used to pin both the oracle (no
silent drift) and the HIP path.

Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from irmv_detection_amd import frames, weights  # noqa: E402
from oracle import oracle  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
K_REF = np.array([957.669211, 0, 345.943891, 0, 969.127115, 284.057302, 0, 0, 1.0])
D_REF = np.array([-0.405274, 0.126058, -0.026939, -0.006503, 0.0])


def pre_cases():
    out = {}
    rng = np.random.Generator(np.random.PCG64(11))
    cfgs = [(0, 1, 0), (0, 0, 0), (0, 1, 1), (1, 1, 0), (1, 0, 1), (0, 1, 0), (1, 1, 0), (0, 0, 1)]
    sizes = [(64, 48), (64, 48), (50, 38), (64, 48), (40, 72), (33, 31), (96, 32), (64, 64)]
    for i, ((mode, rot, swap), (w, h)) in enumerate(zip(cfgs, sizes)):
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        _, u8 = oracle.preprocess(src, 32, mode, bool(rot), bool(swap), want_u8=True)
        out[f"src{i}"] = src
        out[f"cfg{i}"] = np.array([mode, rot, swap])
        out[f"out{i}"] = u8
    np.savez_compressed(os.path.join(G, "pre_cases.npz"), **out)


def rm_test():
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(G, "rm_test.jpg")).convert("RGB"))
    x, u8 = oracle.preprocess(img, 640, 0, True, False, want_u8=True)
    h16 = x.astype(np.float16)
    meta = dict(src_shape=list(img.shape), src_crc32=zlib.crc32(img.tobytes()),
                u8_crc32=zlib.crc32(u8.tobytes()), fp16_chw_crc32=zlib.crc32(h16.tobytes()),
                u8_samples={f"{y},{x_}": u8[y, x_].tolist() for y, x_ in ((0, 0), (100, 200), (320, 320), (639, 639))})
    with open(os.path.join(G, "rm_test_pre.json"), "w") as f:
        json.dump(meta, f, indent=1)


def net_blocks():
    blob = weights.synthetic_blob(0)
    net = oracle.Net(blob)
    x = oracle.preprocess(frames.synthetic_frame(0), 640)[:, 192:256, 320:384].copy()  # 64x64 crop
    out = dict(x=x.astype(np.float16))
    x = out["x"].astype(np.float32)
    head = net.forward(x)
    out["head"] = head
    for tap in ("0", "2", "4", "9", "15", "21", "22.cv2.0.1"):
        _, t = net.forward(x, tap=tap)
        out["tap_" + tap] = t.astype(np.float32)
    np.savez_compressed(os.path.join(G, "net_blocks.npz"), **out)
    # full-size summary
    xf = oracle.preprocess(frames.synthetic_frame(0), 640)
    hf = net.forward(xf)
    d = oracle.decode_nms(hf, 640, 14, 8)
    flat = hf.reshape(-1)
    top = np.argsort(-flat, kind="stable")[:32]
    np.savez_compressed(os.path.join(G, "full_640.npz"), top_idx=top.astype(np.int64), top_val=flat[top],
                        col_mean=hf.mean(0), col_std=hf.std(0), n_candidates=d["n_candidates"],
                        boxes=d["boxes"], scores=d["scores"], classes=d["classes"], anchors=d["anchors"],
                        kpts=d["kpts"])


def iou(a, b):
    ix = max(0.0, min(a[2], b[2]) - max(a[0], b[0])); iy = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = ix * iy
    u = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
    return inter / u if u > 0 else float("nan")


def brute_nms(boxes, classes, thr, max_det):
    keep = []
    for i in range(len(boxes)):
        if len(keep) >= max_det:
            break
        if all(not (classes[j] == classes[i] and iou(boxes[j], boxes[i]) > thr) for j in keep):
            keep.append(i)
    return keep


def nms_cases():
    rng = np.random.Generator(np.random.PCG64(5))
    cases = {}
    cases["empty"] = dict(boxes=[], classes=[], thr=0.45, max_det=100)
    cases["one"] = dict(boxes=[[10, 10, 50, 40]], classes=[3], thr=0.45, max_det=100)
    b = [[100 + i, 100 + i, 200 + i, 180 + i] for i in range(12)]
    cases["overlap_same_class"] = dict(boxes=b, classes=[2] * 12, thr=0.45, max_det=100)
    cases["overlap_diff_class"] = dict(boxes=b, classes=[i % 3 for i in range(12)], thr=0.45, max_det=100)
    many = [[float(30 * (i % 20)), float(30 * (i // 20)), float(30 * (i % 20) + 20), float(30 * (i // 20) + 20)] for i in range(200)]
    cases["more_than_max_det"] = dict(boxes=many, classes=[0] * 200, thr=0.45, max_det=100)
    ties = [[50, 50, 100, 100], [50, 50, 100, 100], [52, 50, 102, 100], [300, 300, 340, 350], [300, 300, 340, 350]]
    cases["ties_identical_boxes"] = dict(boxes=ties, classes=[1, 1, 1, 4, 5], thr=0.45, max_det=100)
    rb = []
    for _ in range(300):
        cx, cy, w, h = rng.uniform(50, 590), rng.uniform(50, 590), rng.uniform(10, 120), rng.uniform(10, 120)
        rb.append([float(np.float32(cx - w / 2)), float(np.float32(cy - h / 2)), float(np.float32(cx + w / 2)), float(np.float32(cy + h / 2))])
    cases["random300"] = dict(boxes=rb, classes=[int(c) for c in rng.integers(0, 14, 300)], thr=0.45, max_det=100)
    for name, c in cases.items():
        exp = brute_nms(np.array(c["boxes"], np.float64).reshape(-1, 4), c["classes"], c["thr"], c["max_det"])
        got = oracle.nms_sorted(np.array(c["boxes"], np.float32).reshape(-1, 4), np.array(c["classes"], np.int32), c["thr"], c["max_det"])
        assert list(got) == exp, name
        c["keep"] = exp
    with open(os.path.join(G, "nms_cases.json"), "w") as f:
        json.dump(cases, f)


def rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-14:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def project(K, D, R, t, obj):
    P = obj @ R.T + t
    x, y = P[:, 0] / P[:, 2], P[:, 1] / P[:, 2]
    r2 = x * x + y * y
    cd = 1 + D[0] * r2 + D[1] * r2 ** 2 + D[4] * r2 ** 3
    xd = x * cd + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * cd + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    return np.stack([K[0] * xd + K[2], K[4] * yd + K[5]], 1)


def pnp_cases():
    rng = np.random.Generator(np.random.PCG64(9))
    # model (x fwd, y left, z up) -> camera optical (x right, y down, z fwd) base rotation
    base = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], float)
    cases = []
    specs = [(0, 2.0, 0, 0, 0, 0, 0), (0, 3.0, 25, 10, 5, 0.3, -0.2), (1, 4.0, -35, -15, 8, -0.5, 0.3), (0, 1.2, 50, 5, -10, 0.1, 0.1),
             (0, 6.0, 10, 30, 2, 0.8, -0.4), (1, 2.5, -20, 20, -5, -0.3, 0.2), (0, 1.5, 5, 2, 1, 0.0, 0.0), (0, 8.0, -45, 0, 0, 1.0, 0.5)]
    for _ in range(4):
        specs.append((int(rng.integers(0, 2)), float(rng.uniform(1, 7)), float(rng.uniform(-50, 50)), float(rng.uniform(-25, 25)),
                      float(rng.uniform(-10, 10)), float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-0.4, 0.4))))
    for size, z, yaw, pitch, roll, tx, ty in specs:
        obj = oracle.armor_object_points(size)
        yw, pt, rl = np.deg2rad([yaw, pitch, roll])
        Rz = np.array([[np.cos(yw), -np.sin(yw), 0], [np.sin(yw), np.cos(yw), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(pt), 0, np.sin(pt)], [0, 1, 0], [-np.sin(pt), 0, np.cos(pt)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(rl), -np.sin(rl)], [0, np.sin(rl), np.cos(rl)]])
        R = base @ Rz @ Ry @ Rx
        t = np.array([tx, ty, z])
        for K, D, tag in ((K_REF, D_REF, "ref"), (K_REF, np.zeros(5), "nodist")):
            uv = project(K, D, R, t, obj).astype(np.float32)
            o = oracle.solve_pnp_ippe(K, D, uv, size)
            cases.append(dict(K=K.tolist(), D=D.tolist(), size=size, tag=tag, R_true=R.tolist(), t_true=t.tolist(),
                              img_pts=uv.reshape(-1).tolist(), ok=bool(o["ok"]), rvec=o["rvec"].tolist(), tvec=o["tvec"].tolist(),
                              rvec2=o["rvec2"].tolist(), tvec2=o["tvec2"].tolist(), err=o["err"].tolist()))
    with open(os.path.join(G, "pnp_cases.json"), "w") as f:
        json.dump(cases, f)


LIGHT_BOXES = [   # xyxy on rm_test.jpg: the armor (two vertical light bars at x 647-654 / 759-767, y 375-404) and non-armors
    (630, 360, 785, 420), (600, 340, 820, 440), (640.5, 370.2, 770.9, 410.7), (500, 300, 900, 500),   # the armor, tight to loose
    (630, 360, 700, 420), (700, 360, 785, 420),                 # one light each
    (630, 388, 785, 420), (630, 360, 785, 390),                 # bars cut by the box edge
    (430, 690, 680, 740), (640, 690, 680, 715),                 # the three horizontal blobs below: no lights (tilt / ratio gates)
    (0, 0, 1280, 1024), (-50, -50, 700, 420), (758, 374, 768, 405), (100, 100, 300, 300), (653.2, 380.0, 760.1, 400.0),
]


def light_cases():
    """Classical extraction (row f1) on the reference's own test image: pins orc_light.c bit for bit."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(G, "rm_test.jpg")).convert("RGB"))
    out = []
    for b in LIGHT_BOXES:
        o = oracle.extract_armor(img, np.array(b, np.float32))
        out.append(dict(box=list(map(float, b)), ok=bool(o["ok"]), size=int(o["size"]) if o["ok"] else -1, n_lights=int(o["n_lights"]),
                        pts=[float(v) for v in np.asarray(o["pts"]).ravel()] if o["ok"] else []))
    with open(os.path.join(G, "light_cases.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    oracle.build()
    pre_cases(); rm_test(); net_blocks(); nms_cases(); pnp_cases(); light_cases()
    print("golden vectors written to", G)
