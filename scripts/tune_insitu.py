#!/usr/bin/env python3
"""Stride-2 tiles measured IN the step, not alone: the per-layer autotuner times a conv in bursts on L2-warm data; in a step the
layer's input comes from memory after other layers have used the caches, so a tile that fetches fewer bytes (128 channels per
workgroup: the patch once; chunk-major: a chunk's weights once per 2 / 4 images) can lose alone and win in place.  For each
stride-2 entry of a tile table (batched, n128) every legal alternative is written into a copy of the table and timed by bench.py
(value = FPS, and the eager kernel sum); prints a line per trial.  All tiles are bitwise neutral.

    python3 scripts/tune_insitu.py profiles/r05_tune_cache.txt [rounds=2]
"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lines = [l.rstrip("\n") for l in open(src) if l.strip()]


def run(table):
    tmp = "/tmp/tune_insitu_cache.txt"
    open(tmp, "w").write("\n".join(table) + "\n")
    env = dict(os.environ, IRMV_TUNE_CACHE=tmp, IRMV_BENCH_SKIP="latency,h2d,config1,config4")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            return d["value"], d["roofline"]["step_kernel_ms_eager"]
    raise RuntimeError(p.stderr[-300:])


# flags: 1 lds, 64 cm2, 128 cm4, 256 w8   (engine.cpp tune_cache_save)
ALTS = [("w8 mt2 nt4 i4", "2 4 257 4"), ("w8 mt2 nt4 cm2", "2 4 321 2"), ("w8 mt1 nt4 cm4", "1 4 385 4"), ("w8 mt1 nt4 i4", "1 4 257 4"),
        ("nt8 i1", "1 8 257 1"), ("nt8 i2", "1 8 257 2"), ("nt8 cm2", "1 8 321 2"), ("mt1 nt4 cm4 (4 waves)", "1 4 129 4"), ("mt2 nt4 cm2 (4 waves)", "2 4 65 2")]
base = [run(lines) for _ in range(rounds)]
print("table as it is:", " ; ".join(f"{v:.0f} FPS, eager {s:.4f} ms" for v, s in base), flush=True)
for i, l in enumerate(lines):
    m = re.match(r"(gfx950\.t5\|3\.2\.0\.1\.0\.1\|(\d+x\d+>\d+x\d+)\|c(\d+)\.0\.0\.0>(\d+)\|.*\|n128\|0) (.*)$", l)
    if not m:
        continue
    key, geo, cin, cout, cur = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), m.group(5)
    print(f"--- {geo} c{cin}>{cout}: tuner's choice {cur}", flush=True)
    for name, alt in ALTS:
        if alt == cur or (alt.split()[1] == "8" and cout % 128 != 0):
            continue
        t = list(lines)
        t[i] = key + " " + alt
        try:
            res = [run(t) for _ in range(rounds)]
        except Exception as ex:
            print(f"    {name:24s} failed: {str(ex)[:80]}", flush=True)
            continue
        print(f"    {name:24s} " + " ; ".join(f"{v:.0f} FPS, eager {s:.4f} ms" for v, s in res), flush=True)
