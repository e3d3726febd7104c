#!/bin/bash
# single-frame latency legs of bench.py under one environment switch, same box: scripts/lat_ab.sh VAR v1 v2 ...
var=$1; shift
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for v in "$@"; do
  out=$(env $var=$v IRMV_BENCH_SKIP=h2d,config4 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1)
  echo "$var=$v: $(echo "$out" | python3 -c 'import sys, json; d = json.loads(sys.stdin.read()); print("one 1280x1024 frame at a time", d["latency_ms_single_frame_h2d_inclusive"], "ms =", d["fps_single_frames_in_flight"]["1"], "FPS; resident", d["latency_ms_single_frame_hbm_resident"], "ms; harness avg", d["latency_harness_ms"]["avg"], "; config1", d["config1"]["latency_ms_h2d_inclusive"], "ms =", d["config1"]["fps_one_frame_at_a_time"], "FPS, worst frame", d["config1"]["latency_ms_h2d_inclusive_min_max"][1])')"
done
