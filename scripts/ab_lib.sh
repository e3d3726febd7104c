#!/bin/bash
# Two builds of libirmv_hip.so side by side on ONE box (boxes of the pool differ by +- 2 %): scripts/ab_lib.sh <old.so> [rounds=3]
# Each line: library, value (FPS, HBM-resident clock), ms per step, eager kernel-time sum per 128 frames.  The old library runs with
# the tile table of ITS round where the committed one does not match (it re-tunes on the spot).
old=$1; rounds=${2:-3}
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export IRMV_BENCH_SKIP=h2d,latency,config1,config4
for r in $(seq 1 $rounds); do
  for l in new old; do
    if [ $l = old ]; then export IRMV_LIB_PATH=$old; else unset IRMV_LIB_PATH; fi
    timeout -k 10 300 python3 bench.py --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$l round $r:', d['value'], 'FPS;', d['ms_per_step'], 'ms/step; eager kernel sum', d['roofline']['step_kernel_ms_eager'], 'ms')
"
  done
done
