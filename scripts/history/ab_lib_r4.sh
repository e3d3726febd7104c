#!/bin/bash
cd /root/repo
export IRMV_BENCH_SKIP=h2d,latency,config1,config4
for l in new old new old new old; do
if [ $l = old ]; then export IRMV_LIB_PATH=/root/repo/build_probe/lib_r4a.so; else unset IRMV_LIB_PATH; fi
timeout -k 10 240 python3 bench.py --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$l', d['value'], d['ms_per_step'], d['roofline']['step_kernel_ms_eager'])
"
done
