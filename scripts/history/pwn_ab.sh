#!/bin/bash
cd /root/repo
export IRMV_BENCH_SKIP=h2d,latency,config1,config4
run() { env "$@" timeout -k 10 300 python3 bench.py --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])
"; }
for i in 1 2; do
run X=default
run IRMV_NO_PWN=1
run IRMV_STREAMS=1
done
