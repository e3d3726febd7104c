#!/bin/bash
cd /root/repo
export IRMV_TUNE_CACHE=/tmp/tc_xcd.txt; cp profiles/r04_tune_cache.txt $IRMV_TUNE_CACHE
for x in 1 0; do
  export IRMV_XCD_IMAGES=$x
  rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
  bash scripts/gpu_stage.sh traffic > /dev/null 2>&1
  cp gpurun_out/traffic.json gpurun_out/traffic_xcd$x.json
  python3 - <<PY
import json
t=json.load(open('gpurun_out/traffic_xcd$x.json'))
tot=sum(v['hbm_bytes_per_launch']*v['launches'] for k,v in t.items() if not k.startswith('_'))/9
print('XCD=$x total HBM MB per 128-frame step', round(tot/1e6,1), {k: round(t[k]['hbm_bytes_per_launch']/1e6,1) for k in ('conv3x3s2_lds_mt1_nt4','conv3x3s1_wres_pp','conv3x3s2_lds_mt1_nt8','conv3x3s1_lds_mt2_nt1','conv3x3s1_lds_mt4_nt4','conv3x3s2_lds_mt2_nt4','conv3x3s1_wres','conv1x1s1_pw_n2') if k in t})
PY
done
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
unset IRMV_TUNE_CACHE
for l in 1 0 1 0 1 0; do
IRMV_XCD_IMAGES=$l IRMV_BENCH_SKIP=h2d,latency,config1,config4 timeout -k 10 240 python3 bench.py --steps 80 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('xcd $l', d['value'], d['ms_per_step'], d['roofline']['step_kernel_ms_eager'])
"
done
