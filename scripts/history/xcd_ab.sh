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
alg={'front_fused':713e6,'c2f32_ab':262.4e6,'c2f2_fused':419e6}
print('XCD=$x', {k: round(t[k]['hbm_bytes_per_launch']/1e6,1) for k in ('c2f2_fused','c2f32_ab','c2f32_a','c2f32_b','front_fused')}, {k: round(t[k]['hbm_bytes_per_launch']/alg[k],3) for k in alg})
PY
  SLOTS=128 TOP=12 python3 scripts/prof_layers.py 2>/dev/null | grep -E "c2f|front|sum of|graph step"
done
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
