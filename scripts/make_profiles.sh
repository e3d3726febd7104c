#!/bin/bash
# Regenerates the round's measurement artifacts on the GPU box (outputs under gpurun_out/; copy what is kept into profiles/).
#   gpurun --timeout 1100 -- 'bash scripts/make_profiles.sh'
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export IRMV_TUNE_CACHE=$O/tune_cache.txt
[ -f $R/profiles/r01_tune_cache.txt ] && cp $R/profiles/r01_tune_cache.txt $IRMV_TUNE_CACHE
echo "[1/4] bench"; python3 $R/bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
echo "[2/4] rocprofv3 kernel trace of the same command (tile choices from the cache, no single-frame extras)"
IRMV_BENCH_SKIP=latency rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/prof.log 2>&1; echo done
echo "[3/4] rocprofv3 --pmc FETCH_SIZE"
IRMV_STREAMS=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 64 --steps 2 > $O/pmc_f.log 2>&1; echo done
echo "[4/4] rocprofv3 --pmc WRITE_SIZE"
IRMV_STREAMS=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 64 --steps 2 > $O/pmc_w.log 2>&1; echo done
cd $R
python3 scripts/collect_traffic.py $O/pmc_f/*/*_counter_collection.csv $O/pmc_w/*/*_counter_collection.csv $O/traffic.json
ls $O/prof/*/ | head
