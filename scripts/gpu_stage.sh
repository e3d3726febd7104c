#!/bin/bash
# Run GPU stages one after another on the box; a stage that is killed, times out or dies on a signal stops the chain
# (nothing more is started on a possibly wedged GPU); an ordinary non-zero exit (failed assertion) is recorded and the chain goes on.
#   gpurun --timeout 1100 -- 'bash scripts/gpu_stage.sh tests layers pmc bench'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
run() {   # run <name> <timeout s> <cmd...>
    local name=$1 to=$2; shift 2
    echo "=== stage $name: $*" | tee -a $O/stages.log
    timeout -k 10 $to "$@" > $O/$name.log 2>&1
    local rc=$?
    echo "=== stage $name exit $rc" | tee -a $O/stages.log
    tail -n 25 $O/$name.log
    if [ $rc -ge 124 ]; then echo "stage $name was killed / timed out / crashed: stopping the chain" | tee -a $O/stages.log; exit $rc; fi
    return 0
}
: > $O/stages.log
for st in "$@"; do
case $st in
tests)   run gpu_tests 900 python3 -m pytest tests -m gpu -x -q -rA --durations=15 ;;
testsall) run gpu_tests 1000 python3 -m pytest tests -m gpu -q -rA --durations=15 ;;
tests_k) run gpu_tests_k 600 python3 -m pytest tests -m gpu -x -q -rA -k "$TESTS_K" ;;
smoke)   run smoke 300 python3 __graft_entry__.py smoke ;;
layers)  SLOTS=64 run layers64 300 python3 scripts/prof_layers.py
         SLOTS=1 TOP=70 run layers1 200 python3 scripts/prof_layers.py ;;
bench)   run bench 600 python3 bench.py ;;
benchhost) for g in 16 32 64; do IRMV_BENCH_SKIP=latency run benchhost_$g 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --host-group $g; done ;;
bench2)  IRMV_DIST_BACKEND=gloo IRMV_FORCE_DEVICE=0 run bench2 500 python3 bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --frames-per-step 32 ;;
bench4)  run bench4 600 python3 bench.py --model shufflenet --net 416 --int8 --steps 100 --warmup 10   # BASELINE configs[4]
         grep '^{' $O/bench4.log | tail -1 > $O/bench4.json ;;
benchq)  run benchq 400 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline ;;
profiles) # everything the round's profiles/ are made of, one tile table for all of it (first use tunes and writes it)
         export IRMV_TUNE_CACHE=$O/tune_cache.txt; rm -f $IRMV_TUNE_CACHE
         run bench 700 python3 bench.py
         grep '^{' $O/bench.log | tail -1 > $O/bench.json
         cd /tmp; export TMPDIR=/tmp
         export IRMV_SYNC_LAUNCH=graph   # the profiled passes: no timing of the two single-frame launch forms at engine creation (96 single-frame steps whose
                                         # small launches would be averaged into the batched kernels' rows: same symbol names)
         IRMV_BENCH_SKIP=latency,h2d run prof_stats 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline
         IRMV_STREAMS=1 run pmc_f 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         IRMV_STREAMS=1 run pmc_w 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         IRMV_STREAMS=1 run pmc_a 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         IRMV_STREAMS=1 run pmc_b 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         cd $R
         python3 scripts/collect_traffic.py $O/pmc_f/*/*_counter_collection.csv $O/pmc_w/*/*_counter_collection.csv $O/traffic.json
         python3 scripts/collect_mfma.py $O/mfma.json $O/pmc_a/*/*_counter_collection.csv $O/pmc_b/*/*_counter_collection.csv
         cp $O/prof_stats/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
         python3 scripts/collect_concurrent.py $O/kernel_stats.csv $O/concurrent.json
         python3 scripts/timeline.py $O/prof_stats/*/*_kernel_trace.csv > $O/timeline.txt 2>&1; cat $O/timeline.txt
         cd /tmp
         IRMV_STREAMS=1 run prof_stats1 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats1 -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 20
         cd $R
         cp $O/prof_stats1/*/*_kernel_stats.csv $O/kernel_stats_single_stream.csv 2>/dev/null
         rm -rf $O/prof_stats1 $O/prof_stats $O/pmc_f $O/pmc_w $O/pmc_a $O/pmc_b     # raw traces: tens of MB; the reductions above are what is kept
         unset IRMV_TUNE_CACHE IRMV_SYNC_LAUNCH
         run bench4 600 python3 bench.py --model shufflenet --net 416 --int8 --steps 100 --warmup 10   # BASELINE configs[4], its own tiles
         grep '^{' $O/bench4.log | tail -1 > $O/bench4.json ;;
stats1)  export IRMV_TUNE_CACHE=$R/profiles/r05_tune_cache.txt   # single-stream eager trace with the committed tile table
         cd /tmp; export TMPDIR=/tmp
         IRMV_STREAMS=1 run prof_stats1 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats1 -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 20
         cd $R
         cp $O/prof_stats1/*/*_kernel_stats.csv $O/kernel_stats_single_stream.csv 2>/dev/null
         rm -rf $O/prof_stats1
         unset IRMV_TUNE_CACHE ;;
stats)   export IRMV_TUNE_CACHE=$O/tune_cache.txt
         cd /tmp; export TMPDIR=/tmp
         IRMV_BENCH_SKIP=latency,h2d run prof_stats 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline
         cd $R ;;
pmc)     cd /tmp; export TMPDIR=/tmp
         IRMV_STREAMS=1 run pmc_a 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         IRMV_STREAMS=1 run pmc_b 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         cd $R
         python3 scripts/collect_mfma.py $O/mfma.json $O/pmc_a/*/*_counter_collection.csv $O/pmc_b/*/*_counter_collection.csv ;;
traffic) cd /tmp; export TMPDIR=/tmp
         IRMV_STREAMS=1 run pmc_f 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         IRMV_STREAMS=1 run pmc_w 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         cd $R
         python3 scripts/collect_traffic.py $O/pmc_f/*/*_counter_collection.csv $O/pmc_w/*/*_counter_collection.csv $O/traffic.json ;;
probe)   run probe_build 120 /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 scripts/probes/stream_probe.cpp -o $O/stream_probe
         run stream_probe 200 $O/stream_probe ;;
trace1)  cd /tmp; export TMPDIR=/tmp
         TAG=trace run trace1 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- python3 $R/scripts/probe.py lat
         cd $R
         python3 - <<PY > $O/trace1_summary.txt 2>&1
import csv, glob, collections
f = glob.glob('$O/trace1/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 65-kernel-ish steps: take the final 2000 dispatches, report per-kernel mean duration and mean gap to the previous kernel
tail = rows[-2000:]
d = collections.defaultdict(list); g = []
for a, b in zip(tail, tail[1:]):
    g.append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
for r in tail:
    d[r['Kernel_Name'][:70]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
import statistics
print('mean kernel duration ns', statistics.mean(x for v in d.values() for x in v), 'median gap ns', statistics.median(g), 'mean gap ns (gaps < 20 us)', statistics.mean(x for x in g if x < 20000))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f'{k:72s} n={len(v):5d} mean {statistics.mean(v)/1000:7.2f} us')
PY
         cat $O/trace1_summary.txt | head -50 ;;
hostprobe) run host_probe 500 python3 scripts/probe.py host ;;
hostprobe2) run host_probe_plain 300 python3 scripts/probe.py host
         PROBE_TORCH=1 run host_probe_torch 300 python3 scripts/probe.py host
         cp profiles/r01_tune_cache.txt $O/tc_probe.txt; IRMV_TUNE_CACHE=$O/tc_probe.txt run host_probe_cache 300 python3 scripts/probe.py host ;;
crashprobe) cd /tmp; export TMPDIR=/tmp
         for v in inline nozc default; do
           case $v in default) E="";; inline) E="IRMV_INLINE_COPIES=1";; nozc) E="IRMV_ZERO_COPY_RESULTS=0";; esac
           echo "=== crashprobe $v" | tee -a $O/stages.log
           env TAG=$v $E timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/crash_$v -- python3 $R/scripts/probe.py crash > $O/crash_$v.log 2>&1
           rc=$?; echo "=== crashprobe $v exit $rc" | tee -a $O/stages.log
           grep "^\[" $O/crash_$v.log
           if [ $rc -ge 124 ]; then echo "crashprobe $v died: stopping the chain" | tee -a $O/stages.log; exit $rc; fi
         done
         cd $R ;;
s2probe) NET=416 run s2_probe 300 python3 scripts/probe.py s2 ;;
pmc1x1)  cd /tmp; export TMPDIR=/tmp
         IRMV_STREAMS=1 run pmc_pw_a 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_pw_a -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         IRMV_STREAMS=1 run pmc_pw_b 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/pmc_pw_b -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         IRMV_STREAMS=1 run pmc_pw_c 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/pmc_pw_c -- python3 $R/scripts/prof_step.py --mode eager --frames-per-step 128 --steps 2
         cd $R
         python3 - <<PY > $O/pmc_pw_summary.txt 2>&1
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$O/pmc_pw_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'pw_kernel' in k or 'conv_mfma_kernel<1, 1, 4, 4' in k or 'lds_kernelILi1ELi4ELi4ELb1ELi0' in k:
            acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f'   {n:32s} n={len(v):4d} mean {sum(v)/len(v):14.1f}')
PY
         cat $O/pmc_pw_summary.txt | head -80
         rm -rf $O/pmc_pw_a $O/pmc_pw_b $O/pmc_pw_c ;;
abwaves) for w in 3 4; do
           IRMV_EXTRA_HIPCC_FLAGS="-DIRMV_LDS_WAVES=$w" run build_w$w 600 python3 -c "from irmv_detection_amd import _build; import os; os.remove(_build.LIB_DIR + '/k_conv.o'); print(_build.build())"
           SLOTS=64 TOP=60 run layers64_w$w 300 python3 scripts/prof_layers.py
         done
         run build_w2 600 python3 -c "from irmv_detection_amd import _build; import os; os.remove(_build.LIB_DIR + '/k_conv.o'); print(_build.build())" ;;
tunev)   SLOTS=1 run tune_verbose 300 python3 scripts/probe.py tune ;;
stamps)  TAG=stamps IRMV_NMS_STAMPS=1 run nms_stamps 200 python3 scripts/probe.py lat ;;
headerr) run head_error 400 python3 scripts/probe.py headerr ;;
lat)     TAG=linear run lat_linear 200 python3 scripts/probe.py lat ;;
repro)   # the round-1 fault sequence, ONCE, under the profiler that exposed it, with every allocation range logged
         cd /tmp; export TMPDIR=/tmp
         IRMV_LOG_ALLOC=1 run repro_prof 300 rocprofv3 --kernel-trace --output-format csv -d $O/repro_prof -- python3 $R/scripts/probe.py repro
         cd $R ;;
*)       run "custom_$(echo "$st" | tr -c 'A-Za-z0-9_.\n' '_' | cut -c1-40)" 900 bash -c "$st" ;;
esac
done
echo "=== all stages done" | tee -a $O/stages.log
