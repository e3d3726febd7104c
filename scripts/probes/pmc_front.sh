#!/bin/bash
# SQ counters of the front kernel, two builds of scripts/probes/front_probe.cpp (build_probe/front_probe_np, build_probe/front_probe_n)
O=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in np n; do
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmcf_$v -- $R/build_probe/front_probe_$v 64 > $O/pmcf_$v.log 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/pmcg_$v -- $R/build_probe/front_probe_$v 64 > $O/pmcg_$v.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out'
for v in ('np','n'):
    for pre in ('pmcf','pmcg'):
        acc=collections.defaultdict(float); n=collections.defaultdict(int)
        for f in glob.glob(f'{O}/{pre}_{v}/*/*counter_collection.csv'):
            for r in csv.DictReader(open(f)):
                if 'front' in r['Kernel_Name']:
                    acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
        print(v,pre,{k:round(acc[k]/max(n[k],1)) for k in acc}, 'launches',max(n.values()) if n else 0)
PY
