"""Per-kernel matrix-core utilisation from rocprofv3 --pmc passes (SQ counters only; --kernel-trace alongside).

    collect_mfma.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]

Every pass runs the same eager replay (scripts/prof_step.py --mode eager), so per-kernel counter sums of different
passes are comparable.  Reported per kernel name (the names bench.py / irmv_engine_profile use):

  mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)      share of the active CUs' SIMD-cycles in which the
                                                                            matrix pipe is busy (4 SIMDs per CU)
  mfma_util_gui  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024)    rocprofiler's own MfmaUtil formula (all 256 CUs x 4)
  valu_per_mfma  = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA          non-matrix vector instructions per MFMA
  wait / stall / issue shares of SQ_WAVE_CYCLES, LDS bank-conflict share of LDS-active cycles.
"""
import collections, csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collect_traffic_names import internal_name  # noqa: E402

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if "irmv" not in r["Kernel_Name"]:
            continue
        k = internal_name(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1

out = {}
tot_busy = tot_cu = 0.0
for k, c in sorted(acc.items()):
    n = max(cnt[k].values())
    per = {name: v / cnt[k][name] for name, v in c.items()}     # per launch
    d = dict(launches_seen=n)
    mf, cu = per.get("SQ_VALU_MFMA_BUSY_CYCLES"), per.get("SQ_BUSY_CU_CYCLES")
    if mf is not None and cu:
        d["mfma_util"] = round(mf / (4.0 * cu), 4)
    if mf is not None and per.get("GRBM_GUI_ACTIVE"):
        d["mfma_util_gui"] = round(mf / (per["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    nm = per.get("SQ_INSTS_MFMA") or per.get("SQ_INSTS_VALU_MFMA_MOPS_F16")
    if per.get("SQ_INSTS_MFMA") and per.get("SQ_INSTS_VALU"):
        d["valu_per_mfma"] = round((per["SQ_INSTS_VALU"] - per["SQ_INSTS_MFMA"]) / per["SQ_INSTS_MFMA"], 2)
    wc = per.get("SQ_WAVE_CYCLES")
    if wc:
        for nm_, key in (("SQ_WAIT_ANY", "wait_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"), ("SQ_ACTIVE_INST_ANY", "issue_share")):
            if nm_ in per:
                d[key] = round(per[nm_] / wc, 4)
    if per.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = round(per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"], 4)
    d["raw_per_launch"] = {name: round(v, 1) for name, v in sorted(per.items())}
    out[k] = d
    if mf is not None and cu and (k.startswith("conv") or k in ("front_fused", "c2f2_fused") or k.startswith("c2f") or k.startswith("kpt3") or k.startswith("bneck64")):
        tot_busy += mf * n
        tot_cu += cu * n
from build_stamp import build_stamp  # noqa: E402
res = dict(build=build_stamp(), conv_mfma_util=round(tot_busy / (4.0 * tot_cu), 4) if tot_cu else None,
           definition="sum over conv kernels of SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)", kernels=out)
json.dump(res, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(f"{len(out)} kernels, conv_mfma_util = {res['conv_mfma_util']} -> {sys.argv[1]}")
