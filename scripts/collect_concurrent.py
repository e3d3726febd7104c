"""rocprofv3 --kernel-trace --stats of the BENCHMARKED replay (bench.py: three concurrently replayed 64-frame graphs) ->
per bench-name launch statistics: what a launch lasts while the other graphs' kernels share the chip.

    collect_concurrent.py <kernel_stats.csv> <out.json>

bench.py reports `roofline.avg_launch_ms_concurrent` and the dominant symbol of the concurrent replay from this file
(the eager single-stream replay the roofline is measured on and the replay the headline times do not have
the same dominant symbol).
"""
import collections, csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collect_traffic_names import internal_name  # noqa: E402

acc = collections.defaultdict(lambda: dict(calls=0, total_ns=0.0, symbols=[]))
for r in csv.DictReader(open(sys.argv[1])):
    if "irmv" not in r["Name"]:
        continue
    a = acc[internal_name(r["Name"])]
    a["calls"] += int(r["Calls"]); a["total_ns"] += float(r["TotalDurationNs"]); a["symbols"].append(r["Name"])
tot = sum(a["total_ns"] for a in acc.values()) or 1.0
out = {k: dict(calls=a["calls"], avg_launch_ms=round(a["total_ns"] / a["calls"] * 1e-6, 6), share_of_kernel_time=round(a["total_ns"] / tot, 4),
               symbols=sorted(a["symbols"])) for k, a in acc.items()}
from build_stamp import build_stamp  # noqa: E402
json.dump(dict(build=build_stamp(), source="rocprofv3 --kernel-trace --stats -- python3 bench.py (benchmarked concurrent replay)", kernels=out), open(sys.argv[2], "w"), indent=1, sort_keys=True)
top = sorted(out.items(), key=lambda kv: -kv[1]["share_of_kernel_time"])[:6]
print(f"{len(out)} kernel names -> {sys.argv[2]}; top: " + ", ".join(f"{k} {v['share_of_kernel_time']:.3f} ({v['avg_launch_ms']*1e3:.1f} us)" for k, v in top))
