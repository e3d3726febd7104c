"""Occupancy of the GPU over time from a rocprofv3 --kernel-trace CSV of the benchmarked replay: how much of the wall time
has 0 / 1 / 2 / 3+ kernels in flight, and which kernels run next to which.  (Diagnostic for DESIGN section 7.)
    timeline.py <kernel_trace.csv> [skip_fraction=0.5]
"""
import collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collect_traffic_names import internal_name
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "irmv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]            # steady state: the second half of the trace
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1, internal_name(r["Kernel_Name"])))
    ev.append((int(r["End_Timestamp"]), -1, internal_name(r["Kernel_Name"])))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
depth_time = collections.Counter()
alone = collections.Counter(); total = collections.Counter()
live = collections.Counter(); depth = 0; prev = t0
for t, d, name in ev:
    dt = t - prev
    depth_time[min(depth, 4)] += dt
    for k, c in live.items():
        if c > 0:
            total[k] += dt * c
            if depth == c:
                alone[k] += dt * c
    prev = t
    depth += d
    live[name] += d
wall = t1 - t0
print(f"{len(rows)} dispatches over {wall/1e6:.2f} ms")
for k in sorted(depth_time):
    print(f"  {k}{'+' if k == 4 else ' '} kernels in flight: {100.0*depth_time[k]/wall:5.1f} % of the wall time")
busy = sum(total.values())
print(f"  sum of kernel durations / wall = {busy/wall:.2f}")
print("kernel name                         share of summed duration   of which with no other kernel in flight")
for k, v in sorted(total.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {k:34s} {100.0*v/busy:5.1f} %   {100.0*alone[k]/v:5.1f} %")
