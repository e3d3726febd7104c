"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM traffic per launch.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request of a wide
(16 B/lane) coalesced read -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both are in KiB.
Usage: collect_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
"""
import collections, csv, json, re, sys


def internal_name(sym):
    """kernel symbol -> the name bench.py / irmv_engine_profile report"""
    m = (re.search(r"conv3x3_lds_kernel<(\d), (\d), (\d), (true|false), (\d)>", sym) or
         re.search(r"conv3x3_lds_kernelILi(\d)ELi(\d)ELi(\d)ELb([01])ELi(\d)E", sym))
    if m:   # images-per-workgroup is a launch argument, not part of the symbol: the "_iN" suffix of the bench name is dropped
        return f"conv3x3s{m.group(1)}_lds_mt{m.group(2)}_nt{m.group(3)}" + ("+1x1" if m.group(5) != "0" else "")
    m = re.search(r"conv_mfma_kernel<(\d), (\d), (\d), (\d), (true|false), (\d), (true|false)>", sym)
    if m:
        ks, st, mt, nt, c16, act, f32 = m.groups()
        return f"conv{ks}x{ks}s{st}_mt{mt}_nt{nt}" + ("_c16" if c16 == "true" else "") + ("_f32" if f32 == "true" else "")
    for k, v in (("front_kernel", "front_fused"), ("c2f2_kernel", "c2f2_fused"), ("light_extract_kernel", "light_extract"), ("preprocess_kernel", "preprocess"), ("conv0_kernel", "conv0_mfma"), ("sppf_pool", "sppf_pool"), ("decode_kernel", "decode"), ("nms_pnp_kernel", "nms_pnp")):
        if k in sym:
            return v
    return sym


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


f, w = load(sys.argv[1]), load(sys.argv[2])
acc = {}
for k in f:
    if "irmv" not in k:
        continue
    a = acc.setdefault(internal_name(k), dict(symbols=[], n=0, fetch=0.0, write=0.0))   # several symbols can share a bench name
    a["symbols"].append(k)
    a["n"] += len(f[k])
    a["fetch"] += sum(f[k])
    a["write"] += sum(w.get(k, [])) * (len(f[k]) / max(len(w.get(k, [])), 1))
out = {}
for name, a in acc.items():
    fs, ws = a["fetch"] / a["n"], a["write"] / a["n"]
    out[name] = dict(symbols=sorted(a["symbols"]), launches=a["n"], fetch_kib_raw=round(fs, 1), write_kib=round(ws, 1),
                     hbm_bytes_per_launch=int((2.0 * fs + ws) * 1024))
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(f"{len(out)} kernels -> {sys.argv[3]}")
