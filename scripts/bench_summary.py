import sys, json
# usage: bench_summary.py <bench.json> [label]   (a file, never stdin: a forgotten pipe must not hang a GPU run)
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(sys.argv[2] if len(sys.argv) > 2 else "", "B", d["config"]["frames_per_step_per_gpu"], "FPS", d["value"], "ms/step", d["ms_per_step"],
          "dom", r["kernel"], r["achieved"], r["unit"], "; all conv TF/s", r["all_conv_tflops"], "lat1", d.get("latency_ms_single_frame_h2d_inclusive"),
          "pcie", d.get("fps_pcie_inclusive_1gpu"))
