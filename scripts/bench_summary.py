import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    print(sys.argv[1] if len(sys.argv) > 1 else "", "B", d["config"]["frames_per_step_per_gpu"], "FPS", d["value"], "ms/step", d["ms_per_step"],
          "dom", r["kernel"], r["achieved"], "TF/s; all conv", r["all_conv_tflops"], "lat1", d.get("latency_ms_single_frame_h2d_inclusive"),
          "pcie", d.get("fps_pcie_inclusive_1gpu"))
