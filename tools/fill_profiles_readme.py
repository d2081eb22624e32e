#!/usr/bin/env python3
"""Fill the @PLACEHOLDERS@ of profiles/README.md's current-round section from the round's evidence files (profiles/rNN_*.json)."""
import json
import sys
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = Path(__file__).resolve().parents[1] / "profiles"
j = lambda n: json.loads((P / f"{tag}_{n}.json").read_text())
n1, v12, v8, b8 = j("bench_n1"), j("bench_dtu3d_12views"), j("bench_dtu3d_8views"), j("bench_b8views")
ro, rr = n1["roofline"], n1["roofline_rasteriser"]
vals = {
    "N1_VALUE": n1["value"], "N1_MS": n1["ms_per_step"], "N1_TF": ro["achieved"], "N1_FRAC": ro["frac"], "N1_KMS": ro["kernel_avg_ms"],
    "N1_ALLTF": ro["all_conv_kernels_tflops"], "N1_ALLFRAC": ro["all_conv_frac"], "RAS_GBS": rr["achieved"], "RAS_FRAC": rr["frac"],
    "CPU": n1["cpu_baseline"]["value"], "ING1": n1["with_ingest"]["single_file_views_per_s"], "INGF": n1["with_ingest"]["folder_views_per_s"],
    "FAST": n1["fast_mode"]["value"], "FAST16": n1["fast16_mode"]["value"], "FAST16_MS": n1["fast16_mode"]["ms_per_step"],
    "FAST16_TF": n1["fast16_mode"]["fast_kernel_fp32_equivalent_tflops"],
    "V12": v12["value"], "MS12": v12["ms_per_step"], "F12": v12["roofline"]["all_conv_frac"],
    "V8": v8["value"], "F8": v8["roofline"]["all_conv_frac"], "VB8": b8["value"], "FB8": b8["roofline"]["all_conv_frac"],
    "MOM96": j("bench_moment_96views")["value"], "MOM12": j("bench_moment_12views")["value"],
    "V64": j("bench_dtu3d_rgb_64views")["value"], "MP_MS": j("bench_mediapipe_478x128")["ms_per_step"],
}
readme = P / "README.md"
s = readme.read_text()
for k, v in vals.items():
    s = s.replace(f"@{k}@", str(v))
readme.write_text(s)
print({k: v for k, v in vals.items()})
