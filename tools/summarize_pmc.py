#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs per kernel name (sum over dispatches)."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])
WORKLOAD = sys.argv[2] if len(sys.argv) > 2 else "unknown"  # bench.py's workload key, e.g. "bu3dfe-rgbd-96:96v/gpu"
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in (out / sub).rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                k = k.replace("(anonymous namespace)::", "").replace("void ", "")
                k = k.split("(ConvArgs")[0].split("(rm_vert")[0].split("(float")[0].split("(double")[0][:90]
                k = k.replace(", false>, ", ">, ")  # Cfg<..., SPLITK=false> prints like the older 6-parameter form
                c = row.get("Counter_Name")
                v = float(row.get("Counter_Value", 0) or 0)
                agg[k][c] += v
                if c in ("SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
                    cnt[(k, c)] += 1
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    k = k.split("(ConvArgs")[0].split("(rm_vert")[0].split("(float")[0].split("(double")[0][:90]
    return k.replace(", false>, ", ">, ")


# wall time of every kernel in the SQ pass itself (its own kernel trace): effective clock = GRBM_GUI_ACTIVE / 8 / that time
# (MI355X_MICROARCH.md "DVFS give-back": the chip lowers its clock under load; within 3 % on dispatches of >= 10 ms, reads
# high below ~0.3 ms), and the share of the matrix pipes' cycles in which an MFMA was executing
dur_ns = defaultdict(float)
for f in (out / "pmc_sq").rglob("*kernel_trace.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            dur_ns[short(row.get("Kernel_Name", "?"))] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
names = sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0))
for k in names[:24]:
    a = agg[k]
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    print(f"{k}")
    print("   dispatches", cnt.get((k, "SQ_WAVES"), 0), " waves", int(a.get("SQ_WAVES", 0)),
          " wave_cycles(quad)", f"{wc:.3e}",
          " wait_any %.1f%%" % (100 * a.get("SQ_WAIT_ANY", 0) / wc),
          " wait_inst_any %.1f%%" % (100 * a.get("SQ_WAIT_INST_ANY", 0) / wc),
          " active_inst %.1f%%" % (100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc),
          " mfma_busy_cycles", f"{a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3e}",
          " busy_cu_cycles", f"{a.get('SQ_BUSY_CU_CYCLES', 0):.3e}",
          " lds_bank_conflict", f"{a.get('SQ_LDS_BANK_CONFLICT', 0):.3e}",
          " grbm_gui_active(sum 8 XCD)", f"{a.get('GRBM_GUI_ACTIVE', 0):.3e}")
    gui = a.get("GRBM_GUI_ACTIVE", 0) / 8
    if gui > 0:
        clk = f"{gui / dur_ns[k]:.2f} GHz over {dur_ns[k] * 1e-6:.2f} ms" if dur_ns.get(k) else "n/a"
        print("   derived: MFMA busy = mfma_busy_cycles / (grbm_gui_active / 8 x 1024 SIMDs) = %.3f; effective clock = grbm_gui_active / 8 / kernel time = %s"
              % (a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024), clk))
    if "FETCH_SIZE" in a or "WRITE_SIZE" in a:
        # FETCH_SIZE is in KB and, on gfx950, counts half the bytes of wide coalesced reads
        # (MI355X_MICROARCH.md "HBM"): doubled here.  WRITE_SIZE is exact for wide stores.
        print("   FETCH_SIZE KB raw %.0f -> x2 = %.1f MB   WRITE_SIZE KB %.0f = %.1f MB   (summed over %d / %d dispatches)" % (
            a.get("FETCH_SIZE", 0), 2 * a.get("FETCH_SIZE", 0) / 1024, a.get("WRITE_SIZE", 0), a.get("WRITE_SIZE", 0) / 1024,
            cnt.get((k, "FETCH_SIZE"), 0), cnt.get((k, "WRITE_SIZE"), 0)))

# per-launch HBM traffic of the conv variants, for bench.py's roofline.traffic
import json
import re
# "<template arguments of Cfg>" -> variant name, read from the variant table itself (conv_variants.h)
VARIANT = {}
for m in re.finditer(r'X\((\d+), "([^"]+)", Cfg<([^>]*)>\)', (Path(__file__).resolve().parents[1] / "mvlm_amd" / "csrc" / "conv_variants.h").read_text()):
    VARIANT[m[3]] = m[2]
traffic = {}
for k, a in agg.items():
    m = re.search(r"Cfg<([^>]*)>, (true|false)", k)
    if not m or m.group(1) not in VARIANT:
        continue
    if m.group(2) != "false" and VARIANT[m.group(1)] in traffic:
        continue
    nf, nw = cnt.get((k, "FETCH_SIZE"), 0), cnt.get((k, "WRITE_SIZE"), 0)
    if not nf or not nw:
        continue
    fetch = 2 * a["FETCH_SIZE"] * 1024 / nf   # gfx950: FETCH_SIZE counts half the bytes of wide reads
    write = a["WRITE_SIZE"] * 1024 / nw
    traffic[VARIANT[m.group(1)]] = {"launches": nf, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                                    "hbm_bytes_per_launch": fetch + write}
# the rasteriser: the five kernels of one mvlm_render together, per render call
RASTER = ("transform_kernel", "classify_kernel", "scan_kernel", "bin_fill_kernel", "tile_kernel")
rf = rw = 0.0
calls = 0
for k, a in agg.items():
    if any(k.startswith(r) or ("::" + r) in k or k.split("(")[0].endswith(r) for r in RASTER):
        rf += 2 * a.get("FETCH_SIZE", 0) * 1024
        rw += a.get("WRITE_SIZE", 0) * 1024
        if "tile_kernel" in k:
            calls = cnt.get((k, "WRITE_SIZE"), 0)
if calls:
    traffic["rasteriser"] = {"launches": calls, "fetch_bytes_per_launch": rf / calls, "write_bytes_per_launch": rw / calls,
                             "hbm_bytes_per_launch": (rf + rw) / calls}
(out / "traffic.json").write_text(json.dumps({"workload": WORKLOAD, "kernels": traffic}, indent=1, sort_keys=True))
