#!/bin/bash
# Per-kernel times of the rasteriser (tools/raster_bench.py under rocprofv3 --kernel-trace): tools/raster_trace.sh <tag> [views]
# -> gpurun_out/<tag>/: the bench's own lines and the mean duration of every kernel per (mesh, shading) case.
TAG=${1:-ras}; VIEWS=${2:-96}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/raster_bench.py $VIEWS 2>&1 | grep "grid" | tee $OUT/bench.txt
cd $ROOT
python3 - "$OUT" <<'EOF' | tee $OUT/kernels.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = ["transform_kernel", "cull_kernel", "classify_kernel", "scan_kernel", "fill_kernel", "bin_fill_kernel", "tile_kernel", "fillBuffer", "copyBuffer"]
per_case = 33  # 3 warm-up + 30 measured renders per case (raster_bench.py)
n_cases = max(sum(1 for r in rows if "tile_kernel" in r["Kernel_Name"]) // per_case, 1)
for case in range(n_cases):
    acc, cnt = {n: [] for n in names}, {n: 0 for n in names}
    for r in rows:
        for n in names:
            if n in r["Kernel_Name"]:
                per = per_case * (2 if n == "fillBuffer" else 1)
                cnt[n] += 1
                k = cnt[n] - case * per
                if per // 3 < k <= per:
                    acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("case", case, " ".join(f"{n.replace('_kernel', '')}:{sum(v) / len(v):.1f}" for n, v in acc.items() if v))
EOF
find $OUT -name "*.csv" -size +1M -delete
