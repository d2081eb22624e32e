set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tl12
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MVLM_BENCH_NO_INGEST=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config dtu3d-geomdepth-96 --views-total 12 --steps 6 --warmup 3 --cpu-views 0 --no-fast-mode --no-kernel-profile > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 $ROOT/tools/timeline_gaps.py $f 2000 > $OUT/gaps.txt 2>&1
python3 - "$f" > $OUT/last_step.txt <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-70:]))
rows.sort()
# last step = kernels after the last raster transform kernel start
idx = max(i for i, r in enumerate(rows) if "transform" in r[2])
step = rows[idx:]
t0 = step[0][0]
prev_end = t0
for s, e, n in step:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {n}")
    prev_end = max(prev_end, e)
PY
find $OUT/trace -name '*.csv' -size +4M -delete
cat $OUT/bench.json | cut -c1-300
