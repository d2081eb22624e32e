# configs[4] (478 landmarks x 128 views, render + fusion, no network): the GPU timeline of one step - kernels and copies with
# the gaps between them - under rocprofv3 kernel + memory-copy tracing.  -> gpurun_out/mp_tl/last_steps.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mp_tl${1:-}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MVLM_BENCH_NO_INGEST=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config mediapipe-478x128 --steps 20 --warmup 5 --cpu-views 0 --no-live-traffic --no-kernel-profile > $OUT/bench.json 2> $OUT/bench.err
k=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
m=$(find $OUT/trace -name '*memory_copy_trace.csv' | head -1)
python3 - "$k" "$m" > $OUT/last_steps.txt <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
try:
    with open(sys.argv[2]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "?"))))
except Exception as e:
    print("no copy trace:", e)
rows.sort()
starts = [i for i, r in enumerate(rows) if "transform_kernel" in r[2]]
# the timed region = steps 8 .. 27 (3 set-up + 5 warm-up before); print steps 20 and 21 in full, and every step's span
spans = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    busy = sum(e - s for s, e, _ in seg)
    spans.append((rows[b][0] - rows[a][0], busy, len(seg)))
for i, (w, busy, n) in enumerate(spans):
    print(f"step {i:3d}: transform-to-transform {w / 1e3:8.1f} us, busy {busy / 1e3:7.1f} us, {n} items")
for which in (20, 21):
    if which + 1 >= len(starts):
        continue
    seg = rows[starts[which] - 3:starts[which + 1]]
    t0 = seg[0][0]
    prev = t0
    print(f"--- step {which}")
    for s, e, n in seg:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {n}")
        prev = max(prev, e)
PY
find $OUT/trace -name '*.csv' -size +4M -delete
tail -c 300 $OUT/bench.json
