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
python3 $ROOT/tools/timeline_steps.py "$k" "$m" 20 > $OUT/last_steps.txt 2>&1
find $OUT/trace -name '*.csv' -size +4M -delete
tail -c 300 $OUT/bench.json
