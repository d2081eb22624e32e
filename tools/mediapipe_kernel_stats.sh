set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mp478
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MVLM_BENCH_NO_INGEST=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --config mediapipe-478x128 --steps 20 --warmup 5 --cpu-views 0 > $OUT/bench.json 2> $OUT/bench.err
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
cut -c1-300 $OUT/bench.json; echo; cat $OUT/kernel_stats.csv | cut -c1-150 | head -20
find $OUT -name '*.csv' -size +4M -delete
