#!/bin/bash
# Everything profiles/README.md cites for one round, in one pass on the GPU box: usage tools/collect_evidence.sh r02
# (results under gpurun_out/evidence_<tag>/ with the names they get in profiles/)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/evidence_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
echo "== default bench" ; date
timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.stderr.txt || exit 1
export MVLM_BENCH_NO_INGEST=1
echo "== other configs" ; date
timeout -k 10 300 python3 bench.py --config bu3dfe-depth-8 --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode > $OUT/${TAG}_bench_b8views.json 2> $OUT/${TAG}_bench_b8views.stderr.txt || exit 1
timeout -k 10 300 python3 bench.py --config dtu3d-geomdepth-96 --views-total 12 --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode > $OUT/${TAG}_bench_b12views.json 2> $OUT/${TAG}_bench_b12views.stderr.txt || exit 1
timeout -k 10 300 python3 bench.py --config dtu3d-rgb-64 --steps 10 --warmup 3 --cpu-views 0 --no-fast-mode > $OUT/${TAG}_bench_dtu3d_rgb_64views.json 2> /dev/null || exit 1
timeout -k 10 300 python3 bench.py --config mediapipe-478x128 --steps 20 --warmup 5 --cpu-views 0 > $OUT/${TAG}_bench_mediapipe_478x128.json 2> /dev/null || exit 1
echo "== parity reports" ; date
timeout -k 10 600 python3 tests/reports/parity_stats.py 8 > $OUT/${TAG}_parity_stats.txt 2>&1 || exit 1
timeout -k 10 900 python3 tests/reports/e2e_parity_report.py 96 224 bu3dfe > $OUT/${TAG}_e2e_parity_96views.txt 2>&1 || exit 1
echo "== rocprofv3" ; date
WORKLOAD="bu3dfe-rgbd-96:96v/gpu" timeout -k 10 1200 bash tools/profile_gpu.sh $TAG > $OUT/profile_log.txt 2>&1 || exit 1
P=$ROOT/gpurun_out/prof_$TAG
cp $P/kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
cp $P/pmc_summary.txt $OUT/${TAG}_pmc_summary.txt
cp $P/traffic.json $OUT/${TAG}_traffic.json
cp $P/bench_trace.json $OUT/${TAG}_bench_under_rocprof.json
date; ls -la $OUT
