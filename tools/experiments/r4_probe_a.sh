#!/bin/bash
# Round 4, first GPU pass: two-level grid barrier, tail-round measurement of the dominant tile, 12 / 8-view baselines
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4a
mkdir -p $OUT
cd $ROOT
timeout -k 10 120 tools/experiments/probes/bin/xcd_barrier_probe > $OUT/xcd_barrier_probe.txt 2>&1 || exit 1
# c128_t8x32 (variant 0) on 256 / 512 / 768 / 1024 tiles: what does the half-empty last round of 768 tiles cost?
timeout -k 10 300 python3 tools/conv_shape_bench.py 4,256,128,128,3,0 8,256,128,128,3,0 12,256,128,128,3,0 16,256,128,128,3,0 \
   4,128,128,128,3,0 8,128,128,128,3,0 12,128,128,128,3,0 16,128,128,128,3,0 \
   4,256,256,128,12,0 8,256,256,128,12,0 12,256,256,128,12,0 > $OUT/tail_round.txt 2>&1 || exit 1
export MVLM_BENCH_NO_INGEST=1
for v in 12 8; do
  MVLM_BENCH_PER_LAYER=1 timeout -k 10 300 python3 bench.py --config dtu3d-geomdepth-96 --views-total $v --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode > $OUT/bench_${v}views.json 2> $OUT/bench_${v}views.stderr.txt || exit 1
  python3 tools/per_level_table.py $OUT/bench_${v}views.stderr.txt > $OUT/per_level_${v}views.txt
done
cat $OUT/xcd_barrier_probe.txt $OUT/tail_round.txt $OUT/per_level_12views.txt
