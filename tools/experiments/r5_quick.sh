#!/bin/bash
# quick round-5 measurement set on the GPU box: small batches (12 / 8 views) with per-layer tables, fast16 per layer
set -u
export MVLM_BENCH_LIVE_TRAFFIC=0
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/quick_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MVLM_BENCH_NO_INGEST=1 MVLM_BENCH_PER_LAYER=1
for v in 12 8; do
  timeout -k 10 300 python3 bench.py --config dtu3d-geomdepth-96 --views-total $v --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode > $OUT/bench_dtu3d_${v}views.json 2> $OUT/bench_dtu3d_${v}views.stderr.txt || exit 1
  python3 tools/per_level_table.py $OUT/bench_dtu3d_${v}views.stderr.txt > $OUT/per_level_${v}views.txt
  cat $OUT/per_level_${v}views.txt
done
if [ "${FAST16:-1}" = "1" ]; then
timeout -k 10 300 python3 bench.py --precision fast16 --steps 10 --warmup 3 --cpu-views 0 > $OUT/bench_fast16.json 2> $OUT/bench_fast16.stderr.txt || exit 1
python3 -c "
import json;r=json.loads(open('$OUT/bench_fast16.json').read().strip().splitlines()[-1]);print('fast16', r['value'], r['ms_per_step'], r['roofline']['achieved'], r['roofline']['kernel_launches_per_step'], r['roofline']['conv_ms_per_step'])"
fi
if [ "${EXACT96:-0}" = "1" ]; then
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-views 0 --no-fast-mode > $OUT/bench_96.json 2> $OUT/bench_96.stderr.txt || exit 1
python3 -c "
import json;r=json.loads(open('$OUT/bench_96.json').read().strip().splitlines()[-1]);print('exact96', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['all_conv_frac'])"
fi
