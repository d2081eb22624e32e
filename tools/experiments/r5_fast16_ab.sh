#!/bin/bash
# A/B of the f16x2 kernel's tile forms inside the network (real data): MVLM_FAST16_HALF_TILES = 0 full tiles (round 4), 1 half
# tiles for the 128-channel layers, 3 for the 64-channel layers too.  usage: tools/experiments/r5_fast16_ab.sh "0 1 3"
set -u
export MVLM_BENCH_LIVE_TRAFFIC=0
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export MVLM_BENCH_NO_INGEST=1
for H in ${1:-0 1 3}; do
  for rep in 1 2; do
    MVLM_FAST16_HALF_TILES=$H python3 bench.py --precision fast16 --steps 10 --warmup 3 --cpu-views 0 2>/tmp/err_$H.txt | python3 -c "
import json,sys;r=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('half_tiles=$H', r['value'], 'views/s', r['ms_per_step'], 'ms; f16 kernel', r['roofline']['achieved'], 'TF-eq over', r['roofline']['kernel_launches_per_step'], 'launches; conv ms', r['roofline']['conv_ms_per_step'])"
  done
done
