#!/bin/bash
# A/B of experimental builds of conv_fast.hip inside the network (real data: the clock under load is part of the result):
# usage: tools/experiments/r5_fast_exp.sh "NONE SETPRIO ..."  -> bench.py --precision fast16 per build
set -u
export MVLM_BENCH_LIVE_TRAFFIC=0
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/mvlm_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Wno-unused-value"
OTHERS=$(ls build/*.o | grep -v conv_fast.o)
for V in ${1:-NONE SETPRIO}; do
  D=""; [ $V != NONE ] && D="-DMVLM_FAST_EXP_$V"
  /opt/rocm/bin/hipcc $FLAGS $D -c conv_fast.hip -o /tmp/conv_fast_$V.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvlm_$V.so /tmp/conv_fast_$V.o $OTHERS || exit 1
  for rep in 1 2; do
  MVLM_HIP_LIB=/tmp/libmvlm_$V.so MVLM_BENCH_NO_INGEST=1 python3 $ROOT/bench.py --precision fast16 --steps 10 --warmup 3 --cpu-views 0 2>/dev/null | python3 -c "
import json,sys;r=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$V', r['value'], 'views/s', r['ms_per_step'], 'ms; f16 kernel', r['roofline']['achieved'], 'TF-eq, conv ms', r['roofline']['conv_ms_per_step'])"
  done
done
