#!/bin/bash
# What would a conv11 kernel gain that stages its input tile once for all four parities?  Timing-only builds of the 2x2 parity
# tile (conv2x2_c84_t8x32, variant 29; wrong results) with parts switched off, conv11's launch time read from a network pass.
set -u
export MVLM_BENCH_LIVE_TRAFFIC=0
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/mvlm_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Wno-unused-value"
OTHERS=$(ls build/*.o | grep -v -e conv_inst_g14.o)
for V in NONE NO_STAGING NO_LOADS NO_WRITES NO_BARRIER; do
  D=""; [ $V != NONE ] && D="-DMVLM_ABLATE_$V"
  /opt/rocm/bin/hipcc $FLAGS $D -c conv_inst_g14.hip -o /tmp/conv_inst_g14_$V.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvlm_$V.so /tmp/conv_inst_g14_$V.o $OTHERS || exit 1
  echo "== $V"
  MVLM_HIP_LIB=/tmp/libmvlm_$V.so MVLM_BENCH_PER_LAYER=1 MVLM_BENCH_NO_INGEST=1 python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-views 0 --no-fast-mode 2>&1 >/dev/null | grep "conv11\|conv2x2"
done
