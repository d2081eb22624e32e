#!/bin/bash
# Timing experiments on the 64-channel tile (conv3x3_c64_t8x32, 18 % of the 96-view step) and the dominant 128-channel
# tile: builds with parts of the kernel switched off (wrong results, timing only).  usage: tools/experiments/c64_ablate.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/mvlm_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Wno-unused-value"
OTHERS=$(ls build/*.o | grep -v -e conv_inst_g5.o -e conv_inst_g0.o)
SHAPES="96,128,64,128,3,10 96,64,64,128,1,10 96,128,64,64,3,10 96,64,64,64,1,10 96,256,128,128,3,0 96,256,128,64,3,0"
for V in NONE NO_STAGING NO_LOADS NO_WRITES NO_BARRIER NO_EPILOGUE; do
  D=""; [ $V != NONE ] && D="-DMVLM_ABLATE_$V"
  for g in 5 0; do /opt/rocm/bin/hipcc $FLAGS $D -c conv_inst_g$g.hip -o /tmp/conv_inst_g${g}_$V.o & done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvlm_$V.so /tmp/conv_inst_g5_$V.o /tmp/conv_inst_g0_$V.o $OTHERS || exit 1
  echo "== $V"
  MVLM_HIP_LIB=/tmp/libmvlm_$V.so python3 $ROOT/tools/conv_shape_bench.py $SHAPES 2>&1 | grep rc
done
