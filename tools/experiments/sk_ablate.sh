#!/bin/bash
# Timing experiments on the split-K tiles of the small hourglass levels (run on the GPU box): builds of the split-K
# instantiation units with parts of the kernel switched off (wrong results, timing only).  usage: tools/experiments/sk_ablate.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/mvlm_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Wno-unused-value"
OTHERS=$(ls build/*.o | grep -v -e conv_inst_g9.o -e conv_inst_g10.o)
SHAPES="12,256,128,8,3,13/19 12,256,128,4,3,14/20 12,256,128,16,3,12/18 12,128,64,8,3,13/19 96,256,128,8,3,13/19"
for V in NONE NO_STAGING NO_LOADS NO_WRITES NO_BARRIER NO_EPILOGUE; do
  D=""; [ $V != NONE ] && D="-DMVLM_ABLATE_$V"
  for g in 9 10; do /opt/rocm/bin/hipcc $FLAGS $D -c conv_inst_g$g.hip -o /tmp/conv_inst_g${g}_$V.o & done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvlm_$V.so /tmp/conv_inst_g9_$V.o /tmp/conv_inst_g10_$V.o $OTHERS || exit 1
  echo "== $V"
  MVLM_HIP_LIB=/tmp/libmvlm_$V.so python3 $ROOT/tools/conv_shape_bench.py $SHAPES 2>&1 | grep rc
done
