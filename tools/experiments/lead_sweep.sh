#!/bin/bash
# MFMAs issued before a k-step's share of the staging work (MVLM_MFMA_LEAD): sweep on the big tiles.  usage: tools/experiments/lead_sweep.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/mvlm_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Wno-unused-value"
GROUPS_="0 1 5 7"
PAT=""; for g in $GROUPS_; do PAT="$PAT -e conv_inst_g$g.o"; done
OTHERS=$(ls build/*.o | grep -v $PAT)
SHAPES="96,256,128,128,3,0 96,256,256,128,12,0 96,128,64,128,3,10 96,64,64,128,1,10 96,256,84,128,4,1 96,64,32,256,3,3"
for L in 2 0 1 3 4 8; do
  OBJS=""
  for g in $GROUPS_; do /opt/rocm/bin/hipcc $FLAGS -DMVLM_MFMA_LEAD=$L -c conv_inst_g$g.hip -o /tmp/conv_inst_g${g}_l$L.o & OBJS="$OBJS /tmp/conv_inst_g${g}_l$L.o"; done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvlm_l$L.so $OBJS $OTHERS || exit 1
  echo "== lead $L"
  MVLM_HIP_LIB=/tmp/libmvlm_l$L.so python3 $ROOT/tools/conv_shape_bench.py $SHAPES 2>&1 | grep rc
done
