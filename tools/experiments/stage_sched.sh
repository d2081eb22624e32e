#!/bin/bash
# Timing of the staging schedule inside a K-chunk (run on the GPU box): loads over the first 1/I of the k-steps, LDS writes
# over the last 1/W; builds of the c128 / c64 / c96 / c32 instantiation units per (I, W).  usage: tools/experiments/stage_sched.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/mvlm_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Wno-unused-value"
GROUPS_="0 1 5 7"
PAT=""; for g in $GROUPS_; do PAT="$PAT -e conv_inst_g$g.o"; done
OTHERS=$(ls build/*.o | grep -v $PAT)
SHAPES="96,256,128,128,3,0 96,256,256,128,12,0 96,128,64,128,3,10 96,64,64,128,1,10 96,64,64,64,1,10 96,256,84,128,4,1 96,64,32,256,3,3"
for V in "2 2" "3 3" "4 4" "3 2" "6 3" "4 2" "6 6"; do
  set -- $V
  TAG=i$1w$2
  OBJS=""
  for g in $GROUPS_; do /opt/rocm/bin/hipcc $FLAGS -DMVLM_STAGE_ISSUE_DIV=$1 -DMVLM_STAGE_WRITE_DIV=$2 -c conv_inst_g$g.hip -o /tmp/conv_inst_g${g}_$TAG.o & OBJS="$OBJS /tmp/conv_inst_g${g}_$TAG.o"; done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmvlm_$TAG.so $OBJS $OTHERS || exit 1
  echo "== issue 1/$1 write 1/$2"
  MVLM_HIP_LIB=/tmp/libmvlm_$TAG.so python3 $ROOT/tools/conv_shape_bench.py $SHAPES 2>&1 | grep rc
done
