#!/bin/bash
# Round 4, second GPU pass: the pair kernel's tests, then the pair tuner
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4b
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 900 python3 tools/tune_conv_pairs.py --write-header --out $OUT/conv_pair_tune.json > $OUT/conv_pair_tune.txt 2>&1 || { tail -20 $OUT/conv_pair_tune.txt; exit 1; }
grep "^batch" $OUT/conv_pair_tune.txt
