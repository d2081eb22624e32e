# Round 6, review item 7: the 64-channel (and 128-channel) 8x32 tiles as persistent workgroups (MVLM_CONV_PERSIST), against
# NOTE: the kernel form this script switches on (MVLM_CONV_PERSIST, MVLM_CONV_PERSIST_ABLATE_PROLOGUE) was an experiment and has been removed
# again; it is in commit b6a0294 (mvlm_amd/csrc/conv_kernel.h).  Result: profiles/r06_persistent_tile_experiment.txt.
# the plain launch: single layers on idle data (A/B only), the whole 96-view step on real data, and the timing-only upper
# bound "as if a perfect prefetch hid every later tile's first-chunk round trip" (MVLM_CONV_PERSIST_ABLATE_PROLOGUE: wrong results).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/persist
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
SHAPES="96,128,64,128,3,10 96,64,64,128,3,10 96,64,64,256,3,10 96,128,64,64,3,10 96,256,128,128,3,1 96,256,256,128,12,1 96,256,128,64,3,1"
echo "== single layers, plain launch" > $OUT/layers.txt
python3 tools/conv_shape_bench.py $SHAPES 2>&1 | grep rc >> $OUT/layers.txt
echo "== single layers, persistent workgroups" >> $OUT/layers.txt
MVLM_CONV_PERSIST=3 python3 tools/conv_shape_bench.py $SHAPES 2>&1 | grep rc >> $OUT/layers.txt
echo "== single layers, persistent + no first-chunk staging after a workgroup's first tile (timing only)" >> $OUT/layers.txt
MVLM_CONV_PERSIST=3 MVLM_CONV_PERSIST_ABLATE_PROLOGUE=1 python3 tools/conv_shape_bench.py $SHAPES 2>&1 | grep rc >> $OUT/layers.txt
export MVLM_BENCH_NO_INGEST=1
for mode in plain persist ablate; do
  case $mode in
    plain) unset MVLM_CONV_PERSIST MVLM_CONV_PERSIST_ABLATE_PROLOGUE ;;
    persist) export MVLM_CONV_PERSIST=3 ;;
    ablate) export MVLM_CONV_PERSIST=3 MVLM_CONV_PERSIST_ABLATE_PROLOGUE=1 ;;
  esac
  python3 bench.py --steps 10 --warmup 3 --cpu-views 0 --no-fast-mode --no-live-traffic > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  echo "== $mode"; python3 -c "import json,sys; d=json.loads(open('$OUT/bench_$mode.json').read().strip().splitlines()[-1]); r=d['roofline']; print('views/s', d['value'], 'ms', d['ms_per_step'], 'dominant', r['kernel'], r['frac'], 'all conv', r['all_conv_frac'], 'conv ms', r['conv_ms_per_step'])"
  grep "launches/step" $OUT/bench_$mode.err | tail -14 | grep "c64_t8x32\|c128_t8x32\|c84\|c32_t16"
done
