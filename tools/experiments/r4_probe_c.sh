#!/bin/bash
# Round 4: full GPU suite, then per-level tables at 12 / 8 / 96 views with the pair table in place
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r4c}
mkdir -p $OUT
cd $ROOT
python3 -c "
import importlib.util
spec = importlib.util.spec_from_file_location('bench', 'bench.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
print('visible_gpus() from sysfs:', m.visible_gpus())" > $OUT/visible_gpus.txt 2>&1
cat $OUT/visible_gpus.txt
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $OUT/tests.log 2>&1; rc=$?
tail -15 $OUT/tests.log
echo "pytest rc $rc"
export MVLM_BENCH_NO_INGEST=1
for v in 12 8; do
  MVLM_BENCH_PER_LAYER=1 timeout -k 10 300 python3 bench.py --config dtu3d-geomdepth-96 --views-total $v --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode > $OUT/bench_${v}views.json 2> $OUT/bench_${v}views.stderr.txt || exit 1
  python3 tools/per_level_table.py $OUT/bench_${v}views.stderr.txt > $OUT/per_level_${v}views.txt
  cat $OUT/per_level_${v}views.txt
done
MVLM_BENCH_PER_LAYER=1 timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --cpu-views 0 > $OUT/bench_96views.json 2> $OUT/bench_96views.stderr.txt || exit 1
python3 tools/per_level_table.py $OUT/bench_96views.stderr.txt > $OUT/per_level_96views.txt
cat $OUT/per_level_96views.txt
python3 -c "
import json
for v in (12, 8, 96):
    r = json.load(open('$OUT/bench_%dviews.json' % v))
    print(v, 'views:', r['ms_per_step'], 'ms', r['value'], 'views/s  all_conv_frac', r['roofline']['all_conv_frac'], 'dominant', r['roofline']['frac'])
"
python3 -c "
import json
r = json.load(open('$OUT/bench_96views.json'))
for k in ('fast_mode', 'fast16_mode'):
    print(k, json.dumps(r.get(k)))
"
