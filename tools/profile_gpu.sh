#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC passes of bench.py.
# usage: WORKLOAD="bu3dfe-rgbd-96:96v/gpu" tools/profile_gpu.sh <tag> [bench args...]
# (WORKLOAD = bench.py's workload key "<config>:<views per GPU>v/gpu" of the profiled command; it is stored in
#  traffic.json so that bench.py only quotes HBM traffic measured on the workload it is running)
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp MVLM_BENCH_NO_INGEST=1 MVLM_BENCH_LIVE_TRAFFIC=0
ARGS="--steps 3 --warmup 2 --cpu-views 0 --no-fast-mode $*"
echo "== kernel trace" | tee -a $OUT/log.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "rc=$?" >> $OUT/log.txt
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
head -40 $OUT/kernel_stats.csv
if [ "${PMC:-1}" = "1" ]; then
  echo "== pmc sq" | tee -a $OUT/log.txt
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
  echo "rc=$?" >> $OUT/log.txt
  echo "== pmc fetch" | tee -a $OUT/log.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
  echo "rc=$?" >> $OUT/log.txt
  echo "== pmc write" | tee -a $OUT/log.txt
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
  echo "rc=$?" >> $OUT/log.txt
  python3 $ROOT/tools/summarize_pmc.py $OUT "${WORKLOAD:-bu3dfe-rgbd-96:96v/gpu}" > $OUT/pmc_summary.txt 2>&1
  cat $OUT/pmc_summary.txt
fi
# keep the merged-back payload small
find $OUT -name '*.csv' -size +8M -delete
du -sh $OUT
