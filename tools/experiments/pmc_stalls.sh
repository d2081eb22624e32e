#!/bin/bash
# Where the waves of the dominant tiles wait, from SQ counters (run on the GPU box).  PC sampling is not available on
# this pool (gpurun refuses --pc-sampling-beta-enabled), so stall attribution stays at the counter level:
# separate --pmc passes with --kernel-trace only.  usage: tools/experiments/pmc_stalls.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_stalls
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHAPES="96,256,128,128,3,0 96,128,64,128,3,10"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/conv_shape_bench.py $SHAPES > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?" >> $OUT/log.txt
done
python3 - $OUT > $OUT/summary.txt 2>&1 <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_mfma_kernel" not in k:
            continue
        key = k[k.index("Cfg<"):k.index(">", k.index("Cfg<")) + 1]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
for key, c in acc.items():
    print(key)
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    for name in sorted(c):
        print(f"   {name:28s} {c[name]:.4e}   / wave_cycles {c[name] / wc:.3f}")
PY
cat $OUT/log.txt; cat $OUT/summary.txt
find $OUT -name '*.csv' -size +8M -delete
