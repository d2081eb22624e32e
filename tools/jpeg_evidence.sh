#!/bin/bash
# GPU box: the device JPEG decoder against Pillow (528 files + four 2048^2 textures) and its kernels under rocprofv3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/jpeg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $ROOT/tools/experiments/probes/jpeg_probe.py > $OUT/probe.txt 2>&1 || { tail -20 $OUT/probe.txt; exit 1; }
tail -8 $OUT/probe.txt
rm -rf $OUT/prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/tools/experiments/probes/jpeg_profile.py > $OUT/profile_run.txt 2>&1 || { tail $OUT/profile_run.txt; exit 1; }
grep "^rc" $OUT/profile_run.txt
for f in $(find $OUT/prof -name '*kernel_stats.csv'); do sed 's/(anonymous namespace):://g; s/([^"]*)//' $f | head -12 > $OUT/kernel_stats.txt; done
cat $OUT/kernel_stats.txt
find $OUT/prof -name '*kernel_trace.csv' -delete
