#!/bin/bash
# Everything profiles/README.md cites for one round, in one pass on the GPU box: usage tools/collect_evidence.sh r04
# (results under gpurun_out/evidence_<tag>/ with the names they get in profiles/)
set -u
# PART=a: benches, parity reports, rehearsals; PART=b: the rocprofv3 passes and kernel traces; default both (a gpurun call ends
# after 20 minutes: the two halves fit one call each)
TAG=${1:-r04}
PART=${PART:-ab}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/evidence_$TAG
[[ $PART == *a* ]] && rm -rf $OUT
mkdir -p $OUT
cd $ROOT
if [[ $PART == *a* ]]; then
echo "== default bench" ; date
timeout -k 10 900 python3 bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.stderr.txt || exit 1
export MVLM_BENCH_NO_INGEST=1 MVLM_BENCH_LIVE_TRAFFIC=0
echo "== other configs" ; date
timeout -k 10 300 python3 bench.py --config bu3dfe-depth-8 --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode > $OUT/${TAG}_bench_b8views.json 2> $OUT/${TAG}_bench_b8views.stderr.txt || exit 1
timeout -k 10 300 python3 bench.py --config dtu3d-rgb-64 --steps 10 --warmup 3 --cpu-views 0 --no-fast-mode > $OUT/${TAG}_bench_dtu3d_rgb_64views.json 2> /dev/null || exit 1
timeout -k 10 300 python3 bench.py --config mediapipe-478x128 --steps 200 --warmup 5 --cpu-views 0 > $OUT/${TAG}_bench_mediapipe_478x128.json 2> /dev/null || exit 1
echo "== one GPU's share of the headline configuration at N = 2 / 4 / 8 (48 / 24 / 12 of the 96 views)" ; date
for v in 48 24 12; do
  timeout -k 10 300 python3 bench.py --views-total $v --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode > $OUT/${TAG}_bench_bu3dfe_rgbd_${v}views.json 2> /dev/null || exit 1
done
echo "== parity reports" ; date
timeout -k 10 600 python3 tests/reports/parity_stats.py 8 > $OUT/${TAG}_parity_stats.txt 2>&1 || exit 1
timeout -k 10 900 python3 tests/reports/e2e_parity_report.py 96 224 bu3dfe > $OUT/${TAG}_e2e_parity_96views.txt 2>&1 || exit 1
timeout -k 10 900 python3 tests/reports/e2e_parity_report.py 96 224 bu3dfe fast RGB+depth > $OUT/${TAG}_fast_vs_oracle_bu3dfe_rgbd_96views.txt 2>&1 || exit 1
timeout -k 10 900 python3 tests/reports/e2e_parity_report.py 64 224 dtu3d fast RGB > $OUT/${TAG}_fast_vs_oracle_dtu3d_rgb_64views.txt 2>&1 || exit 1
timeout -k 10 900 python3 tests/reports/e2e_parity_report.py 96 224 bu3dfe fast16 RGB+depth > $OUT/${TAG}_fast16_vs_oracle_bu3dfe_rgbd_96views.txt 2>&1 || exit 1
timeout -k 10 900 python3 tests/reports/e2e_parity_report.py 64 224 dtu3d fast16 RGB > $OUT/${TAG}_fast16_vs_oracle_dtu3d_rgb_64views.txt 2>&1 || exit 1
echo "== the renderer contract against a real OpenGL's rendering (tests/golden/gl_raster.npz), the reference's own textures" ; date
timeout -k 10 300 python3 tests/reports/gl_contract_report.py > $OUT/${TAG}_gl_contract.txt 2>&1 || exit 1
timeout -k 10 300 python3 tests/reports/gl_sensitivity_report.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_gl_sensitivity.txt || exit 1
timeout -k 10 300 python3 -m pytest tests/test_real_textures.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep "synchronisation rounds\|passed\|failed" > $OUT/${TAG}_real_textures.txt || exit 1
echo "== moment selection (fused) beside simple" ; date
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-views 0 --no-fast-mode --selection moment > $OUT/${TAG}_bench_moment_96views.json 2> /dev/null || exit 1
timeout -k 10 300 python3 bench.py --config dtu3d-geomdepth-96 --views-total 12 --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode --selection moment > $OUT/${TAG}_bench_moment_12views.json 2> /dev/null || exit 1
echo "== ingest segments, per-level tables" ; date
timeout -k 10 300 python3 tools/ingest_segments.py > $OUT/${TAG}_ingest_segments.txt 2>&1 || exit 1  # (JPEG on the device and on the host)
for v in 8 12; do
  MVLM_BENCH_PER_LAYER=1 timeout -k 10 300 python3 bench.py --config dtu3d-geomdepth-96 --views-total $v --steps 20 --warmup 5 --cpu-views 0 --no-fast-mode > $OUT/${TAG}_bench_dtu3d_${v}views.json 2> $OUT/${TAG}_bench_dtu3d_${v}views.stderr.txt || exit 1
  python3 tools/per_level_table.py $OUT/${TAG}_bench_dtu3d_${v}views.stderr.txt > $OUT/${TAG}_per_level_${v}views.txt
done
MVLM_BENCH_PER_LAYER=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-views 0 --no-fast-mode > /dev/null 2> $OUT/per_layer_96.err || exit 1
python3 tools/per_level_table.py $OUT/per_layer_96.err > $OUT/${TAG}_per_level_96views.txt
echo "== bench.py --gpus 5 as a plain process (five gloo ranks sharing this GPU: the box's process guard admits six processes), configs[2] / [3] / [4] at the 8-GPU shard sizes" ; date
timeout -k 10 900 bash tools/rehearsal.sh $TAG > $OUT/rehearsal_log.txt 2>&1 || exit 1
cp $ROOT/gpurun_out/rehearsal_$TAG/${TAG}_rehearsal_5ranks_*.json $OUT/
fi
if [[ $PART == *b* ]]; then
export MVLM_BENCH_NO_INGEST=1 MVLM_BENCH_LIVE_TRAFFIC=0
echo "== rocprofv3" ; date
WORKLOAD="bu3dfe-rgbd-96:96v/gpu" timeout -k 10 1200 bash tools/profile_gpu.sh $TAG > $OUT/profile_log.txt 2>&1 || exit 1
P=$ROOT/gpurun_out/prof_$TAG
cp $P/kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
cp $P/pmc_summary.txt $OUT/${TAG}_pmc_summary.txt
cp $P/traffic.json $OUT/${TAG}_traffic.json
cp $P/bench_trace.json $OUT/${TAG}_bench_under_rocprof.json
echo "== rocprofv3, the opt-in f16x2 precision" ; date
WORKLOAD="bu3dfe-rgbd-96:96v/gpu" timeout -k 10 900 bash tools/profile_gpu.sh ${TAG}_fast16 --precision fast16 > $OUT/profile_log_fast16.txt 2>&1 || exit 1
P=$ROOT/gpurun_out/prof_${TAG}_fast16
cp $P/kernel_stats.csv $OUT/${TAG}_fast16_kernel_stats.csv
cp $P/pmc_summary.txt $OUT/${TAG}_fast16_pmc_summary.txt
cp $P/bench_trace.json $OUT/${TAG}_fast16_bench_under_rocprof.json
echo "== rocprofv3, the 12-view shard (paired launches, split-K tiles)" ; date
WORKLOAD="dtu3d-geomdepth-96:12v/gpu" timeout -k 10 900 bash tools/profile_gpu.sh ${TAG}_12views --config dtu3d-geomdepth-96 --views-total 12 > $OUT/profile_log_12views.txt 2>&1 || exit 1
P=$ROOT/gpurun_out/prof_${TAG}_12views
cp $P/kernel_stats.csv $OUT/${TAG}_12views_kernel_stats.csv
cp $P/pmc_summary.txt $OUT/${TAG}_12views_pmc_summary.txt
cp $P/traffic.json $OUT/${TAG}_12views_traffic.json
echo "== the device JPEG decoder: 528 files + four 2048^2 textures against Pillow, its kernels under rocprofv3" ; date
timeout -k 10 600 bash tools/jpeg_evidence.sh > $OUT/jpeg_log.txt 2>&1 || exit 1
cp $ROOT/gpurun_out/jpeg/probe.txt $OUT/${TAG}_jpeg_probe.txt
(cat $ROOT/gpurun_out/jpeg/kernel_stats.txt; grep "^rc" $ROOT/gpurun_out/jpeg/profile_run.txt) > $OUT/${TAG}_jpeg_kernel_stats.csv
echo "== rasteriser alone (per-kernel times, six cases) and the kernels of configs[4]" ; date
timeout -k 10 300 bash tools/raster_trace.sh raster_$TAG > /dev/null 2>&1 || exit 1
(cat $ROOT/gpurun_out/raster_$TAG/bench.txt; echo; cat $ROOT/gpurun_out/raster_$TAG/kernels.txt) > $OUT/${TAG}_raster_trace.txt
timeout -k 10 300 bash tools/mediapipe_kernel_stats.sh > /dev/null 2>&1 || exit 1
cut -c1-260 $ROOT/gpurun_out/mp478/kernel_stats.csv | head -24 > $OUT/${TAG}_mediapipe_kernel_stats.csv
timeout -k 10 300 bash tools/mediapipe_timeline.sh _$TAG > /dev/null 2>&1 || exit 1
cp $ROOT/gpurun_out/mp_tl_$TAG/last_steps.txt $OUT/${TAG}_mediapipe_timeline.txt
fi
date; ls -la $OUT
