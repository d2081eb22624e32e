#!/bin/bash
# Rehearsal of the driver's multi-GPU command on the ONE-GPU box: `python bench.py --gpus N` as a plain process (it starts
# its N ranks itself), every rank on device 0 over gloo (MVLM_BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device).
# N = 5, not 8: the GPU pool's process guard admits at most six processes with the card open - five ranks and the
# torch.distributed.run agent that starts them (a run with six ranks was killed by it: "7 processes had the GPU open") - and
# an N = 8 run is the driver's to launch.  The per-rank shard sizes of the 8-GPU run are kept instead: configs[3] 12 views per
# rank (60 views in total), configs[4] 16 per rank (80 in total); configs[2] runs its 96 views on 5 ranks (19-20 per rank).
# usage: tools/rehearsal.sh <tag>   -> gpurun_out/rehearsal_<tag>/<tag>_rehearsal_5ranks_*.json (+ wall seconds of each whole command)
set -u
export MVLM_BENCH_LIVE_TRAFFIC=0
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/rehearsal_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
export MVLM_BENCH_SHARE_GPU=1 MVLM_BENCH_NO_INGEST=1
run() {  # name, bench arguments...
  local name=$1; shift
  local t0=$(date +%s.%N)
  timeout -k 10 900 python3 bench.py --gpus 5 --steps 5 --warmup 2 --cpu-views 0 --no-fast-mode "$@" > $OUT/${TAG}_rehearsal_5ranks_$name.json 2> $OUT/$name.err || return 1
  local t1=$(date +%s.%N)
  python3 - "$OUT/${TAG}_rehearsal_5ranks_$name.json" "$t0" "$t1" <<'PY'
import json, sys
f, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
rec = json.loads(open(f).read().strip().splitlines()[-1])
rec["whole_command_wall_s"] = round(t1 - t0, 1)
rec["rehearsal"] = "5 gloo ranks sharing ONE MI355X (process guard of the GPU pool: <= 6 processes with the card open, the launcher agent included); the driver's run is 8 RCCL ranks, one per GPU"
open(f, "w").write(json.dumps(rec) + "\n")
print(f, rec["value"], rec["unit"], "wall", rec["whole_command_wall_s"], "s", rec["scaling_breakdown"]["per_rank_ms_per_step"])
PY
}
run configs2_bu3dfe_rgbd_96views || exit 1
run configs3_dtu3d_geomdepth_12_per_rank --config dtu3d-geomdepth-96 --views-total 60 || exit 1
run configs4_mediapipe_478_16_per_rank --config mediapipe-478x128 --views-total 80 || exit 1
ls -la $OUT
