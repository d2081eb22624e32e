#!/bin/bash
# the 64-channel 8x32 tile with 8-channel chunks (variant 33) against the 4-channel-chunk tile (10): single layers on idle data, then
# inside the network (tune_in_network at 96 / 48 / 24 views without writing the table)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python3 tools/conv_shape_bench.py 96,128,64,128,3,10/33 96,64,64,128,3,10/33 96,128,128,128,3,10/33 96,64,64,256,3,10/33 96,84,256,128,6,10/33 96,128,64,64,3,10/33/28 24,128,64,128,3,10/33
python3 tools/tune_in_network.py --batches ${BATCHES:-96} --nets bu3dfe:RGB+depth --out gpurun_out/r05_c64k8_net.json 2>&1 | grep "==\|<--\|c64"
