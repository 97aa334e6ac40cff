#!/bin/bash
# round 6: the one-off 36-75 ms stall at the ~72nd-79th step of the N > 1 path (one-rank RCCL group) — what is it?
#   gpurun --timeout 900 -- 'bash tools/gpu_r06_stall.sh r06g'
set -u
TAG=${1:-r06x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
run() { echo "== mode $1" >> $OUT/${TAG}_dist_stall.log; GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + RANDOM % 300)) RANK=0 WORLD_SIZE=1 timeout 200 python tools/diag_dist_stall.py $1 2>/dev/null | grep -v "^\[\|version\|Hostname\|Librccl" >> $OUT/${TAG}_dist_stall.log; }
for m in steps snap gcfreeze gcoff steps; do run $m; done
cat $OUT/${TAG}_dist_stall.log
