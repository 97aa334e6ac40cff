#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== tile order on / off: dense workload, 4K, default"
BENCH_ARGS="--density 2.5 --opacity-shift -2 --steps 50" bash tools/ab/run_multi.sh 2 "dense-order|-|" "dense-noorder|-|GSLIC_NO_TILE_ORDER=1"
BENCH_ARGS="--density 1.6 --opacity-shift -4 --steps 50" bash tools/ab/run_multi.sh 2 "faint-order|-|" "faint-noorder|-|GSLIC_NO_TILE_ORDER=1"
BENCH_ARGS="--gaussians 5000000 --width 3840 --height 2160 --steps 30" bash tools/ab/run_multi.sh 2 "4k-order|-|" "4k-noorder|-|GSLIC_NO_TILE_ORDER=1"
BENCH_ARGS="--scene lidar --gaussians 500000" bash tools/ab/run_multi.sh 2 "c2-order|-|" "c2-noorder|-|GSLIC_NO_TILE_ORDER=1"
} > gpurun_out/r03_call9.log 2>&1
cat gpurun_out/r03_call9.log
