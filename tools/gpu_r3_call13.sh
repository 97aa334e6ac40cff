#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_shim_gpu.py -m gpu -q 2>&1 | tail -n 4
bash tools/ab/run_multi.sh 2 "strict-split2|-|" "strict-split4|-|GSLIC_FWD_SPLIT=4" "fast-split2|-|GSLIC_FAST_MATH=1" "fast-split4|-|GSLIC_FAST_MATH=1 GSLIC_FWD_SPLIT=4"
BENCH_ARGS="--gaussians 5000000 --width 3840 --height 2160 --steps 30" bash tools/ab/run_multi.sh 1 "4k-strict-split1|-|" "4k-strict-split2|-|GSLIC_FWD_SPLIT=2" "4k-fast-split1|-|GSLIC_FAST_MATH=1" "4k-fast-split2|-|GSLIC_FAST_MATH=1 GSLIC_FWD_SPLIT=2"
} > gpurun_out/r03_call13.log 2>&1
cat gpurun_out/r03_call13.log
