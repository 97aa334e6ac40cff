#!/bin/bash
# render_fwd with one record set (no prefetch of the next entry): 64 VGPRs, 8 waves per SIMD instead of 72 / 7
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libgslic_hip_single.so timeout 900 python -m pytest tests/test_vs_reference_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q 2>&1 | tail -n 3
bash tools/ab/run_multi.sh 3 "double|-|" "single|tools/ab/libgslic_hip_single.so|" "double-fast|-|GSLIC_FAST_MATH=1" "single-fast|tools/ab/libgslic_hip_single.so|GSLIC_FAST_MATH=1"
BENCH_ARGS="--density 1.6 --opacity-shift -4" bash tools/ab/run_multi.sh 1 "faint-double|-|" "faint-single|tools/ab/libgslic_hip_single.so|"
} > gpurun_out/r03_call32.log 2>&1
cat gpurun_out/r03_call32.log
