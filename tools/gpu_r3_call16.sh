#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fullsize_reference_gpu.py tests/test_fused_gpu.py tests/test_capacity_graph_gpu.py tests/test_camera_grad.py -m gpu -q 2>&1 | tail -n 6
bash tools/ab/run_multi.sh 3 "prev-strict|tools/ab/libgslic_hip_prev.so|" "even-strict|-|" "prev-fast|tools/ab/libgslic_hip_prev.so|GSLIC_FAST_MATH=1" "even-fast|-|GSLIC_FAST_MATH=1"
} > gpurun_out/r03_call16.log 2>&1
cat gpurun_out/r03_call16.log
