#!/bin/bash
# render_bwd: hit mask by v_bfi / v_bfe_i32 / v_and, offset add in place; strict render_fwd: in-place T / last updates; Horner power variant
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fullsize_reference_gpu.py tests/test_capacity_graph_gpu.py tests/test_fused_gpu.py -m gpu -q 2>&1 | tail -n 6
GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libgslic_hip_horner.so timeout 900 python -m pytest tests/test_vs_reference_kernels_gpu.py tests/test_fullsize_reference_gpu.py -m gpu -q 2>&1 | tail -n 4
bash tools/ab/run_multi.sh 3 "prev|tools/ab/libgslic_hip_prev.so|" "new|-|" "horner|tools/ab/libgslic_hip_horner.so|" "prev-fast|tools/ab/libgslic_hip_prev.so|GSLIC_FAST_MATH=1" "new-fast|-|GSLIC_FAST_MATH=1"
} > gpurun_out/r03_call18.log 2>&1
cat gpurun_out/r03_call18.log
