#!/bin/bash
# round 3, third GPU call: strict backward on the forward's recorded decision bits
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== A/B (2 rounds)"
bash tools/ab/run_multi.sh 2 "new2-strict|tools/ab/libgslic_hip_new2.so|" "new3-strict|tools/ab/libgslic_hip_new3.so|" \
   "new2-fast|tools/ab/libgslic_hip_new2.so|GSLIC_FAST_MATH=1" "new3-fast|tools/ab/libgslic_hip_new3.so|GSLIC_FAST_MATH=1"
echo "== 5M / 4K"
BENCH_ARGS="--gaussians 5000000 --width 3840 --height 2160 --steps 30" bash tools/ab/run_multi.sh 1 "new2-strict-4k|tools/ab/libgslic_hip_new2.so|" "new3-strict-4k|tools/ab/libgslic_hip_new3.so|" "new3-fast-4k|tools/ab/libgslic_hip_new3.so|GSLIC_FAST_MATH=1"
echo "== parity"
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fullsize_reference_gpu.py tests/test_fused_gpu.py tests/test_capacity_graph_gpu.py tests/test_extend.py tests/test_training_trajectory_gpu.py -m gpu -q 2>&1 | tail -n 15
echo "== fuzz (150 scenes)"
timeout 900 python tests/fuzz_vs_reference.py 150 5000 2>&1 | tail -n 3
} > gpurun_out/r03_call3.log 2>&1
tail -n 40 gpurun_out/r03_call3.log
