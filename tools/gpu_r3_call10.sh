#!/bin/bash
# round 3, call 10: 8x8 quadrants + early-out in render_fwd
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== parity"
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fullsize_reference_gpu.py tests/test_fused_gpu.py tests/test_capacity_graph_gpu.py tests/test_camera_grad.py tests/test_extend.py tests/test_training_trajectory_gpu.py -m gpu -q 2>&1 | tail -n 15
echo "== fuzz"
timeout 900 python tests/fuzz_vs_reference.py 120 7000 2>&1 | grep -v "strict OK" | tail -n 6
echo "== A/B: ABI-5 build of the previous kernels is not available; same-box comparison by per-kernel HIP events against the default run"
bash tools/ab/run_multi.sh 2 "new5-strict|-|" "new5-fast|-|GSLIC_FAST_MATH=1"
BENCH_ARGS="--gaussians 5000000 --width 3840 --height 2160 --steps 30" bash tools/ab/run_multi.sh 1 "new5-strict-4k|-|" "new5-fast-4k|-|GSLIC_FAST_MATH=1"
BENCH_ARGS="--density 1.6 --opacity-shift -4 --steps 50" bash tools/ab/run_multi.sh 1 "faint-strict|-|" "faint-fast|-|GSLIC_FAST_MATH=1"
BENCH_ARGS="--scene lidar --gaussians 500000" bash tools/ab/run_multi.sh 1 "c2-strict|-|" "c2-fast|-|GSLIC_FAST_MATH=1"
} > gpurun_out/r03_call10.log 2>&1
cat gpurun_out/r03_call10.log | tail -40
