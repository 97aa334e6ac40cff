#!/bin/bash
# round 3, fourth GPU call: pixel-major decision masks; new bench.py legs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== A/B (2 rounds)"
bash tools/ab/run_multi.sh 2 "new2-strict|tools/ab/libgslic_hip_new2.so|" "new3-strict|tools/ab/libgslic_hip_new3.so|" "new4-strict|tools/ab/libgslic_hip_new4.so|" "new4-fast|tools/ab/libgslic_hip_new4.so|GSLIC_FAST_MATH=1"
echo "== 5M / 4K"
BENCH_ARGS="--gaussians 5000000 --width 3840 --height 2160 --steps 30" bash tools/ab/run_multi.sh 1 "new4-strict-4k|tools/ab/libgslic_hip_new4.so|" "new4-fast-4k|tools/ab/libgslic_hip_new4.so|GSLIC_FAST_MATH=1"
echo "== parity"
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fullsize_reference_gpu.py tests/test_fused_gpu.py tests/test_capacity_graph_gpu.py -m gpu -q 2>&1 | tail -n 15
echo "== fuzz (150 scenes, the seed of call 3)"
timeout 900 python tests/fuzz_vs_reference.py 150 5000 2>&1 | grep -v "strict OK" | tail -n 12
echo "== default bench line"
timeout 600 python bench.py 2>&1 | tail -n 3
echo "== dense workload"
timeout 600 python bench.py --density 2.5 --opacity-shift -2 --steps 50 --no-cpu-baseline --no-extras --profile-all 2>&1 | tail -n 2
} > gpurun_out/r03_call4.log 2>&1
tail -n 60 gpurun_out/r03_call4.log
