#!/bin/bash
# round 3, first GPU call: the strict arithmetic on its new instruction sequence
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== expf replica"; timeout 120 tools/ubench/expf_replica
echo "== A/B (2 rounds)"
bash tools/ab/run_multi.sh 2 "r2-fast|tools/ab/libgslic_hip_r2.so|" "r2-strict|tools/ab/libgslic_hip_r2.so|GSLIC_STRICT_MATH=1" \
   "new-strict|tools/ab/libgslic_hip_new.so|" "new-strict-split1|tools/ab/libgslic_hip_new.so|GSLIC_FWD_SPLIT=1" "new-fast|tools/ab/libgslic_hip_new.so|GSLIC_FAST_MATH=1" \
   "sbranch-strict|tools/ab/libgslic_hip_sbranch.so|"
echo "== parity suites (default = strict)"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fused_gpu.py tests/test_capacity_graph_gpu.py -m gpu -q 2>&1 | tail -n 15
echo "== full-size parity report + reference-vs-reference yardstick"
timeout 1500 python tests/parity_report.py --out gpurun_out/r03_parity_fullsize.json 2>&1 | tail -n 40
echo "== fuzz (100 scenes)"
timeout 900 python tests/fuzz_vs_reference.py 100 3000 2>&1 | tail -n 4
} > gpurun_out/r03_call1.log 2>&1
tail -n 80 gpurun_out/r03_call1.log
