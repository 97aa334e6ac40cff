#!/bin/bash
# round 6: the group-outer row-scan backward (5 waves per SIMD, no spills) against the in-tree kernel, parity first; the step-79 stall diagnosis
#   gpurun --timeout 1200 -- 'bash tools/gpu_r06_bwd.sh r06e'
set -u
TAG=${1:-r06x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
GSLIC_HIP_LIB=$R/tools/ab/libgslic_hip_go.so timeout 600 python -m pytest tests/test_timed_path_reference_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $OUT/${TAG}_go_parity.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_go_parity.log
tail -3 $OUT/${TAG}_go_parity.log
bash tools/ab/run_multi.sh 3 "base|-|" "group-outer-5-waves|tools/ab/libgslic_hip_go.so|" > $OUT/${TAG}_ab.log 2>&1
cat $OUT/${TAG}_ab.log
BENCH_PROFILE="" bash tools/ab/run_multi.sh 3 "base|-|" "group-outer-5-waves|tools/ab/libgslic_hip_go.so|" > $OUT/${TAG}_ab_unprofiled.log 2>&1
cat $OUT/${TAG}_ab_unprofiled.log
for m in steps count async; do
  GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + RANDOM % 300)) RANK=0 WORLD_SIZE=1 timeout 200 python tools/diag_dist_stall.py $m 400 2>/dev/null | grep -v "^\[" >> $OUT/${TAG}_dist_stall.log
done
cat $OUT/${TAG}_dist_stall.log
