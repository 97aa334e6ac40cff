#!/bin/bash
# round 6: the timed configuration against the reference's kernels (tests + the table), on one box
#   gpurun --timeout 1500 -- 'bash tools/gpu_r06_parity.sh r06b'
set -u
TAG=${1:-r06x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_timed_path_reference_gpu.py tests/test_abi.py tests/test_binning_gpu.py tests/test_morton_order_gpu.py -x -q -m gpu -s > $OUT/${TAG}_timed_path_tests.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_timed_path_tests.log
tail -5 $OUT/${TAG}_timed_path_tests.log
timeout 600 python -m pytest tests/test_fuzz_vs_reference_gpu.py tests/test_pose_reference_gpu.py tests/test_fullsize_reference_gpu.py -x -q -m gpu -k "fuzz or c3" > $OUT/${TAG}_fuzz_forced_paths.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_fuzz_forced_paths.log
tail -5 $OUT/${TAG}_fuzz_forced_paths.log
timeout 600 python tests/parity_report.py --timed-path --out $OUT/${TAG}_parity_timed_path.json > $OUT/${TAG}_parity_timed_path.log 2>&1
echo "rc $?" >> $OUT/${TAG}_parity_timed_path.log
tail -25 $OUT/${TAG}_parity_timed_path.log
