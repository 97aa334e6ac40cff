#!/bin/bash
# round 2, GPU call A: full-size parity vs the reference kernels (fast + strict), knn goldens, new tests, pipeline model, baseline bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python tests/parity_report.py --out gpurun_out/parity_report.json --configs small,c2,c3,c5 ) > gpurun_out/parity_report.log 2>&1
echo "parity_report rc=$?"; tail -n 25 gpurun_out/parity_report.log
( timeout 300 python oracle/ref_build/make_golden.py gpurun_out/golden knn ) > gpurun_out/golden_knn.log 2>&1; echo "golden knn rc=$?"; tail -n 4 gpurun_out/golden_knn.log
( timeout 900 python -m pytest tests/test_eval_gpu.py tests/test_shim_gpu.py tests/test_camera_grad.py tests/test_vs_reference_kernels_gpu.py tests/test_ops_gpu.py tests/test_parity_gpu.py -m gpu -x -q -s ) > gpurun_out/tests_a.log 2>&1
echo "tests rc=$?"; tail -n 15 gpurun_out/tests_a.log
( timeout 600 python tools/bwd_util.py ) > gpurun_out/bwd_util.log 2>&1; echo "bwd_util rc=$?"; cat gpurun_out/bwd_util.log
( timeout 300 python bench.py --steps 50 --no-cpu-baseline ) > gpurun_out/bench_a.log 2>&1; echo "bench rc=$?"; tail -n 2 gpurun_out/bench_a.log
