#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== new"; timeout 600 python -m pytest "tests/test_training_trajectory_gpu.py" -m gpu -q -x 2>&1 | grep -v "^$" | tail -n 40
echo "== prev"; GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libgslic_hip_prev.so timeout 600 python -m pytest "tests/test_training_trajectory_gpu.py" -m gpu -q 2>&1 | tail -n 5
echo "== new again"; timeout 600 python -m pytest "tests/test_training_trajectory_gpu.py" -m gpu -q 2>&1 | tail -n 5
} > gpurun_out/r03_call29.log 2>&1
cat gpurun_out/r03_call29.log
