#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_rank1_exchange_gpu.py tests/test_dist_gpu.py tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -n 12
