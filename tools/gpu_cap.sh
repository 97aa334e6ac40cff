#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_capacity_graph_gpu.py tests/test_parity_gpu.py tests/test_fused_gpu.py tests/test_extend.py -m gpu -x -q 2>&1 | tail -n 25
