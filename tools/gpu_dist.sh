#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_shim_gpu.py tests/test_dist_gpu.py -m gpu -x -q 2>&1 | tail -n 12
timeout 600 python tools/exchange_bytes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exchange_bytes.txt
