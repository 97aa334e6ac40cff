#!/bin/bash
# parity suites + one default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -n 5
timeout 300 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'views/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'], 'dom', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'other', d['other_host_path'])
"
