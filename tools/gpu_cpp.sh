#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_shim_gpu.py -m gpu -x -q 2>&1 | tail -n 8
timeout 600 python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'views/s', d['ms_per_step'], 'ms', 'other', d['other_host_path'], 'cpp', d['cpp_fused_host'], 'graph', d['graphed'])
"
