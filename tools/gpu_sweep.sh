#!/bin/bash
# chain-length sweep of the default bench:  bash tools/gpu_sweep.sh "1 2 4 8"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -n 3
for k in ${1:-"1 8"}; do
  echo "== chain $k"; GSLIC_BWD_CHAIN=$k timeout 300 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'views/s', d['ms_per_step'], 'ms', {k: v for k, v in d['kernel_ms_per_step'].items() if k in ('render_bwd', 'render_fwd', 'preprocess_bwd')}, 'dom', d['roofline']['kernel'], d['roofline']['avg_launch_ms'])
"
done
