#!/bin/bash
# parity suites + bench lines under a list of environment settings ("VAR=val" per argument; "-" = defaults)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -n 5
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  env $envs timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_step_instrumented']; print('$cfg', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('render_bwd', 'preprocess_bwd', 'render_fwd', 'preprocess')}, 'dom', d['roofline']['kernel'], d['roofline']['avg_launch_ms'])
"
done
