#!/bin/bash
# the growth-schedule leg is bimodal (2.2 vs 2.97 ms per iteration): three default runs with its per-kernel times and per-segment wall times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 1 2 3 4; do
timeout 300 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); g = d['growth_schedule']; print(d['value'], 'growth', {k: g[k] for k in g if k not in ('workload',)})
"
done > gpurun_out/r03_call23.log 2>&1
cat gpurun_out/r03_call23.log
