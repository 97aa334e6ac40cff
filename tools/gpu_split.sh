#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for cfg in "--scene lidar --gaussians 500000" "--gaussians 5000000 --width 3840 --height 2160 --steps 30" ""; do
for sp in 1 2 4; do
GSLIC_FWD_SPLIT=$sp timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras $cfg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_step_instrumented']; print('$cfg split $sp', d['value'], 'views/s', {n: k[n] for n in ('render_bwd', 'render_fwd')})
"
done; done
