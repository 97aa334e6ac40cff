#!/bin/bash
# LDS diets decided by the occupancy sweep: preprocess (SH rows in two rounds of 32: 5.7 KB, 20 waves per CU), tile-sort scatter (one payload
# staging buffer: 38 KB, four workgroups per CU): full GPU suite + same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 8
for r in 1 2 3; do
for spec in "prev|tools/ab/libgslic_hip_prev.so" "diet|-"; do
  IFS='|' read -r label lib <<< "$spec"
  if [ "$lib" = "-" ]; then libenv="X=1"; else libenv="GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/$lib"; fi
  env $libenv timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras --profile-all 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$label', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('preprocess', 'sort_scatter', 'dsort_scatter', 'keybuild', 'render_fwd', 'render_bwd', 'preprocess_bwd') if n in k})
"
done
done
} > gpurun_out/r03_call25.log 2>&1
cat gpurun_out/r03_call25.log
