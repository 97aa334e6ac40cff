#!/bin/bash
# tile counts in sorted order written by the last depth-sort pass (no gather in the offsets scan) + 1024-thread bucket scan: suite + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 8
for r in 1 2 3; do
for spec in "prev|tools/ab/libgslic_hip_prev.so" "new|-"; do
  IFS='|' read -r label lib <<< "$spec"
  if [ "$lib" = "-" ]; then libenv="X=1"; else libenv="GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/$lib"; fi
  env $libenv timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras --profile-all 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$label', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('scan_apply', 'dsort_scatter', 'bucket_count', 'preprocess', 'keybuild') if n in k})
"
done
done
} > gpurun_out/r03_call26.log 2>&1
cat gpurun_out/r03_call26.log
