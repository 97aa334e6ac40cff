#!/bin/bash
# config-5 shape (5M Gaussians, 4K): the scans over P as one chained launch (1221 tiles) instead of the three-launch fallback
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "large_scene or parity" 2>&1 | tail -n 3
for r in 1 2; do
for spec in "prev|tools/ab/libgslic_hip_prev.so" "new|-"; do
  IFS='|' read -r label lib <<< "$spec"
  if [ "$lib" = "-" ]; then libenv="X=1"; else libenv="GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/$lib"; fi
  env $libenv timeout 300 python bench.py --gaussians 5000000 --width 3840 --height 2160 --steps 30 --no-cpu-baseline --no-extras --profile-all 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$label', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('scan_apply', 'scan_reduce', 'scan_spine', 'bucket_count', 'preprocess', 'keybuild') if n in k})
"
done
done
} > gpurun_out/r03_call27.log 2>&1
cat gpurun_out/r03_call27.log
