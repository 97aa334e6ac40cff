#!/bin/bash
# SSIM / loss kernels with the window weights in VGPRs: tests + same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_eval_gpu.py tests/test_training_trajectory_gpu.py -m gpu -q 2>&1 | tail -n 4
export BENCH_ARGS=""
for r in 1 2 3; do
for spec in "prev|tools/ab/libgslic_hip_prev.so" "taps-vgpr|-"; do
  IFS='|' read -r label lib <<< "$spec"
  if [ "$lib" = "-" ]; then libenv=""; else libenv="GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/$lib"; fi
  env $libenv timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras --profile-all 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$label', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: v for n, v in k.items() if 'loss' in n or 'ssim' in n or n in ('render_bwd', 'render_fwd')})
"
done
done
} > gpurun_out/r03_call19.log 2>&1
cat gpurun_out/r03_call19.log
