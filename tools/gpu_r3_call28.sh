#!/bin/bash
# SSIM / loss kernels: horizontal-pass results overwrite the halo rows they came from (forward 41.0 -> 35.6 KB, backward 37.3 -> 21.2 KB of LDS)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_eval_gpu.py tests/test_training_trajectory_gpu.py tests/test_shim_gpu.py tests/test_vs_reference_kernels_gpu.py -m gpu -q 2>&1 | tail -n 4
for cfg in "" "--gaussians 5000000 --width 3840 --height 2160 --steps 30"; do
for r in 1 2; do
for spec in "prev|tools/ab/libgslic_hip_prev.so" "alias|-"; do
  IFS='|' read -r label lib <<< "$spec"
  if [ "$lib" = "-" ]; then libenv="X=1"; else libenv="GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/$lib"; fi
  env $libenv timeout 300 python bench.py --steps 100 $cfg --no-cpu-baseline --no-extras --profile-all 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$label', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('ssim_fwd', 'ssim_bwd', 'render_fwd', 'render_bwd') if n in k})
"
done
done
done
} > gpurun_out/r03_call28.log 2>&1
cat gpurun_out/r03_call28.log
