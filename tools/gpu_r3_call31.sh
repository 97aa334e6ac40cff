#!/bin/bash
# sh_grad_from_rgb (the N > 1 step's SH rebuild + Adam) with 32 Gaussians per wave (6 KB of LDS, 20 waves per CU): tests + one-rank RCCL A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_rank1_exchange_gpu.py tests/test_dist_gpu.py tests/test_shim_gpu.py -m gpu -q 2>&1 | tail -n 4
port=29600
for r in 1 2 3; do
for spec in "prev|tools/ab/libgslic_hip_prev.so" "sgr32|-"; do
  IFS='|' read -r label lib <<< "$spec"
  if [ "$lib" = "-" ]; then libenv="X=1"; else libenv="GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/$lib"; fi
  port=$((port+1))
  env $libenv GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RANK=0 WORLD_SIZE=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras --profile-all 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$label', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('adam', 'preprocess_bwd', 'render_bwd') if n in k}, d['exchange']['exchange_window_ms'])
"
done
done
} > gpurun_out/r03_call31.log 2>&1
cat gpurun_out/r03_call31.log
