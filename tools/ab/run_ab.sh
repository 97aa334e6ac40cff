#!/bin/bash
# A/B of two builds of the library on the SAME box (box-to-box spread is larger than most kernel changes):
#   gpurun -- 'bash tools/ab/run_ab.sh [rounds]'   with tools/ab/libgslic_hip_{A,B}.so in place
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
N=${1:-2}
for r in $(seq $N); do
  for v in A B; do
    GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libgslic_hip_$v.so timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras --profile-all 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$v', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('render_bwd', 'preprocess_bwd', 'render_fwd', 'preprocess', 'ssim_fwd', 'ssim_bwd')})
"
  done
done
