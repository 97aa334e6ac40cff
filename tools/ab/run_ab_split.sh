#!/bin/bash
# like run_ab.sh, with Adam as its own launch (--split-adam: the adam_groups kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
N=${1:-2}
for r in $(seq $N); do
  for v in A B; do
    GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libgslic_hip_$v.so timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras --profile-all --split-adam 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$v split-adam', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('render_bwd', 'preprocess_bwd', 'adam', 'render_fwd')})
"
  done
done
