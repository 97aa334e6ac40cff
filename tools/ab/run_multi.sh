#!/bin/bash
# Same-box comparison of any number of library builds / environment settings (box-to-box spread is larger than most kernel changes):
#   gpurun -- 'bash tools/ab/run_multi.sh <rounds> "label|lib-or-'-'|ENV=val ENV2=val ..." ...'
# BENCH_PROFILE="" drops --profile-all (a HIP-event pair around every launch changes how consecutive kernels overlap: use it for the step time).
# lib = a path relative to the repository root, "-" = the in-tree library.  One line per run: views/s and the per-launch kernel times.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
N=$1; shift
for r in $(seq $N); do
  for spec in "$@"; do
    IFS='|' read -r label lib envs <<< "$spec"
    if [ "$lib" = "-" ] || [ -z "$lib" ]; then libenv=""; else libenv="GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/$lib"; fi
    env $libenv $envs timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras ${BENCH_PROFILE---profile-all} ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed'] or {}; print('$label', d['value'], 'views/s', d['ms_per_step'], 'ms', {n: k[n] for n in ('render_bwd', 'preprocess_bwd', 'render_fwd', 'preprocess', 'keybuild', 'sort_scatter', 'tile_hist', 'tile_scan', 'tile_bin', 'tile_lsort', 'tile_lsort_long', 'ssim_fwd', 'ssim_bwd') if n in k})
"
  done
done
