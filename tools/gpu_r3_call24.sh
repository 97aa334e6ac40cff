#!/bin/bash
# which kernels lose how much per lost wave of occupancy (dynamic-LDS padding through GSLIC_LDS_PAD): what an LDS diet could buy where
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras --profile-all 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_launch_timed']; print('$label', d['value'], {n: k[n] for n in ('preprocess', 'keybuild', 'sort_scatter', 'dsort_scatter', 'ssim_fwd', 'ssim_bwd', 'render_fwd', 'render_bwd', 'preprocess_bwd') if n in k})
"
}
{
run base X=1
run preprocess-13to10 GSLIC_LDS_PAD=preprocess=3000
run keybuild-5to4blocks GSLIC_LDS_PAD=keybuild=7000
run sort_scatter-2to1blocks GSLIC_LDS_PAD=sort_scatter=27000
run dsort_scatter-4to2blocks GSLIC_LDS_PAD=dsort_scatter=15000
run loss_fwd-3to2blocks GSLIC_LDS_PAD=ssim_fwd=13000
run loss_bwd-4to3blocks GSLIC_LDS_PAD=ssim_bwd=4000
run render_fwd-pad8k GSLIC_LDS_PAD=render_fwd=8000
run base2 X=1
} > gpurun_out/r03_call24.log 2>&1
cat gpurun_out/r03_call24.log
