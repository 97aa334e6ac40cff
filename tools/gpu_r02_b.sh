#!/bin/bash
# round 2, GPU call B: chained backward — correctness (parity suites), chain-length sweep, flips vs the reference
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fused_gpu.py tests/test_ops_gpu.py -m gpu -x -q ) > gpurun_out/tests_b.log 2>&1
echo "tests rc=$?"; tail -n 15 gpurun_out/tests_b.log
for k in 1 2 4 8 16 32; do
  echo "== chain $k"; GSLIC_BWD_CHAIN=$k timeout 300 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'views/s', d['ms_per_step'], 'ms', {k: v for k, v in d['kernel_ms_per_step'].items() if k in ('render_bwd', 'render_fwd', 'preprocess_bwd')}, 'dom', d['roofline']['kernel'], d['roofline']['avg_launch_ms'])
"
done
( timeout 600 python tests/parity_report.py --out gpurun_out/parity_report_b.json --configs small,c2,c3 ) > gpurun_out/parity_report_b.log 2>&1
echo "parity_report rc=$?"; grep -v amdgpu.ids gpurun_out/parity_report_b.log | cut -c1-900
