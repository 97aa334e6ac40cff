#!/bin/bash
# full GPU suite on the current tree + smoke + where the C++ fused host spends its step (kernel-trace timeline of fused_check)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
GSLIC_CPP_HOST_DIR=/tmp/cpph timeout 600 python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('python host', d['value'], 'graphed', (d.get('graphed') or {}).get('value'), 'cpp', d.get('cpp_fused_host'))
"
ls /tmp/cpph | head -3
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cpp -o cpp -- $R/gaussian-lic_amd/fused_check /tmp/cpph 2000000 1920 1080 3 1 40 0.01 2>&1 | grep -i "views_per_s"
python $R/tools/rocpd_timeline.py $(find /tmp/prof_cpp -name "*.db" | head -1) preprocess_kernel 30 > $R/gpurun_out/r03_cpp_fused_timeline.txt 2>&1
tail -45 $R/gpurun_out/r03_cpp_fused_timeline.txt
} > gpurun_out/r03_call20.log 2>&1
cat gpurun_out/r03_call20.log
