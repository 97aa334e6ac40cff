#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for i in 1 2; do
timeout 600 python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default line: value', d['value'], 'long', d['value_long']['value'], 'graphed', d['graphed']['value'], 'cpp', d['cpp_fused_host']['value'], 'dropin', d['other_host_path']['value'], 'modes', d['math_modes']['strict']['value'], d['math_modes']['fast']['value'])"
timeout 600 python bench.py --steps 200 --no-cpu-baseline --no-extras --graph 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('--graph: value', d['value'])"
done
} > gpurun_out/r03_call15.log 2>&1
cat gpurun_out/r03_call15.log
