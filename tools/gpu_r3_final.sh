#!/bin/bash
# what the driver runs at round end, on the final tree: smoke(), then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r03u_final_bench_line.json
python -c "
import json; d = json.load(open('gpurun_out/r03u_final_bench_line.json'))
print({k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'dtype', 'scaling', 'vs_baseline')})
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'bound', 'achieved', 'peak', 'frac', 'traffic', 'avg_launch_ms')})
print('cpu_baseline', {k: d['cpu_baseline'][k] for k in ('value', 'unit', 'cores', 'kind')})
print('math_modes', d['math_modes']['strict']['value'], d['math_modes']['fast']['value'], 'growth', d['growth_schedule']['value'], d['growth_schedule'].get('kernel_ms_per_launch'))
"
} > gpurun_out/r03_final.log 2>&1
cat gpurun_out/r03_final.log
