#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_shim_gpu.py -m gpu -q 2>&1 | tail -n 5
for mode in rank1 dense; do
GSLIC_EXCHANGE=$mode GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras --profile-all 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$mode', d['value'], 'views/s', d['ms_per_step'], 'ms', d['exchange']['compute_ms'], d['exchange']['exchange_window_ms'], d['kernel_ms_per_launch_timed'])"
done
} > gpurun_out/r03_call7.log 2>&1
tail -n 40 gpurun_out/r03_call7.log
