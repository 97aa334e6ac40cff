#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_rank1_exchange_gpu.py tests/test_fused_gpu.py -m gpu -q -x 2>&1 | tail -n 12
for ch in 1 4; do
GSLIC_EXCHANGE_CHUNKS=$ch GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('chunks $ch', d['value'], 'views/s', d['ms_per_step'], 'ms', d['exchange'])"
done
} > gpurun_out/r03_call12.log 2>&1
cat gpurun_out/r03_call12.log
