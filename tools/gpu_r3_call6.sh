#!/bin/bash
# round 3, sixth GPU call: ABI 5 (status words, view_stride), GraphedStep window, packed rank-1 exchange, current-stream shim
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== full gpu suite"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -n 25
echo "== one-rank RCCL group: rank-1 / dense"
for mode in rank1 dense; do
GSLIC_EXCHANGE=$mode GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$mode', d['value'], 'views/s', d['ms_per_step'], 'ms', d['exchange'])"
done
} > gpurun_out/r03_call6.log 2>&1
tail -n 40 gpurun_out/r03_call6.log
