#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for mode in rank1 dense; do
GSLIC_EXCHANGE=$mode GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$mode -o t -- python $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras > /tmp/tl_$mode.log 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/tl_$mode -name "*.db" | head -1) preprocess_kernel 12 > $R/gpurun_out/r03_timeline_$mode.txt 2>&1
done
cat $R/gpurun_out/r03_timeline_rank1.txt; cat $R/gpurun_out/r03_timeline_dense.txt | tail -22
