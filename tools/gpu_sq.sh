#!/bin/bash
# SQ counters of the default bench workload for the given chain lengths:  bash tools/gpu_sq.sh "1 8"
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in ${1:-8}; do
  rm -rf /tmp/sq_$k
  GSLIC_BWD_CHAIN=$k timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d /tmp/sq_$k -o sq -- python $R/tools/pmc_run.py > /tmp/sq_$k.log 2>&1
  echo "== chain $k"; python $R/tools/pmc_sq_extract.py $(find /tmp/sq_$k -name "*.db" | head -1) render | tee $OUT/sq_chain_$k.txt
done
