#!/bin/bash
# SQ counters of the default bench workload, two passes:  bash tools/gpu_sq.sh [name-filter]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT
F=${1:-render}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqa /tmp/sqb
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace -d /tmp/sqa -o sq -- python $R/tools/pmc_run.py > /tmp/sqa.log 2>&1
python $R/tools/pmc_sq_extract.py $(find /tmp/sqa -name "*.db" | head -1) $F | tee $OUT/sq_a.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_IFETCH --kernel-trace -d /tmp/sqb -o sq -- python $R/tools/pmc_run.py > /tmp/sqb.log 2>&1
python $R/tools/pmc_sq_extract.py $(find /tmp/sqb -name "*.db" | head -1) $F | tee $OUT/sq_b.txt
