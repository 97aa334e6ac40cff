#!/bin/bash
# preprocess_bwd with the cooperative SH column pass (parameters read once, 16 waves per CU): full GPU suite + same-box A/B + counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 8
bash tools/ab/run_multi.sh 3 "prev|tools/ab/libgslic_hip_prev.so|" "columns|-|"
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o f -- python $R/tools/pmc_run.py > /tmp/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o w -- python $R/tools/pmc_run.py > /tmp/pw.log 2>&1
python $R/tools/pmc_extract.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) $R/gpurun_out/r03r_pmc_traffic.json r03r | cut -c1-900
} > gpurun_out/r03_call22.log 2>&1
cat gpurun_out/r03_call22.log
