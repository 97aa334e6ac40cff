#!/bin/bash
# logf threshold + in-kernel scan-state clear + XCD-aware bucket order of render_bwd: parity, same-box A/B, fetch counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_vs_reference_kernels_gpu.py tests/test_fullsize_reference_gpu.py tests/test_capacity_graph_gpu.py -m gpu -q 2>&1 | tail -n 6
bash tools/ab/run_multi.sh 2 "prev|tools/ab/libgslic_hip_prev.so|" "new-run0|-|GSLIC_BWD_XCD_RUN=0" "new-run4|-|GSLIC_BWD_XCD_RUN=4" "new-run16|-|GSLIC_BWD_XCD_RUN=16" "new-run64|-|GSLIC_BWD_XCD_RUN=64"
cd /tmp
for run in 0 16; do
  GSLIC_BWD_XCD_RUN=$run timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$run -o f -- python $R/tools/pmc_run.py > /tmp/pf_$run.log 2>&1
  GSLIC_BWD_XCD_RUN=$run timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw_$run -o w -- python $R/tools/pmc_run.py > /tmp/pw_$run.log 2>&1
  python $R/tools/pmc_extract.py $(find /tmp/pf_$run -name "*.db" | head -1) $(find /tmp/pw_$run -name "*.db" | head -1) $R/gpurun_out/r03l_pmc_run$run.json run$run | grep -i "render_bwd\|render_fwd\|preprocess_bwd" | head -6
done
} > gpurun_out/r03_call17.log 2>&1
cat gpurun_out/r03_call17.log
