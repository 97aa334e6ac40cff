#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
{
cd $R && bash tools/ab/run_multi.sh 3 "base|-|" "fulllines|tools/ab/libgslic_hip_fulllines.so|"
cd /tmp
for v in base fulllines; do
  if [ $v = base ]; then L=""; else L="GSLIC_HIP_LIB=$R/tools/ab/libgslic_hip_fulllines.so"; fi
  env $L timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$v -o f -- python $R/tools/pmc_run.py > /tmp/pf.log 2>&1
  env $L timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw_$v -o w -- python $R/tools/pmc_run.py > /tmp/pw.log 2>&1
  python $R/tools/pmc_extract.py $(find /tmp/pf_$v -name "*.db" | head -1) $(find /tmp/pw_$v -name "*.db" | head -1) /tmp/pmc_$v.json $v > /dev/null
  python - <<PY
import json
d = json.load(open("/tmp/pmc_$v.json"))
k = d["kernels"]["preprocess_bwd_kernel"]
print("$v preprocess_bwd traffic GB: total %.3f fetch %.3f write %.3f" % (k["hbm_bytes_per_launch"] / 1e9, k["fetch_bytes"] / 1e9, k["write_bytes"] / 1e9))
PY
done
} > $R/gpurun_out/r03_call14.log 2>&1
cat $R/gpurun_out/r03_call14.log
