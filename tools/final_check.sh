#!/bin/bash
# What the driver runs at round end, on the final tree, plus the driver's bench command once more under rocprofv3 (agreement check):
#   gpurun --timeout 1500 -- 'bash tools/final_check.sh r06z'
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_gpu_suite.log 2>&1; echo "pytest rc $?" >> $OUT/${TAG}_gpu_suite.log
tail -3 $OUT/${TAG}_gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc $?" >> $OUT/${TAG}_smoke.log
tail -2 $OUT/${TAG}_smoke.log
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_driver_command.json
timeout 200 python3 bench.py --gaussians 5000000 --width 3840 --height 2160 --no-cpu-baseline --no-extras --steps 30 --profile-all 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_config5_shape_all_kernels.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/prof_$TAG.log 2>&1
grep "^{" /tmp/prof_$TAG.log | tail -1 > $OUT/${TAG}_bench_line_driver_command_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG -name "*.db" | head -1) $OUT/${TAG}_driver_command_kernel_stats > /dev/null
python $R/tools/rocpd_timeline.py $(find /tmp/prof_$TAG -name "*.db" | head -1) preprocess_kernel 12 > $OUT/${TAG}_fused_timeline.txt 2>&1
# HBM counter traffic of the final tree (round 6: part of the final check, so that bench.py's replayed `roofline.traffic` comes from THIS tree): two
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes (tools/pmc_extract.py);
# copy ${TAG}_pmc_traffic.json to profiles/pmc_traffic.json
PMC_UNITS=/tmp/pmc_units_$TAG.json timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$TAG -o f -- python $R/tools/pmc_run.py > /tmp/pf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw_$TAG -o w -- python $R/tools/pmc_run.py > /tmp/pw.log 2>&1
python $R/tools/pmc_extract.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) $OUT/${TAG}_pmc_traffic.json $TAG /tmp/pmc_units_$TAG.json > /dev/null
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/l2_$TAG -o l2 -- python $R/tools/pmc_run.py > /tmp/l2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d /tmp/lds_$TAG -o lds -- python $R/tools/pmc_run.py > /tmp/lds.log 2>&1
python $R/tools/pmc_cache_lds.py $(find /tmp/l2_$TAG -name "*.db" | head -1) $(find /tmp/lds_$TAG -name "*.db" | head -1) > $OUT/${TAG}_cache_lds.md 2>&1
ls -la $OUT | grep $TAG
