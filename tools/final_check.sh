#!/bin/bash
# What the driver runs at round end, on the final tree, plus the driver's bench command once more under rocprofv3 (agreement check):
#   gpurun --timeout 900 -- 'bash tools/final_check.sh r04z'
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_gpu_suite.log 2>&1; echo "pytest rc $?" >> $OUT/${TAG}_gpu_suite.log
tail -3 $OUT/${TAG}_gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc $?" >> $OUT/${TAG}_smoke.log
tail -2 $OUT/${TAG}_smoke.log
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_driver_command.json
timeout 200 python3 bench.py --gaussians 5000000 --width 3840 --height 2160 --no-cpu-baseline --no-extras --steps 30 --profile-all 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_config5_shape_all_kernels.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/prof_$TAG.log 2>&1
grep "^{" /tmp/prof_$TAG.log | tail -1 > $OUT/${TAG}_bench_line_driver_command_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG -name "*.db" | head -1) $OUT/${TAG}_driver_command_kernel_stats > /dev/null
ls -la $OUT | grep $TAG
