#!/bin/bash
# round 6: parity of the split binned rows, same-box A/B (base / new / forward elimination bits), the default bench line with its drop-in leg, and
# the reference's unmodified C++ host under rocprofv3
#   gpurun --timeout 1500 -- 'bash tools/gpu_r06_dropin.sh r06d'
set -u
TAG=${1:-r06x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_binning_gpu.py tests/test_morton_order_gpu.py tests/test_timed_path_reference_gpu.py tests/test_parity_gpu.py tests/test_capacity_graph_gpu.py -x -q -m gpu > $OUT/${TAG}_split_rows_tests.log 2>&1
echo "pytest rc $?" >> $OUT/${TAG}_split_rows_tests.log
tail -4 $OUT/${TAG}_split_rows_tests.log
bash tools/ab/run_multi.sh 2 "base-16B-rows|tools/ab/libgslic_hip_base.so|" "split-rows|-|" "fwd-skip2-nockpt|tools/ab/libgslic_hip_fskip2.so|" "fwd-skip4-nomasks|tools/ab/libgslic_hip_fskip4.so|" "fwd-skip6-neither|tools/ab/libgslic_hip_fskip6.so|" > $OUT/${TAG}_ab.log 2>&1
cat $OUT/${TAG}_ab.log
BENCH_PROFILE="" bash tools/ab/run_multi.sh 2 "base-16B-rows|tools/ab/libgslic_hip_base.so|" "split-rows|-|" > $OUT/${TAG}_ab_unprofiled.log 2>&1
cat $OUT/${TAG}_ab_unprofiled.log
export GSLIC_CPP_HOST_DIR=/tmp/cpp_host
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/${TAG}_bench_stderr.log | tail -1 > $OUT/${TAG}_bench_line_driver_command.json
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench_line_driver_command.json"))
print({k: d.get(k) for k in ("value", "ms_per_step")}, json.dumps(d["config"].get("dropin_host"), indent=1))
print(json.dumps(d.get("dropin_host"), indent=1)[:6000])
PY
cd /tmp && export TMPDIR=/tmp
for prog in dropin_check_render_refhost dropin_check_render; do
  GSLIC_CHECK_TIME=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$prog -o $prog -- $R/gaussian-lic_amd/$prog /tmp/cpp_host/rows_insertion 2000000 1920 1080 3 43 > /tmp/prof_$prog.log 2>&1
  tail -2 /tmp/prof_$prog.log
  python $R/tools/rocpd_summary.py $(find /tmp/prof_${TAG}_$prog -name "*.db" | head -1) $OUT/${TAG}_${prog}_kernel_stats > /dev/null
  python $R/tools/rocpd_timeline.py $(find /tmp/prof_${TAG}_$prog -name "*.db" | head -1) preprocess_kernel 40 > $OUT/${TAG}_${prog}_timeline.txt 2>&1
done
ls -la $OUT | grep $TAG
