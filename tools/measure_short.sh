#!/bin/bash
# Reduced version of measure_all.sh for a tight GPU budget (~7 min): the lines and counters DESIGN.md section 6 / the bench roofline replay need.
#   gpurun --timeout 900 -- 'bash tools/measure_short.sh r04x'
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="timeout 300 python $R/bench.py"
X="--no-cpu-baseline --no-extras --steps 100"
{
  $B --extras 2>/dev/null | tail -1                                          # the default line (config 3, strict arithmetic, Morton row order) + every secondary leg (second process)
  $B $X --map-order insertion 2>/dev/null | tail -1                          # the same step on the rows as generated (random order)
  GSLIC_FAST_MATH=1 $B $X 2>/dev/null | tail -1                              # the opt-in fast arithmetic (pipeline backward)
  GSLIC_BWD_SCAN=0 $B $X 2>/dev/null | tail -1                               # strict arithmetic on round 3's pipeline backward
  $B --scene lidar --gaussians 500000 $X 2>/dev/null | tail -1               # config 2 at its own size
  $B --gaussians 5000000 --width 3840 --height 2160 $X --steps 30 2>/dev/null | tail -1   # config 5 shape on one GPU
  $B --density 1.6 --opacity-shift -4 $X --steps 50 --profile-all 2>/dev/null | tail -1   # long faint lists
  GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 $B $X 2>/dev/null | grep "^{" | tail -1
  GSLIC_EXCHANGE=dense GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29557 RANK=0 WORLD_SIZE=1 $B $X 2>/dev/null | grep "^{" | tail -1
} > $OUT/${TAG}_bench_lines.jsonl
# per-kernel durations of the driver's own command
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline --no-extras > /tmp/prof_$TAG.log 2>&1
grep "^{" /tmp/prof_$TAG.log | tail -1 > $OUT/${TAG}_bench_line_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG -name "*.db" | head -1) $OUT/${TAG}_train_2M_1080p_kernel_stats > /dev/null
python $R/tools/rocpd_timeline.py $(find /tmp/prof_$TAG -name "*.db" | head -1) preprocess_kernel 12 > $OUT/${TAG}_fused_timeline.txt 2>&1
# HBM traffic (separate passes), SQ counters, L2 / LDS
PMC_UNITS=/tmp/pmc_units_$TAG.json timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$TAG -o f -- python $R/tools/pmc_run.py > /tmp/pf.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw_$TAG -o w -- python $R/tools/pmc_run.py > /tmp/pw.log 2>&1
python $R/tools/pmc_extract.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) $OUT/${TAG}_pmc_traffic.json $TAG /tmp/pmc_units_$TAG.json > /dev/null
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d /tmp/sq_$TAG -o sq -- python $R/tools/pmc_run.py > /tmp/sq.log 2>&1
python $R/tools/pmc_sq_extract.py $(find /tmp/sq_$TAG -name "*.db" | head -1) > $OUT/${TAG}_sq_counters.txt 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/l2_$TAG -o l2 -- python $R/tools/pmc_run.py > /tmp/l2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d /tmp/lds_$TAG -o lds -- python $R/tools/pmc_run.py > /tmp/lds.log 2>&1
python $R/tools/pmc_cache_lds.py $(find /tmp/l2_$TAG -name "*.db" | head -1) $(find /tmp/lds_$TAG -name "*.db" | head -1) > $OUT/${TAG}_cache_lds.md 2>&1
ls -la $OUT | grep $TAG
