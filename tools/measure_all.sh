#!/bin/bash
# Everything profiles/ and DESIGN.md section 6 quote, in one call on the MI355X box:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/measure_all.sh r03a'
# Writes small summaries to gpurun_out/<tag>_*; rocprofv3 databases stay in /tmp (too large to merge back).
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="timeout 600 python $R/bench.py"
X="--no-cpu-baseline --no-extras --steps 100"
# a saveMap-format PLY of the LiDAR-seeded scene (io_ply.save_map: byte-identical to the reference's tinyply output), for the --ply leg
python - <<PY
import sys, types; sys.path.insert(0, "$R")
import gaussian_lic_amd
from gaussian_lic_amd import io_ply
from gaussian_lic_amd.synthetic import lidar_scene
raw = lidar_scene(500000, 1920, 1080, sh_degree=3, seed=0)
m = types.SimpleNamespace(xyz=raw["xyz"], features_dc=raw["features_dc"], features_rest=raw["features_rest"], opacity=raw["opacity"], scaling=raw["scaling"], rotation=raw["rotation"])
print(io_ply.save_map(m, "/tmp/lidar_map.ply"), "Gaussians written to /tmp/lidar_map.ply")
PY
{
  $B 2>/dev/null | tail -1                                                   # the driver's default line (config 3, strict arithmetic), every secondary leg
  GSLIC_FAST_MATH=1 $B $X 2>/dev/null | tail -1                              # the opt-in fast arithmetic of the blend kernels
  $B --mode render $X 2>/dev/null | tail -1                                  # bare fwd + bwd
  $B --split-adam $X 2>/dev/null | tail -1                                   # Adam as its own launch (compute path of a rank at N > 1)
  $B --graph $X 2>/dev/null | tail -1                                        # the step as one hipGraph replay
  $B --scene lidar --gaussians 500000 $X 2>/dev/null | tail -1               # config 2 at its own size
  $B --ply /tmp/lidar_map.ply $X 2>/dev/null | tail -1                       # the same map loaded from its saveMap PLY file
  $B --gaussians 5000000 --width 3840 --height 2160 $X --steps 30 2>/dev/null | tail -1   # config 5 shape on one GPU
  $B --density 2.5 --opacity-shift -2 $X --steps 50 --profile-all 2>/dev/null | tail -1   # dense: 11 instances per visible Gaussian (R = 15M: the tile sort's regime)
  $B --density 1.6 --opacity-shift -4 $X --steps 50 --profile-all 2>/dev/null | tail -1   # long faint lists: the blend kernels walk 3x further
  # the per-rank compute leg of the N > 1 step, in a ONE-rank RCCL group (collectives degenerate to copies): rank-1 exchange (default) and dense slab
  GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 $B $X 2>/dev/null | grep "^{" | tail -1
  GSLIC_EXCHANGE=dense GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29557 RANK=0 WORLD_SIZE=1 $B $X 2>/dev/null | grep "^{" | tail -1
} > $OUT/${TAG}_bench_lines.jsonl
# per-kernel durations
# (the default step counts, so that the trace's average of the dominant kernel and the HIP-event average bench.py prints for it come from the
# SAME process: the line is kept next to the stats)
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline --no-extras > /tmp/prof_$TAG.log 2>&1
grep "^{" /tmp/prof_$TAG.log | tail -1 > $OUT/${TAG}_bench_line_under_rocprof.json
python $R/tools/rocpd_summary.py $(find /tmp/prof_$TAG -name "*.db" | head -1) $OUT/${TAG}_train_2M_1080p_kernel_stats > /dev/null
# the same step through the reference's operator API + LibTorch autograd (what an unmodified reference host runs): where its extra time goes
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_dropin_$TAG -o $TAG -- python $R/bench.py --host dropin --steps 20 --warmup 4 --no-cpu-baseline --no-extras > /tmp/prof_dropin_$TAG.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_dropin_$TAG -name "*.db" | head -1) $OUT/${TAG}_dropin_kernel_stats > /dev/null
python $R/tools/rocpd_timeline.py $(find /tmp/prof_dropin_$TAG -name "*.db" | head -1) preprocess_kernel 12 > $OUT/${TAG}_dropin_timeline.txt 2>&1
python $R/tools/rocpd_timeline.py $(find /tmp/prof_$TAG -name "*.db" | head -1) preprocess_kernel 12 > $OUT/${TAG}_fused_timeline.txt 2>&1
# HBM traffic (separate passes), SQ counters
PMC_UNITS=/tmp/pmc_units_$TAG.json timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$TAG -o f -- python $R/tools/pmc_run.py > /tmp/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw_$TAG -o w -- python $R/tools/pmc_run.py > /tmp/pw.log 2>&1
python $R/tools/pmc_extract.py $(find /tmp/pf_$TAG -name "*.db" | head -1) $(find /tmp/pw_$TAG -name "*.db" | head -1) $OUT/${TAG}_pmc_traffic.json $TAG /tmp/pmc_units_$TAG.json > /dev/null
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d /tmp/sq_$TAG -o sq -- python $R/tools/pmc_run.py > /tmp/sq.log 2>&1
python $R/tools/pmc_sq_extract.py $(find /tmp/sq_$TAG -name "*.db" | head -1) > $OUT/${TAG}_sq_counters.txt 2>&1
# L2 hit rate and LDS bank conflicts (the "LDS-hit counters" of the north-star), one pass each
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/l2_$TAG -o l2 -- python $R/tools/pmc_run.py > /tmp/l2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d /tmp/lds_$TAG -o lds -- python $R/tools/pmc_run.py > /tmp/lds.log 2>&1
python $R/tools/pmc_cache_lds.py $(find /tmp/l2_$TAG -name "*.db" | head -1) $(find /tmp/lds_$TAG -name "*.db" | head -1) > $OUT/${TAG}_cache_lds.md 2>&1
timeout 60 $R/tools/ubench/issue_rate > $OUT/${TAG}_ubench.txt 2>&1
timeout 60 $R/tools/ubench/exec_rate >> $OUT/${TAG}_ubench.txt 2>&1
timeout 60 $R/tools/ubench/hbm_rate >> $OUT/${TAG}_ubench.txt 2>&1
ls -la $OUT | tail -12
