#!/bin/bash
# round 3, second GPU call: render_bwd LDS diet + dead flags + 36-byte rows; forward: XCD-aware grid, longest-first tile order
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== A/B (2 rounds)"
bash tools/ab/run_multi.sh 2 "new-strict|tools/ab/libgslic_hip_new.so|" "new2-strict|tools/ab/libgslic_hip_new2.so|" "new2-strict-noorder|tools/ab/libgslic_hip_new2.so|GSLIC_NO_TILE_ORDER=1" \
   "new-fast|tools/ab/libgslic_hip_new.so|GSLIC_FAST_MATH=1" "new2-fast|tools/ab/libgslic_hip_new2.so|GSLIC_FAST_MATH=1" "new2-fast-noorder|tools/ab/libgslic_hip_new2.so|GSLIC_FAST_MATH=1 GSLIC_NO_TILE_ORDER=1"
echo "== 5M / 4K"
BENCH_ARGS="--gaussians 5000000 --width 3840 --height 2160 --steps 30" bash tools/ab/run_multi.sh 1 "new-strict-4k|tools/ab/libgslic_hip_new.so|" "new2-strict-4k|tools/ab/libgslic_hip_new2.so|" "new2-fast-4k|tools/ab/libgslic_hip_new2.so|GSLIC_FAST_MATH=1"
echo "== full gpu suite"
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 15
} > gpurun_out/r03_call2.log 2>&1
tail -n 60 gpurun_out/r03_call2.log
