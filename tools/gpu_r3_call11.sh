#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== render_bwd occupancy sensitivity (dynamic LDS padding)"
bash tools/ab/run_multi.sh 2 "strict-pad0|-|" "strict-pad2k|-|GSLIC_LDS_PAD=render_bwd=2048" "strict-pad5k|-|GSLIC_LDS_PAD=render_bwd=5120" "fast-pad0|-|GSLIC_FAST_MATH=1" "fast-pad2k|-|GSLIC_FAST_MATH=1 GSLIC_LDS_PAD=render_bwd=2048"
} > gpurun_out/r03_call11.log 2>&1
cat gpurun_out/r03_call11.log
