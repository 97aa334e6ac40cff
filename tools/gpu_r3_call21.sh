#!/bin/bash
# preprocess_bwd under dynamic-LDS padding: how much does its occupancy (13 waves per CU at 11.6 KB) matter?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/ab/run_multi.sh 2 "pad0-13waves|-|" "pad1800-12waves|-|GSLIC_LDS_PAD=preprocess_bwd=1800" "pad3000-10waves|-|GSLIC_LDS_PAD=preprocess_bwd=3000" "pad8000-8waves|-|GSLIC_LDS_PAD=preprocess_bwd=8000" "pad15000-6waves|-|GSLIC_LDS_PAD=preprocess_bwd=15000" > gpurun_out/r03_call21.log 2>&1
cat gpurun_out/r03_call21.log
