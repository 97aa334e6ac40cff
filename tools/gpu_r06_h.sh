#!/bin/bash
set -u
TAG=${1:-r06x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
bash tools/final_check.sh $TAG
bash tools/ab/run_multi.sh 2 "sht19-base|-|" "sht18|tools/ab/libgslic_hip_sht18.so|" "sht20|tools/ab/libgslic_hip_sht20.so|" "sht21|tools/ab/libgslic_hip_sht21.so|" "sht23|tools/ab/libgslic_hip_sht23.so|" > $OUT/${TAG}_pbwd_table_stride_ab.log 2>&1
cat $OUT/${TAG}_pbwd_table_stride_ab.log
