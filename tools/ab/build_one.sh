#!/bin/bash
# A variant library that differs from the in-tree one in ONE source file's flags (the other objects are the in-tree build's):
#   tools/ab/build_one.sh render.hip "-DGS_FWD_SKIP=2" tools/ab/libgslic_hip_fskip2.so
set -e
cd "$(dirname "$0")/../.."
SRC=$1; EXTRA=$2; OUT=$3
C=gaussian-lic_amd/csrc
O=$C/build
TMP=$(mktemp -d)
PER=""
case $SRC in
  render.hip|render_bwd_scan.hip) PER="-fno-slp-vectorize";;
  preprocess.hip|ssim.hip|extend.hip) PER="-ffp-contract=off";;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -Wno-inline-asm -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function $PER $EXTRA -c $C/$SRC -o $TMP/v.o
OBJS=""
for f in api scan radix_sort tile_bin preprocess render render_bwd_scan preprocess_bwd adam ssim knn extend; do
  if [ "$f.hip" = "$SRC" ]; then OBJS="$OBJS $TMP/v.o"; else OBJS="$OBJS $O/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
rm -rf $TMP
echo $OUT
