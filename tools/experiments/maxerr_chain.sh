cd $GRAFT_REPO_ROOT
for lib in "" tools/ab/libgslic_hip_sc.so; do
  echo "== ${lib:-in-tree}"
  if [ -n "$lib" ]; then export GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset GSLIC_HIP_LIB; fi
  timeout 1200 python -m pytest tests/test_pose_reference_gpu.py tests/test_fuzz_vs_reference_gpu.py tests/test_fullsize_reference_gpu.py tests/test_timed_path_reference_gpu.py -q -m gpu -s 2>&1 | tee /tmp/out_$$.log | grep -E "\[strict\]" | grep -oE "dL_(dmean3D|dcov3D|dscale|drot) over=[0-9]+/[0-9]+ max=[0-9.e-]+" | awk '{split($3,a,"="); k=$1; if (a[2]+0 > m[k]) m[k]=a[2]+0; s[k]+=a[2]; n[k]++} END{for (k in m) printf "%s worst %.2e mean %.2e over %d cases\n", k, m[k], s[k]/n[k], n[k]}'
  tail -1 /tmp/out_$$.log
done
