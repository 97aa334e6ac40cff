cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python tests/fuzz_vs_reference.py 150 5000 130 2>&1 | tail -3
GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libgslic_hip_r2.so python tests/fuzz_vs_reference.py 150 5000 130 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_dist_gpu.py -m gpu -q -x 2>&1 | tail -5
