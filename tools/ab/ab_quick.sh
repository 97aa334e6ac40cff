# parity suites on the in-tree library, then same-box A/B against tools/ab/libgslic_hip_base.so (a copy of the library built BEFORE the change under test):
#   cp gaussian-lic_amd/libgslic_hip.so tools/ab/libgslic_hip_base.so; <edit>; python gaussian-lic_amd/build.py; gpurun -- "bash tools/ab/ab_quick.sh 2"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_reference_gpu.py tests/test_fuzz_vs_reference_gpu.py tests/test_vs_reference_kernels_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab/run_multi.sh ${1:-2} "base|tools/ab/libgslic_hip_base.so|" "new|-|"
BENCH_ARGS="--density 1.6 --opacity-shift -4 --steps 50" bash tools/ab/run_multi.sh 1 "base-faint|tools/ab/libgslic_hip_base.so|" "new-faint|-|"
