cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 420 python -m pytest tests -x -q -m gpu > gpurun_out/r04zz_gpu_suite.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04zz_gpu_suite.log
tail -3 gpurun_out/r04zz_gpu_suite.log
timeout 100 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 > gpurun_out/r04zz_bench_line.json
head -c 300 gpurun_out/r04zz_bench_line.json
