#!/bin/bash
# round 6: 300 + 300 fuzz scenes with the binning path and the row order forced per case; the bench line after the bench.py / bench_legs.py split; skip reasons
set -u
TAG=${1:-r06x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python tests/fuzz_vs_reference.py 300 6001 > $OUT/${TAG}_fuzz_300_forced_paths.txt 2>&1; echo "rc $?" >> $OUT/${TAG}_fuzz_300_forced_paths.txt
tail -2 $OUT/${TAG}_fuzz_300_forced_paths.txt
timeout 900 python tests/fuzz_vs_reference.py --poses 300 6002 > $OUT/${TAG}_fuzz_poses_300_forced_paths.txt 2>&1; echo "rc $?" >> $OUT/${TAG}_fuzz_poses_300_forced_paths.txt
tail -2 $OUT/${TAG}_fuzz_poses_300_forced_paths.txt
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/${TAG}_bench_stderr.log | tail -1 > $OUT/${TAG}_bench_line_driver_command.json
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench_line_driver_command.json"))
print({k: d.get(k) for k in ("value", "ms_per_step")}, d["config"].get("growth_schedule"), d["config"]["dropin_host"]["value"], d["dropin_host"]["map_order_sort_ms"], d["dropin_host"]["process_seconds"])
PY
timeout 600 python -m pytest tests/test_shim_gpu.py tests/test_dist_gpu.py tests/test_pose_reference_gpu.py -q -m gpu -rs -k "two_ranks or golden or rccl or reference_host" 2>&1 | tail -8
