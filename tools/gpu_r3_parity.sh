#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python tests/parity_report.py --out gpurun_out/r03_parity_fullsize.json > gpurun_out/r03_parity_fullsize.log 2>&1
timeout 1500 python tests/fuzz_vs_reference.py 400 1000 > gpurun_out/r03_fuzz_full.txt 2>&1
grep -v "strict OK" gpurun_out/r03_fuzz_full.txt | tail -5 > gpurun_out/r03_fuzz_vs_reference.txt
tail -3 gpurun_out/r03_fuzz_vs_reference.txt; grep -c "" gpurun_out/r03_parity_fullsize.log
