#!/bin/bash
# Round 6, second session: where does the capacity-mode step (no host round trip) spend what the two mailbox round trips of the callback
# forward cost?  Un-instrumented A/B, then a kernel trace + one-step timeline of each.
#   gpurun --timeout 900 -- 'bash tools/gpu_r06_cap.sh r06m'
set -u
TAG=${1:-r06m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
PROBE_WHICH=ce PROBE_HEADROOM=1.05 timeout 300 python tools/experiments/capacity_eager_probe.py 2000000 1920 1080 300 > $OUT/${TAG}_cap_probe.log 2>&1
tail -2 $OUT/${TAG}_cap_probe.log
cd /tmp && export TMPDIR=/tmp
for w in e c; do
  PROBE_WHICH=$w PROBE_HEADROOM=1.05 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$w -o $w -- python $R/tools/experiments/capacity_eager_probe.py 2000000 1920 1080 100 > /tmp/prof_${TAG}_$w.log 2>&1
  DB=$(find /tmp/prof_${TAG}_$w -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB $OUT/${TAG}_cap_${w}_kernel_stats > /dev/null
  python $R/tools/rocpd_timeline.py $DB preprocess_kernel -3 > $OUT/${TAG}_cap_${w}_timeline.txt 2>&1
  python $R/tools/rocpd_timeline.py $DB preprocess_kernel -5 >> $OUT/${TAG}_cap_${w}_timeline.txt 2>&1
done
ls -la $OUT | grep $TAG
