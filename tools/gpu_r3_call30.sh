#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
python tools/ab/dump_loss.py /tmp/new.npz 2>&1 | tail -1
GSLIC_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libgslic_hip_prev.so python tools/ab/dump_loss.py /tmp/prev.npz 2>&1 | tail -1
python tools/ab/dump_loss.py /tmp/new2.npz 2>&1 | tail -1
python - <<'PY'
import numpy as np
a, b, c = np.load('/tmp/new.npz'), np.load('/tmp/prev.npz'), np.load('/tmp/new2.npz')
for k in a.files:
    d = np.abs(a[k] - b[k]); d2 = np.abs(a[k] - c[k])
    print(k, 'new vs prev: differing elements', int((a[k] != b[k]).sum()), 'of', a[k].size, 'max abs', float(d.max()), 'rel to max', float(d.max() / np.abs(b[k]).max()), '| new vs new (run to run):', int((a[k] != c[k]).sum()))
PY
} > gpurun_out/r03_call30.log 2>&1
cat gpurun_out/r03_call30.log
