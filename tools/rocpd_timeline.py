"""One step of a rocprofv3 rocpd kernel trace as a timeline: every dispatch between two consecutive launches of an anchor kernel, with its
start relative to the anchor, its duration and the idle gap in front of it — where a step's GPU time goes when the kernels' durations do
not add up to it (host-bound launches, collectives, copies).
usage: python tools/rocpd_timeline.py <results.db> [anchor substring = preprocess_kernel] [which step = -3]"""
import sqlite3
import sys

db = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "preprocess_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
a, b = idx[which], idx[which + 1]
t0, prev_end, busy = rows[a][1], rows[a][1], 0
print(f"step of {(rows[b][1] - t0) / 1e3:.1f} us, {b - a} dispatches")
for name, s, e in rows[a:b]:
    short = name.split("(")[0][-70:]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"busy {busy / 1e3:.1f} us of {(rows[b][1] - t0) / 1e3:.1f}")
