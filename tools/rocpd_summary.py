"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a per-kernel CSV + markdown table.
usage: python tools/rocpd_summary.py <results.db> <out_prefix>"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {namecol} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(out + ".csv", "w") as f:
    f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
    for r in rows:
        f.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.0f},{r[4]},{r[5]},{100.0*r[2]/tot:.2f}\n")
with open(out + ".md", "w") as f:
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:40]:
        n = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        f.write(f"| `{n}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {100.0*r[2]/tot:.1f} |\n")
print(open(out + ".md").read())
