"""Per-kernel averages of SQ counters from a rocprofv3 --pmc rocpd database. usage: pmc_sq_extract.py <db> [name-filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else "gslic::"
rows = c.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events group by name, counter_name").fetchall()
out = {}
for name, cn, n, avg, dur in rows:
    if flt not in name: continue
    short = name.split("gslic::")[1].split("(")[0] if "gslic::" in name else name[:40]
    out.setdefault(short, {"n": n, "dur_us": dur / 1e3})[cn] = avg
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["dur_us"] * kv[1]["n"]):
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
