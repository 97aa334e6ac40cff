"""L2 hit rate and LDS bank-conflict share per kernel from two rocprofv3 --pmc databases.
usage: pmc_cache_lds.py <l2.db (TCC_HIT_sum TCC_MISS_sum)> <lds.db (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS)>"""
import sqlite3
import sys


def load(db):
    c = sqlite3.connect(db)
    out = {}
    for name, cn, n, tot, dur in c.execute("select name, counter_name, count(*), sum(counter_value), sum(duration) from pmc_events group by name, counter_name"):
        if "gslic::" not in name:
            continue
        short = name.split("gslic::")[1].split("(")[0]
        out.setdefault(short, {"dur_ns": dur})[cn] = tot
    return out


l2, lds = load(sys.argv[1]), load(sys.argv[2])
rows = sorted(set(l2) | set(lds), key=lambda k: -(l2.get(k, lds.get(k))["dur_ns"]))
print("| kernel | L2 hit rate (TCC_HIT / (HIT + MISS)) | LDS bank-conflict cycles / LDS-active cycles | LDS instructions per launch-set |")
print("|---|---|---|---|")
for k in rows:
    a, b = l2.get(k, {}), lds.get(k, {})
    hit, miss = a.get("TCC_HIT_sum", 0.0), a.get("TCC_MISS_sum", 0.0)
    conf, act = b.get("SQ_LDS_BANK_CONFLICT", 0.0), b.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"| `{k}` | {hit / (hit + miss):.3f} |" if hit + miss > 0 else f"| `{k}` | — |", end="")
    print(f" {conf / act:.3f} |" if act > 0 else " — |", end="")
    print(f" {b.get('SQ_INSTS_LDS', 0.0):.3g} |")
