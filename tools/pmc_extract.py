"""Reads the two rocprofv3 --pmc databases (FETCH_SIZE pass, WRITE_SIZE pass) and writes profiles/pmc_traffic.json:
per-kernel average counter values, the calibration factors measured on the known 1 GiB copy, and calibrated HBM bytes
per launch.  usage: python tools/pmc_extract.py <fetch.db> <write.db> <out.json> [tag] [units.json]"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
    rows = c.execute("select * from pmc_events limit 1").fetchall()
    q = None
    # rocpd view: pmc_events(... kernel name column, counter name column, value)
    namecol = next((x for x in ("name", "kernel_name") if x in cols), None)
    ccol = next((x for x in ("counter_name", "pmc_name", "symbol") if x in cols), None)
    vcol = next((x for x in ("counter_value", "value") if x in cols), None)
    assert namecol and ccol and vcol, cols
    out = {}
    for name, n, avg, mx in c.execute(f"select {namecol}, count(*), avg({vcol}), max({vcol}) from pmc_events where {ccol}=? group by {namecol}", (counter,)):
        out[name] = (n, avg, mx)
    return out


def main(fetch_db, write_db, out, tag="untagged", units_file=None):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    GiB = float(1 << 30)
    cal = "__amd_rocclr_copyBuffer"   # tensor.copy_ of 1 GiB = hipMemcpyAsync D2D; its largest dispatches are the calibration copies
    # counters are in KB (rocprofv3 derived metrics); factors convert the reported value to true bytes on this access pattern
    fetch_factor = GiB / (f[cal][2] * 1024.0)
    write_factor = GiB / (w[cal][2] * 1024.0)
    res = {"workload": "random-2000000-1920x1080", "tag": tag, "calibration_kernel": cal, "fetch_factor": fetch_factor, "write_factor": write_factor,
           "note": "bytes = counter_KB * 1024 * factor; factors measured on a 1 GiB torch copy in the same runs (guide: FETCH_SIZE reads 1/2 on wide loads on gfx950)",
           "kernels": {}}
    if units_file:   # unit counts of the workload the counters were collected on (tools/pmc_run.py, PMC_UNITS): bench.py compares them with its own
        u = json.load(open(units_file))
        res["units"] = {k: u[k] for k in ("P", "V", "R", "B", "R_live", "B_live", "steps_before")}
        res["strict"] = bool(u.get("strict", True))
        res["map_order"] = u.get("map_order", "insertion")   # row order of the map the counters were collected on (bench.py --map-order)
    for k in sorted(set(f) | set(w)):
        if "gslic::" not in k:
            continue
        short = k.split("gslic::")[1].split("(")[0]
        fb = f.get(k, (0, 0.0))[1] * 1024.0 * fetch_factor
        wb = w.get(k, (0, 0.0))[1] * 1024.0 * write_factor
        res["kernels"][short] = {"launches": f.get(k, (0, 0))[0], "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb,
                                 "raw_FETCH_SIZE_KB": f.get(k, (0, 0.0))[1], "raw_WRITE_SIZE_KB": w.get(k, (0, 0.0))[1]}
    # template kernels also under their bare name when that is unambiguous (bench.py looks kernels up as "<name>_kernel")
    bare = {}
    for k in res["kernels"]:
        bare.setdefault(k.split("<")[0], []).append(k)
    for b, ks in bare.items():
        if len(ks) == 1 and b not in res["kernels"]:
            res["kernels"][b] = res["kernels"][ks[0]]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in res["kernels"].items()}), "MB/launch; factors", fetch_factor, write_factor)


if __name__ == "__main__":
    main(*sys.argv[1:6])
