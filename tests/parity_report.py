#!/usr/bin/env python
"""Prints (and writes as JSON) the element-wise comparison of the HIP path with the reference's own kernels at BASELINE's full
sizes, fast and strict arithmetic, plus the yardstick column `ref_contract`: the reference's kernels against themselves under
-ffp-contract=fast vs off — the table of DESIGN.md section 2 comes from this script.

    python tests/parity_report.py [--out gpurun_out/parity_report.json] [--configs small,c2,c3,c5]

Test infrastructure (uses oracle/_ref); needs the MI355X."""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

CONFIGS = {
    "small": ("random", 125184, 480, 270, 3, 0),        # the 1/16-scale sample bench.py's cpu_baseline leg uses
    "c2": ("lidar", 500224, 1920, 1080, 3, 0),          # BASELINE config 2
    "c3": ("random", 2000128, 1920, 1080, 3, 0),        # BASELINE config 3 / 4 (the headline size)
    "c5": ("random", 5000192, 3840, 2160, 3, 0),        # BASELINE config 5
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join("gpurun_out", "parity_report.json"))
    ap.add_argument("--configs", default="small,c2,c3,c5")
    ap.add_argument("--no-ref-contract", action="store_true", help="skip the reference-vs-reference (contraction on / off) yardstick column")
    args = ap.parse_args()
    from oracle.ref_build import refkernels
    from refcompare import compare, compare_reference_builds, summarize, summarize_reference_builds
    results = {}
    for name in args.configs.split(","):
        t0 = time.time()
        res = compare(*CONFIGS[name])
        if not args.no_ref_contract and refkernels.available(fma=True):
            res["ref_contract"] = compare_reference_builds(*CONFIGS[name])
        res["seconds"] = round(time.time() - t0, 1)
        results[name] = res
        print(f"== {name} ({res['seconds']} s)\n" + summarize(res), flush=True)
        if "ref_contract" in res:
            print(summarize_reference_builds(res["ref_contract"]), flush=True)
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
