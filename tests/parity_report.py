#!/usr/bin/env python
"""Prints (and writes as JSON) the element-wise comparison of the HIP path with the reference's own kernels at BASELINE's full
sizes, fast and strict arithmetic, plus the yardstick column `ref_contract`: the reference's kernels against themselves under
-ffp-contract=fast vs off — the table of DESIGN.md section 2 comes from this script.

    python tests/parity_report.py [--out gpurun_out/parity_report.json] [--configs small,c2,c3,c5]
    python tests/parity_report.py --poses [--out gpurun_out/parity_poses.json]      # round 5: the per-view table at NON-identity cameras

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
    ap.add_argument("--poses", action="store_true", help="2M Gaussians / 1920x1080 at the eight config-4 views and the four general SE(3) poses of "
                                                         "camera.SE3_POSES, a clamp-masked case and two scale_modifier cases; one line per view")
    ap.add_argument("--timed-path", action="store_true", help="round 6: the configuration bench.py times (rows in Morton order + tie_rank, atomic binning forced, "
                                                              "fused step) against the reference's kernels at full size: the cases of "
                                                              "tests/test_timed_path_reference_gpu.py as a table, the path taken printed per case")
    args = ap.parse_args()
    if args.poses:
        return poses(args)
    if args.timed_path:
        return timed_path(args)
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


def timed_path(args):
    """What tests/test_timed_path_reference_gpu.py asserts, printed: per case the grouping that was forced AND the one that ran (gslic_get_binning_path,
    launch counts of both paths' kernels), the row order, and every parity count; then the fused step on the Morton model against the reference chain."""
    from refcompare import GRADS, compare, summarize
    from test_timed_path_reference_gpu import CASES, fused_step_vs_reference_chain, summarize_fused
    results, rows = {}, []
    for name, scene, view, binning, morton in CASES:
        t0 = time.time()
        res = compare(*scene, modes=("strict",), view=view, binning=binning, morton=morton)
        res["seconds"] = round(time.time() - t0, 1)
        results[name] = res
        print(f"== {name} ({res['seconds']} s)\n" + summarize(res), flush=True)
        rows.append((name, res))
    print("\n| case | binning forced -> path taken (launches atomic / radix kernels) | rows | visible | instances R | integer stages + lists + ranges + geometry + SH colour | "
          "image / final_T / n_contrib | gradient elements over 1e-4 | max gradient error |\n|---|---|---|---|---|---|---|---|---|")
    for name, res in rows:
        st = res["strict"]
        ints = (st["radii_mismatch"] == 0 and st["tiles_touched_mismatch"] == 0 and st["point_list_equal"] and st["ranges_equal"] and
                all(st[k + "_bit_equal"] for k in ("means2D", "depths", "conic_opacity", "rgb")))
        img = st["color"]["bit_equal"] and st["final_T"]["bit_equal"] and st["n_contrib_mismatch"] == 0
        print(f"| {name} | {res['scene']['binning']} -> {st['binning_path']} ({st['path_launches']['atomic']} / {st['path_launches']['radix']}) | "
              f"{'Morton + tie_rank' if res['scene']['morton'] else 'insertion'} | {res['ref']['visible']} | {res['ref']['R']} | {'bit-identical' if ints else 'DIFFERENT'} | "
              f"{'bit-identical' if img else 'DIFFERENT'} | {sum(st[k]['over'] for k in GRADS)} | {max(st[k]['max_rel'] for k in GRADS):.1e} |", flush=True)
    fused = fused_step_vs_reference_chain()
    results["fused_step_vs_reference_chain"] = fused
    print("\n" + summarize_fused(fused), flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1, default=str)


def poses(args):
    """HIP path vs the reference's own kernels at non-identity cameras, full size: what tests/test_pose_reference_gpu.py asserts, as a table."""
    from refcompare import GRADS, compare, summarize
    cases = [(f"config-4 view {k}", dict(view=k)) for k in range(8)]
    cases += [(f"general pose {n}", dict(view=n)) for n in ("se3_a", "se3_b", "se3_c", "se3_d")]
    results, rows = {}, []
    for name, kw in cases:
        t0 = time.time()
        res = compare("random", 2000128, 1920, 1080, 3, 0, **kw)
        res["seconds"] = round(time.time() - t0, 1)
        results[name] = res
        print(f"== {name} ({res['seconds']} s)\n" + summarize(res), flush=True)
        rows.append((name, res))
    extra = [("clamp-masked, se3_c, sigma x4, 100096 Gaussians 320x180", ("random", 100096, 320, 180, 3, 7), dict(view="se3_c", sigma_scale=4.0)),
             ("scale_modifier 0.7, se3_a, 500224 Gaussians", ("random", 500224, 1920, 1080, 3, 3), dict(view="se3_a", scale_modifier=0.7)),
             ("scale_modifier 1.6, se3_c, 500224 Gaussians", ("random", 500224, 1920, 1080, 3, 3), dict(view="se3_c", scale_modifier=1.6))]
    for name, scene, kw in extra:
        res = compare(*scene, **kw)
        results[name] = res
        print(f"== {name}\n" + summarize(res), flush=True)
        rows.append((name, res))
    print("\n| camera | visible | instances R | clamp-masked visible | integer stages + geometry + SH colour | image / final_T / n_contrib (strict) | "
          "gradient elements over 1e-4 (strict) | max gradient error (strict) | fast mode: image elements over 1e-4 |\n|---|---|---|---|---|---|---|---|---|")
    for name, res in rows:
        st, fa = res["strict"], res["fast"]
        ints = (st["radii_mismatch"] == 0 and st["tiles_touched_mismatch"] == 0 and st["point_list_equal"] and st["ranges_equal"] and
                all(st[k + "_bit_equal"] for k in ("means2D", "depths", "conic_opacity", "rgb")))
        img = st["color"]["bit_equal"] and st["final_T"]["bit_equal"] and st["n_contrib_mismatch"] == 0
        print(f"| {name} | {res['ref']['visible']} | {res['ref']['R']} | {res['ref']['clamp_masked_visible']} | {'bit-identical' if ints else 'DIFFERENT'} | "
              f"{'bit-identical' if img else 'DIFFERENT'} | {sum(st[k]['over'] for k in GRADS)} | {max(st[k]['max_rel'] for k in GRADS):.1e} | "
              f"{fa['color']['over']} of {fa['color']['n']} |", flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1, default=str)


if __name__ == "__main__":
    main()
