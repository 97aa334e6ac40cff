"""HIP path vs the REFERENCE's own kernels (oracle/_ref/libref_hip.so) on the same MI355X and the same inputs: one function that
runs both and returns every count the parity gate needs — integer-stage mismatches, the exact number of elements over the 1e-4
bar, the maximum error — for the fast and the strict arithmetic of the blend kernels.  Shared by
tests/test_fullsize_reference_gpu.py (asserts) and tests/parity_report.py (prints / writes profiles/*parity*.json).
Test infrastructure: imports oracle/ (the checker)."""
import numpy as np

TOL = 1e-4
GRADS = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale", "dL_drot")


def _err_stats(got, ref, scale=None):
    got = np.asarray(got, np.float64).reshape(-1)
    ref = np.asarray(ref, np.float64).reshape(-1)
    if ref.size == 0:
        return dict(n=0, over=0, max_rel=0.0, bit_equal=True)
    scale = max(float(np.abs(ref).max()), 1e-30) if scale is None else scale
    err = np.abs(got - ref) / scale
    return dict(n=int(ref.size), over=int((err > TOL).sum()), max_rel=float(err.max()), bit_equal=bool(np.array_equal(got, ref)))


def compare(kind, P, W, H, deg, seed, modes=("fast", "strict"), backward=True):
    """Returns {"scene":…, "ref": {...unit counts}, "<mode>": {stage: stats}}.  P must be a multiple of 256 (with a partial last block
    the reference's duplicateWithKeys races pad keys over the last Gaussian's slots, rasterizer_impl.cu:73-131)."""
    import torch
    from conftest import make_scene
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd import _lib
    from gaussian_lic_amd.synthetic import pixel_grad
    from oracle.ref_build import refkernels
    assert P % 256 == 0
    rk = refkernels.RefKernels()
    raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed)
    dL = pixel_grad(H, W, seed=1)
    ref = rk.run(sc, camd, dL.numpy() if backward else None)
    vis = ref["radii"] > 0
    out = dict(scene=dict(kind=kind, P=P, W=W, H=H, deg=deg, seed=seed),
               ref=dict(R=int(ref["R"]), B32=int(ref["B"]), visible=int(vis.sum())))
    for mode in modes:
        prev = _lib.set_math_mode(mode == "strict")
        try:
            got = hip_forward(raw, cam, export=("tiles_touched", "means2D", "depths", "conic_opacity", "rgb", "point_list", "ranges",
                                                "n_contrib"))
            d = got["dbg"]
            st = {}
            tt_h = npy(d["tiles_touched"]).astype(np.uint32)
            bad = np.nonzero(tt_h != ref["tiles_touched"])[0]
            st["radii_mismatch"] = int((npy(got["radii"]) != ref["radii"]).sum())
            st["tiles_touched_mismatch"] = int(bad.size)
            st["R"] = int(got["R"])
            pl_h, pl_r = npy(d["point_list"]).astype(np.uint32), ref["point_list"]
            if bad.size:
                pl_h, pl_r = pl_h[~np.isin(pl_h, bad)], pl_r[~np.isin(pl_r, bad)]
            st["point_list_equal"] = bool(pl_h.shape == pl_r.shape and np.array_equal(pl_h, pl_r))
            st["ranges_equal"] = bool(np.array_equal(npy(d["ranges"]).astype(np.uint32), ref["ranges"])) if not bad.size else None
            for k, rkey in (("means2D", "means2D"), ("depths", "depths"), ("conic_opacity", "conic_opacity"), ("rgb", "rgb")):
                st[k + "_bit_equal"] = bool(np.array_equal(npy(d[k])[vis], ref[rkey][vis]))
            st["color"] = _err_stats(npy(got["color"]), ref["color"])
            st["final_T"] = _err_stats(npy(got["final_T"]), ref["final_T"])
            nc = npy(d["n_contrib"]).astype(np.uint32)
            st["n_contrib_mismatch"] = int((nc != ref["n_contrib"]).sum())
            st["pixels"] = int(nc.size)
            if backward:
                g = hip_backward(got, dL)
                for k in GRADS:
                    scale = None
                    if k == "dL_drot":   # unnormalised-quaternion gradient: scale of the chain it belongs to (as test_vs_reference_kernels_gpu.py)
                        scale = max(float(np.abs(ref["dL_drot"]).max()), float(np.abs(ref["dL_dscale"]).max() * sc["scales"].max()))
                    st[k] = _err_stats(g[k], ref[k], scale)
            out[mode] = st
            del got
            torch.cuda.empty_cache()
        finally:
            _lib.set_math_mode(prev)
    return out


def compare_reference_builds(kind, P, W, H, deg, seed):
    """The yardstick: the reference's kernels against THEMSELVES under the two contraction settings — oracle/_ref/libref_hip.so
    (-ffp-contract=off, the pin of the parity tests) vs libref_hip_fma.so (-ffp-contract=fast: hipcc's default and the analogue of nvcc's
    default --fmad=true that upstream is built with).  Same statistics as compare(): what one compiler flag moves in the reference's own
    output is the noise floor any "bit-identical to the reference" statement sits on."""
    from conftest import make_scene
    from gaussian_lic_amd.synthetic import pixel_grad
    from oracle.ref_build import refkernels
    assert P % 256 == 0
    raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed)
    dL = pixel_grad(H, W, seed=1).numpy()
    a = refkernels.RefKernels().run(sc, camd, dL)
    b = refkernels.RefKernels(fma=True).run(sc, camd, dL)
    vis = (a["radii"] > 0) & (b["radii"] > 0)
    st = {}
    st["radii_mismatch"] = int((a["radii"] != b["radii"]).sum())
    bad = np.nonzero(a["tiles_touched"] != b["tiles_touched"])[0]
    st["tiles_touched_mismatch"] = int(bad.size)
    st["R"] = [int(a["R"]), int(b["R"])]
    pa, pb = a["point_list"], b["point_list"]
    st["point_list_equal_raw"] = bool(pa.shape == pb.shape and np.array_equal(pa, pb))
    if bad.size:
        pa, pb = pa[~np.isin(pa, bad)], pb[~np.isin(pb, bad)]
    st["point_list_equal"] = bool(pa.shape == pb.shape and np.array_equal(pa, pb))   # after removing the Gaussians whose tile count differs
    st["point_list_order_differences"] = int((pa != pb).sum()) if pa.shape == pb.shape else None
    for k in ("means2D", "depths", "conic_opacity", "rgb"):
        st[k + "_bit_equal"] = bool(np.array_equal(a[k][vis], b[k][vis]))
        st[k + "_elements_differing"] = int((a[k][vis] != b[k][vis]).sum())
    st["color"] = _err_stats(b["color"], a["color"])
    st["final_T"] = _err_stats(b["final_T"], a["final_T"])
    st["n_contrib_mismatch"] = int((a["n_contrib"] != b["n_contrib"]).sum())
    st["pixels"] = int(a["n_contrib"].size)
    for k in GRADS:
        scale = None
        if k == "dL_drot":
            scale = max(float(np.abs(a["dL_drot"]).max()), float(np.abs(a["dL_dscale"]).max() * sc["scales"].max()))
        st[k] = _err_stats(b[k], a[k], scale)
    return st


def summarize_reference_builds(st):
    parts = [f"radii!={st['radii_mismatch']}", f"tiles_touched!={st['tiles_touched_mismatch']}", f"R={st['R']}",
             f"lists_equal={st['point_list_equal']} (order differences {st['point_list_order_differences']})",
             "geometry elements differing: " + ", ".join(f"{k}={st[k + '_elements_differing']}" for k in ("means2D", "depths", "conic_opacity", "rgb")),
             f"color over1e-4={st['color']['over']}/{st['color']['n']} max={st['color']['max_rel']:.2e} bit_equal={st['color']['bit_equal']}",
             f"final_T over={st['final_T']['over']} max={st['final_T']['max_rel']:.2e}", f"n_contrib!={st['n_contrib_mismatch']}/{st['pixels']}"]
    for k in GRADS:
        parts.append(f"{k} over={st[k]['over']}/{st[k]['n']} max={st[k]['max_rel']:.2e}")
    return "  [reference -ffp-contract=fast vs reference -ffp-contract=off] " + "  ".join(parts)


def summarize(res):
    """One line per mode: what a reader of the test log needs."""
    lines = []
    s = res["scene"]
    lines.append(f"{s['kind']} P={s['P']} {s['W']}x{s['H']} deg{s['deg']}: R={res['ref']['R']} visible={res['ref']['visible']}")
    for mode in ("fast", "strict"):
        if mode not in res:
            continue
        st = res[mode]
        npx = st["pixels"]
        parts = [f"radii!={st['radii_mismatch']}", f"tiles_touched!={st['tiles_touched_mismatch']}", f"lists_equal={st['point_list_equal']}",
                 f"geom_bits={'ok' if all(st[k + '_bit_equal'] for k in ('means2D', 'depths', 'conic_opacity')) else 'DIFF'}",
                 f"rgb_bits={'ok' if st['rgb_bit_equal'] else 'diff'}",
                 f"color over1e-4={st['color']['over']}/{st['color']['n']} max={st['color']['max_rel']:.2e} bit_equal={st['color']['bit_equal']}",
                 f"final_T over={st['final_T']['over']} max={st['final_T']['max_rel']:.2e}",
                 f"n_contrib!={st['n_contrib_mismatch']}/{npx}"]
        for k in GRADS:
            if k in st:
                parts.append(f"{k} over={st[k]['over']}/{st[k]['n']} max={st[k]['max_rel']:.2e}")
        lines.append(f"  [{mode}] " + "  ".join(parts))
    return "\n".join(lines)
