"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bar: radii / tiles_touched / sorted point_list / ranges / R bit-exact; images and gradients
within 1e-4 of the tensor's max-abs (fp32)."""
import numpy as np
import pytest
import torch

from conftest import make_scene, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4
CASES = [
    ("random", 10000, 640, 480, 3, 0),     # BASELINE config 1 shape, full SH
    ("random", 10000, 640, 480, 0, 0),     # BASELINE config 1: deg 0, sh empty (M = 0)
    ("lidar", 30000, 640, 480, 3, 0),      # 1/16 of config 2
    ("random", 3000, 70, 50, 2, 5),        # ragged image (not a multiple of 16), deg 2
    ("random", 125000, 480, 270, 3, 7),    # 1/16 of config 3 (dense tiles, long lists)
]


@pytest.mark.parametrize("kind,P,W,H,deg,seed", CASES)
def test_forward_backward_parity(oracle32, kind, P, W, H, deg, seed):
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed)
    ref = oracle32.forward(sc, camd)
    got = hip_forward(raw, cam, export=("tiles_touched", "means2D", "depths", "conic_opacity", "rgb", "sorted_keys", "point_list",
                                        "ranges", "n_contrib", "max_contrib"))
    d = got["dbg"]
    # ---- integer stage boundaries: bit-exact
    np.testing.assert_array_equal(npy(got["radii"]), ref["pre"]["radii"])
    np.testing.assert_array_equal(npy(d["tiles_touched"]).astype(np.uint32), ref["pre"]["tiles_touched"])
    assert got["R"] == ref["num_rendered"]
    np.testing.assert_array_equal(npy(d["sorted_keys"]).view(np.uint64), ref["bins"]["keys"])
    np.testing.assert_array_equal(npy(d["point_list"]).astype(np.uint32), ref["bins"]["point_list"])
    np.testing.assert_array_equal(npy(d["ranges"]).astype(np.uint32), ref["bins"]["ranges"])
    # canonical fp32 chain: means2D / depth / conic / opacity are bit-identical too
    vis = ref["pre"]["radii"] > 0
    np.testing.assert_array_equal(npy(d["means2D"])[vis], ref["pre"]["means2D"][vis])
    np.testing.assert_array_equal(npy(d["depths"])[vis], ref["pre"]["depths"][vis])
    np.testing.assert_array_equal(npy(d["conic_opacity"])[vis], ref["pre"]["conic_opacity"][vis])
    assert rel_err(npy(d["rgb"])[vis], ref["pre"]["rgb"][vis]) < 1e-6
    # ---- image
    # (default = strict arithmetic.  The oracle's exp() is the host libm's, the device's is hipcc's expf lowering: both within an ulp
    # of each other, so a pair sitting exactly on a cut could still decide differently; on these seeded cases none does.)
    assert rel_err(npy(got["color"]), ref["color"]) < TOL
    assert rel_err(npy(got["final_T"]), ref["final_T"]) < TOL
    nc = npy(d["n_contrib"]).astype(np.int64)
    assert int((nc != ref["n_contrib"].astype(np.int64)).sum()) == 0
    # bucket count = sum ceil(n_t / 64)
    r = ref["bins"]["ranges"].astype(np.int64)
    assert got["B"] == int(((r[:, 1] - r[:, 0] + 63) // 64).sum())
    # ---- backward
    dL = pixel_grad(H, W, seed=1)
    gref = oracle32.backward(sc, camd, ref, dL.numpy())
    ggot = hip_backward(got, dL)
    for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale", "dL_drot"):
        a_, b_ = ggot[k].reshape(-1).astype(np.float64), gref[k].reshape(-1).astype(np.float64)
        scale = np.abs(b_).max() if b_.size else 0.0
        if k == "dL_drot":  # exactly 0 for isotropic Gaussians: measure against the magnitude of the cancelling terms
            scale = max(scale, float(np.abs(gref["dL_dscale"]).max() * sc["scales"].max()))
        err_ = np.abs(a_ - b_) / max(scale, 1e-30) if b_.size else np.zeros(1)
        assert err_.max() < TOL, f"{k}: {int((err_ > TOL).sum())} elements > {TOL}, max rel err {err_.max():.3e}"
        assert np.all(ggot[k].reshape(P, -1)[~vis] == 0), f"{k}: invisible rows must be exact zeros"


def test_no_color_mode(oracle32):
    """no_color: identical final_T and radii, no sample buffer, B = 0 (forward.cu:338,362,412,446,470)."""
    from gpu_helpers import hip_forward, npy
    raw, sc, camd, cam = make_scene("random", 20000, 320, 240, 3, 2)
    full = hip_forward(raw, cam)
    nc = hip_forward(raw, cam, no_color=True)
    assert nc["B"] == 0 and nc["bufs"][3].numel() == 0
    assert nc["R"] == full["R"]
    np.testing.assert_array_equal(npy(nc["radii"]), npy(full["radii"]))
    np.testing.assert_array_equal(npy(nc["final_T"]), npy(full["final_T"]))
    ref = oracle32.forward(sc, camd, no_color=True)
    assert rel_err(npy(nc["final_T"]), ref["final_T"]) < TOL


def test_empty_and_all_culled():
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene("random", 64, 64, 48, 3, 1)
    # P = 0 (rasterize_points.cu:110,203)
    raw0 = {k: (v[:0] if torch.is_tensor(v) else v) for k, v in raw.items()}
    f0 = hip_forward(raw0, cam)
    assert f0["R"] == 0 and f0["B"] == 0 and float(f0["color"].abs().max()) == 0.0
    # every Gaussian behind the camera: R = 0, T = 1, colour 0, zero gradients
    rawb = dict(raw)
    rawb["xyz"] = raw["xyz"].clone()
    rawb["xyz"][:, 2] = -1.0
    fb = hip_forward(rawb, cam)
    assert fb["R"] == 0 and fb["B"] == 0
    assert float(fb["color"].abs().max()) == 0.0 and float((fb["final_T"] - 1).abs().max()) == 0.0
    assert int(npy(fb["radii"]).max()) == 0
    g = hip_backward(fb, pixel_grad(48, 64))
    assert all(float(np.abs(v).max()) == 0.0 for v in g.values())


def test_determinism_and_lambda_erank(oracle32):
    """No atomics anywhere: two runs give bit-identical gradients; erank regulariser matches (backward.cu:358-375)."""
    from gpu_helpers import hip_backward, hip_forward
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene("random", 20000, 320, 240, 3, 4)
    dL = pixel_grad(240, 320)
    f1 = hip_forward(raw, cam)
    g1 = hip_backward(f1, dL, lambda_erank=0.01)
    f2 = hip_forward(raw, cam)
    g2 = hip_backward(f2, dL, lambda_erank=0.01)
    for k in g1:
        np.testing.assert_array_equal(g1[k], g2[k])
    ref = oracle32.forward(sc, camd)
    gref = oracle32.backward(sc, camd, ref, dL.numpy(), lambda_erank=0.01)
    assert rel_err(g1["dL_dscale"], gref["dL_dscale"]) < TOL


@pytest.mark.parametrize("P,W,H", [(500000, 1920, 1080),     # BASELINE config 2 shape
                                   (2000000, 1920, 1080),    # config 3 / 4: the headline workload of bench.py
                                   (5000000, 3840, 2160)])   # config 5: 32400 tiles, ~15M instances
def test_large_scene_properties(P, W, H):
    """Size-independent properties at BASELINE.json's full sizes (the oracle takes minutes there): sortedness and stability of the
    instance list, ranges partitioning it, multiplicities, bounds of the image statistics; then linearity of the backward in
    dL/dimage and run-to-run bit-reproducibility."""
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene("random", P, W, H, 3, 0)
    f = hip_forward(raw, cam, export=("tiles_touched", "sorted_keys", "point_list", "ranges", "n_contrib", "max_contrib"))
    d = f["dbg"]
    keys = npy(d["sorted_keys"]).view(np.uint64)
    R = f["R"]
    assert R == int(npy(d["tiles_touched"]).astype(np.int64).sum())
    assert np.all(keys[1:] >= keys[:-1]), "keys must be sorted"
    # stable: equal keys keep ascending Gaussian id
    pl = npy(d["point_list"]).astype(np.int64)
    eq = keys[1:] == keys[:-1]
    assert np.all(pl[1:][eq] > pl[:-1][eq])
    # ranges partition [0, R) by tile id
    rg = npy(d["ranges"]).astype(np.int64)
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    cnt = np.bincount(tiles, minlength=rg.shape[0])
    assert np.array_equal(rg[:, 1] - rg[:, 0], cnt)
    nz = cnt > 0
    assert np.array_equal(rg[nz, 0], (np.cumsum(cnt) - cnt)[nz])
    # every instance's Gaussian is visible; multiplicity matches tiles_touched
    tt = npy(d["tiles_touched"]).astype(np.int64)
    assert np.array_equal(np.bincount(pl, minlength=tt.shape[0]), tt)
    T = npy(f["final_T"])
    assert T.min() >= 0.0 and T.max() <= 1.0 and np.isfinite(npy(f["color"])).all()
    ncb = npy(d["n_contrib"]).astype(np.int64)
    assert ncb.max() <= cnt.max()
    assert int(npy(d["max_contrib"]).max()) == int(ncb.max())
    del keys, pl, tiles, d
    # backward: linear in dL/dimage, and bit-reproducible
    d1, d2 = pixel_grad(H, W, seed=1), pixel_grad(H, W, seed=2)
    g1 = hip_backward(f, d1)
    g2 = hip_backward(f, d2)
    g12 = hip_backward(f, d1 + 2.0 * d2)
    g1b = hip_backward(f, d1)
    for k in g1:
        np.testing.assert_array_equal(g1[k], g1b[k])
        lin = g1[k].astype(np.float64) + 2.0 * g2[k].astype(np.float64)
        scale = max(float(np.abs(lin).max()), 1e-30)
        assert float(np.abs(g12[k] - lin).max()) / scale < 2e-4, k


def test_onesweep_sort_variant_is_bit_identical():
    """GSLIC_SORT_ONESWEEP=3 (both sorts single pass per digit with decoupled look-back) must give exactly the default sort's lists."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import make_scene
from gpu_helpers import hip_forward, npy
raw, sc, camd, cam = make_scene('random', 150000, 960, 540, 3, 71)
f = hip_forward(raw, cam, export=('sorted_keys', 'point_list', 'ranges'))
np.savez(sys.argv[1], keys=npy(f['dbg']['sorted_keys']), pl=npy(f['dbg']['point_list']), rg=npy(f['dbg']['ranges']), img=npy(f['color']))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, env in (("a", {}), ("b", {"GSLIC_SORT_ONESWEEP": "3"})):
        path = os.path.join(root, "gpurun_out", f"_sortcmp_{name}.npz") if os.path.isdir(os.path.join(root, "gpurun_out")) else f"/tmp/_sortcmp_{name}.npz"
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", code, path], cwd=root, env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
        os.remove(path)
    for k in ("keys", "pl", "rg", "img"):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])


def test_non_default_stream_and_debug_flag(oracle32):
    """The C-ABI launches on the stream it is given (here a non-default torch stream) and `debug` (sync after every stage,
    CHECK_CUDA in the reference) changes nothing."""
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene("random", 8000, 200, 150, 3, 81)
    base = hip_forward(raw, cam)
    gbase = hip_backward(base, pixel_grad(150, 200))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f = hip_forward(raw, cam, debug=True)
        g = hip_backward(f, pixel_grad(150, 200))
    s.synchronize()
    assert f["R"] == base["R"] and torch.equal(f["color"], base["color"]) and torch.equal(f["final_T"], base["final_T"])
    for k in g:
        np.testing.assert_array_equal(g[k], gbase[k])


def test_error_conventions_on_device():
    """SURVEY.md 8b error conventions through the C-ABI: `prefiltered` with a culled point is an error (the reference __trap()s,
    auxiliary.h:162-166), a NULL allocator result is GSLIC_ERR_ALLOC, colors_precomp is refused; every failure carries a message in gslic_last_error() and leaves the process usable."""
    import ctypes
    from gaussian_lic_amd import _lib
    from gaussian_lic_amd import rasterizer as rz
    from gpu_helpers import hip_forward, settings_from
    from gaussian_lic_amd.synthetic import activate
    raw, sc, camd, cam = make_scene("random", 2000, 96, 64, 3, 91)   # contains points behind the near plane
    dev = torch.device("cuda:0")
    act = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in activate(raw).items()}
    rs = settings_from(cam, 3, dev)
    e = torch.empty(0, device=dev)
    args = lambda pref, colors: (rs.bg, act["means"], colors, act["opac"], act["scales"], act["rots"], 1.0, e, rs.viewmatrix, rs.projmatrix,
                                 rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos,
                                 act["dc"], act["shs"], 3, rs.campos, pref, False, False)
    with pytest.raises(_lib.GslicError, match="prefiltered"):
        rz.rasterize_gaussians(*args(True, e))
    with pytest.raises(_lib.GslicError, match="colors_precomp"):
        rz.rasterize_gaussians(*args(False, torch.zeros(2000, 3, device=dev)))
    # allocator failure: geom callback returns NULL
    L = _lib.lib()
    prm = _lib.RasterParams(2000, 3, 15, 96, 64, rs.tanfovx, rs.tanfovy, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos, 1.0, 0, 0, 0, 0)
    null_cb = _lib.ALLOC_FN(lambda c, n: 0)
    out_c, out_t, radii = torch.zeros(3, 64, 96, device=dev), torch.zeros(64, 96, device=dev), torch.zeros(2000, dtype=torch.int32, device=dev)
    R, B = ctypes.c_int32(0), ctypes.c_int32(0)
    p = _lib.ptr
    rc = L.gslic_rasterize_forward(ctypes.byref(prm), null_cb, None, null_cb, None, null_cb, None, null_cb, None, p(rs.bg), p(act["means"]),
                                   p(act["dc"]), p(act["shs"]), None, p(act["opac"]), p(act["scales"]), p(act["rots"]), None, p(rs.viewmatrix),
                                   p(rs.projmatrix), p(rs.campos), p(out_c), p(out_t), p(radii), ctypes.byref(R), ctypes.byref(B), None)
    assert rc == -3 and b"allocator" in L.gslic_last_error()
    # and the library still works afterwards
    ok = hip_forward(raw, cam)
    assert ok["R"] > 0


@pytest.mark.parametrize("case", ["tiny_image", "one_pixel", "one_tile_long_list", "equal_depths"])
def test_degenerate_shapes(oracle32, case):
    """Shapes at the edges of the launch geometry: an image smaller than one tile, a single pixel, thousands of Gaussians piled on one
    tile (one list of ~50 buckets, every sort key of a pass equal), and exactly equal depths (order = Gaussian id)."""
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd.synthetic import pixel_grad
    if case == "tiny_image":
        raw, sc, camd, cam = make_scene("random", 50, 5, 3, 3, 2)
    elif case == "one_pixel":
        raw, sc, camd, cam = make_scene("random", 20, 1, 1, 3, 3)
    else:
        W, H, P = 48, 32, 3000
        raw, sc, camd, cam = make_scene("random", P, W, H, 3, 4)
        g = torch.Generator().manual_seed(1)
        z = torch.rand(P, generator=g) * 20.0 + 2.0 if case == "one_tile_long_list" else torch.full((P,), 5.0)
        fx, cx, cy = 0.675 * W, 0.4857 * W, 0.5215 * H
        u = 8.0 + torch.randn(P, generator=g)            # all inside tile (0, 0)
        v = 8.0 + torch.randn(P, generator=g)
        raw["xyz"] = torch.stack([(u - cx) * z / fx, (v - cy) * z / fx, z], 1).float().contiguous()
        raw["scaling"] = (torch.log(z / fx) + 0.3).unsqueeze(1).repeat(1, 3).float().contiguous()
        raw["opacity"] = torch.full((P, 1), -3.0)         # faint: the list is consumed to the end
        from gaussian_lic_amd.synthetic import activate, to_numpy
        sc = to_numpy(activate(raw))
    W, H = cam.image_width, cam.image_height
    ref = oracle32.forward(sc, camd)
    got = hip_forward(raw, cam, export=("tiles_touched", "point_list", "ranges"))
    assert got["R"] == ref["num_rendered"]
    R = got["R"]
    np.testing.assert_array_equal(npy(got["radii"]), ref["pre"]["radii"])
    np.testing.assert_array_equal(npy(got["dbg"]["point_list"])[:R].astype(np.uint32), ref["bins"]["point_list"][:R].astype(np.uint32))
    np.testing.assert_array_equal(npy(got["dbg"]["ranges"]).reshape(-1, 2).astype(np.uint32), ref["bins"]["ranges"].astype(np.uint32))
    assert rel_err(npy(got["color"]), ref["color"]) < 1e-4
    dL = pixel_grad(H, W, seed=1)
    g = hip_backward(got, dL)
    rg = oracle32.backward(sc, camd, ref, dL.numpy())
    for k in ("dL_dmean3D", "dL_dopacity", "dL_ddc", "dL_dsh", "dL_dscale"):
        if rg[k].size and np.abs(rg[k]).max() > 0:
            assert rel_err(g[k].reshape(rg[k].shape), rg[k]) < 1e-4, k


def test_math_mode_switched_between_forward_and_backward():
    """The strict backward takes its blend decisions from bits the strict forward recorded.  A forward that ran in the fast mode records
    none: the strict backward must then re-derive them (= the fast backward, bit for bit) instead of reading unwritten memory; a fast
    backward behind a strict forward ignores the bits."""
    from gpu_helpers import hip_backward, hip_forward
    from gaussian_lic_amd import _lib
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene("random", 30000, 320, 240, 3, 9)
    dL = pixel_grad(240, 320, seed=1)
    prev = _lib.set_math_mode(False)
    try:
        f_fast = hip_forward(raw, cam)
        g_ff = hip_backward(f_fast, dL)
        _lib.set_math_mode(True)
        g_fs = hip_backward(f_fast, dL)            # strict backward, no recorded bits
        f_strict = hip_forward(raw, cam)
        g_ss = hip_backward(f_strict, dL)
        _lib.set_math_mode(False)
        g_sf = hip_backward(f_strict, dL)          # fast backward behind a strict forward
    finally:
        _lib.set_math_mode(prev)
    for k in g_ff:
        np.testing.assert_array_equal(g_fs[k], g_ff[k], err_msg=k)
        assert np.all(np.isfinite(g_ss[k])) and np.all(np.isfinite(g_sf[k])), k
        scale = max(float(np.abs(g_ss[k]).max()), 1e-30)
        if k == "dL_drot":
            scale = max(scale, float(np.abs(g_ss["dL_dscale"]).max() * sc["scales"].max()))
        assert float(np.abs(g_sf[k] - g_ss[k]).max()) / scale < 2e-2, k   # the same gradients up to a handful of flipped cuts
