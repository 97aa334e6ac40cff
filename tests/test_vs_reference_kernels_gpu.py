"""-m gpu: the HIP path against the REFERENCE's own kernels (oracle/_ref/libref_hip.so, compiled from the
reference sources by oracle/ref_build/build_ref.py) running on the same MI355X, same inputs.
P is always a multiple of 256: with a partial last block the reference's duplicateWithKeys races pad keys over
the last Gaussian's slots (rasterizer_impl.cu:73-131), a reference bug this implementation does not reproduce.
Skipped when the checker library was not built (it is built wherever /root/reference is mounted and travels to
the GPU box with the snapshot)."""
import numpy as np
import pytest

from conftest import assert_close_flips, make_scene, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _ref():
    from oracle.ref_build import refkernels
    if not refkernels.available():
        pytest.skip("oracle/_ref/libref_hip.so not built")
    return refkernels.RefKernels()


@pytest.mark.parametrize("mode", ["default", "fast"])
@pytest.mark.parametrize("kind,P,W,H,deg,seed", [("random", 20480, 640, 480, 3, 21), ("lidar", 61440, 640, 480, 3, 22),
                                                 ("random", 200192, 960, 540, 3, 23)])
def test_hip_matches_reference_kernels(kind, P, W, H, deg, seed, mode):
    """mode "default" = the library as it comes up (strict arithmetic): image / final_T / n_contrib bit-identical to the reference kernels,
    every gradient within 1e-4 with no exceptions.  mode "fast" = gslic_set_math_mode(0), the opt-in variant, with the flip allowance."""
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd import _lib
    from gaussian_lic_amd.synthetic import pixel_grad
    rk = _ref()
    if mode == "fast":
        prev = _lib.set_math_mode(False)
        try:
            _check_against_reference(rk, kind, P, W, H, deg, seed, fast=True)
        finally:
            _lib.set_math_mode(prev)
    else:
        _check_against_reference(rk, kind, P, W, H, deg, seed, fast=False)


def _check_against_reference(rk, kind, P, W, H, deg, seed, fast):
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed)
    dL = pixel_grad(H, W, seed=1)
    ref = rk.run(sc, camd, dL.numpy())
    got = hip_forward(raw, cam, export=("tiles_touched", "means2D", "depths", "conic_opacity", "point_list", "ranges", "n_contrib"))
    d = got["dbg"]
    vis = ref["radii"] > 0
    # integer stages: exact, the culling threshold included (both sides evaluate it with the toolchain's logf, forward.cu:302)
    rad_mis = int((npy(got["radii"]) != ref["radii"]).sum())
    tt_mis = int((npy(d["tiles_touched"]).astype(np.uint32) != ref["tiles_touched"]).sum())
    assert rad_mis == 0, f"{rad_mis} radii differ"
    assert tt_mis == 0, f"{tt_mis} tiles_touched differ"
    assert got["R"] == ref["R"]
    np.testing.assert_array_equal(npy(d["point_list"]).astype(np.uint32), ref["point_list"])
    np.testing.assert_array_equal(npy(d["ranges"]).astype(np.uint32), ref["ranges"])
    np.testing.assert_array_equal(npy(d["means2D"])[vis], ref["means2D"][vis])
    np.testing.assert_array_equal(npy(d["depths"])[vis], ref["depths"][vis])
    np.testing.assert_array_equal(npy(d["conic_opacity"])[vis], ref["conic_opacity"][vis])
    g = hip_backward(got, dL)
    if fast:
        assert_close_flips(npy(got["color"]), ref["color"], TOL, "color")
        assert_close_flips(npy(got["final_T"]), ref["final_T"], TOL, "final_T")
        for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale"):
            assert_close_flips(g[k], ref[k], TOL, k, flip_bound=2e-2)
    else:
        np.testing.assert_array_equal(npy(got["color"]), ref["color"])
        np.testing.assert_array_equal(npy(got["final_T"]), ref["final_T"])
        np.testing.assert_array_equal(npy(d["n_contrib"]).astype(np.uint32), ref["n_contrib"])
        for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale"):
            assert rel_err(g[k].reshape(-1), ref[k].reshape(-1)) < TOL, k
    scale = max(np.abs(ref["dL_drot"]).max(), np.abs(ref["dL_dscale"]).max() * sc["scales"].max())
    assert np.abs(g["dL_drot"] - ref["dL_drot"]).max() / scale < TOL


@pytest.mark.parametrize("B,CH,H,W", [(1, 3, 50, 70), (2, 3, 270, 480), (1, 3, 33, 17), (1, 1, 64, 96)])
def test_hip_ssim_matches_reference_kernels(B, CH, H, W):
    """fused-SSIM forward / backward of the HIP path against the reference's own kernels (ssim.cu:186-365) on the same MI355X: BIT FOR BIT.
    csrc/ssim.hip is compiled with -ffp-contract=off like the checker, every tap is a separately rounded product and sum in the reference's
    order, so the four maps and the gradient carry the reference's bits whatever the compiler schedules."""
    import torch
    from gaussian_lic_amd import loss
    rk = _ref()
    rng = np.random.default_rng(5)
    a = rng.random((B, CH, H, W)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal((B, CH, H, W)), 0.0, 1.0).astype(np.float32)
    dL = rng.standard_normal((B, CH, H, W)).astype(np.float32)
    rm, r1, r2, r3 = rk.ssim_forward(a, b)
    rg = rk.ssim_backward(a, b, dL, r1, r2, r3)
    ta, tb, tdl = (torch.from_numpy(x).to("cuda:0") for x in (a, b, dL))
    m, d1, d2, d3 = loss.fusedssim(0.01 ** 2, 0.03 ** 2, ta, tb, True)
    for got, ref in ((m, rm), (d1, r1), (d2, r2), (d3, r3)):
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
    g = loss.fusedssim_backward(0.01 ** 2, 0.03 ** 2, ta, tb, tdl, d1, d2, d3)
    np.testing.assert_array_equal(g.cpu().numpy(), rg)
    # inference mode (train = false: empty derivative tensors, ssim.cu:381-388): the same map
    m2 = loss.fusedssim(0.01 ** 2, 0.03 ** 2, ta, tb, False)[0]
    np.testing.assert_array_equal(m2.cpu().numpy(), rm)


@pytest.mark.parametrize("name", ["ssim_1x3x70x50", "ssim_2x3x96x64"])
def test_hip_ssim_matches_golden(name):
    """... and against the committed vectors the reference's kernels produced (tests/golden/ssim_*.npz): bit for bit."""
    import os
    import torch
    from gaussian_lic_amd import loss
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    ta, tb, tdl = (torch.from_numpy(z[k]).to("cuda:0") for k in ("img1", "img2", "dL_dmap"))
    m, d1, d2, d3 = loss.fusedssim(0.01 ** 2, 0.03 ** 2, ta, tb, True)
    for got, key in ((m, "ssim_map"), (d1, "dm_dmu1"), (d2, "dm_dsigma1_sq"), (d3, "dm_dsigma12")):
        np.testing.assert_array_equal(got.cpu().numpy(), z[key])
    g = loss.fusedssim_backward(0.01 ** 2, 0.03 ** 2, ta, tb, tdl, d1, d2, d3)
    np.testing.assert_array_equal(g.cpu().numpy(), z["dL_dimg1"])


@pytest.mark.parametrize("H,W", [(50, 70), (270, 480), (1080, 1920)])
def test_fused_loss_gradient_is_the_reference_chain_bit_for_bit(H, W):
    """gslic_l1_ssim_loss_forward/_backward (L1 folded into the SSIM passes) against the chain the reference's host runs
    (gaussian.cpp:685-691 -> loss_utils.h:30-33,130-193): dL/dimage = (1 - lambda)/N sign(image - gt) + fusedssim_backward(dL_dmap = -lambda/N)
    with the reference's kernels: the same bits (the fused kernel multiplies the derivative maps by dL_dmap where the reference does)."""
    import torch
    from gaussian_lic_amd import loss
    rk = _ref()
    rng = np.random.default_rng(11)
    a = rng.random((1, 3, H, W)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal((1, 3, H, W)), 0.0, 1.0).astype(np.float32)
    a[0, :, :4, :6] = b[0, :, :4, :6]                      # a patch where image == target: sign(0) = 0
    n = float(a.size)
    lam = 0.2
    rm, r1, r2, r3 = rk.ssim_forward(a, b)
    f32 = np.float32   # the two upstream scalars in fp32 arithmetic, as LibTorch's mean / mul backward form them: (1 - lambda) / N and -lambda / N
    w_l1, w_ssim = (f32(1.0) - f32(lam)) / f32(n), -f32(lam) / f32(n)
    ref = w_l1 * np.sign(a - b).astype(np.float32) + rk.ssim_backward(a, b, np.full_like(a, w_ssim), r1, r2, r3)
    fl = loss.FusedLoss(lam)
    dL, terms = fl.forward_backward(torch.from_numpy(a[0]).to("cuda:0"), torch.from_numpy(b[0]).to("cuda:0"))
    np.testing.assert_array_equal(dL.cpu().numpy(), ref[0])
    t = terms.cpu().numpy()
    assert abs(float(t[0]) - float(np.abs(a - b).mean())) < 1e-6 and abs(float(t[1]) - float(rm.mean(dtype=np.float64))) < 1e-6


@pytest.mark.parametrize("P,seed", [(100096, 31), (1500, 32), (5, 34)])
def test_hip_knn_matches_reference_kernels(P, seed):
    """distCUDA2 of the HIP path against the reference's own SimpleKNN::knn (simple_knn.cu:185-221 through wrap_knn.hip) on the same
    MI355X, same points: exact 3-NN on both sides, so only the rounding of the three squared distances may differ."""
    import torch
    from gaussian_lic_amd import knn
    from oracle.ref_build.make_golden import knn_points
    rk = _ref()
    pts = knn_points(P, seed)
    ref = rk.knn(pts)
    got = knn.distCUDA2(torch.from_numpy(pts).to("cuda:0")).cpu().numpy()
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-30)
    print(f"\nknn P={P}: max rel err {err.max():.2e}, elements over 1e-5: {int((err > 1e-5).sum())}")
    assert err.max() < 1e-5


def test_hip_knn_matches_golden():
    import os
    import torch
    from gaussian_lic_amd import knn
    from oracle.ref_build.make_golden import KNN_CASES, knn_points
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, P, seed in KNN_CASES:
        path = os.path.join(gdir, name + ".npz")
        if not os.path.exists(path):
            pytest.skip(f"{name}.npz not generated yet")
        z = np.load(path)
        got = knn.distCUDA2(torch.from_numpy(knn_points(P, seed)).to("cuda:0")).cpu().numpy()
        ref = z["mean_dist2"]
        if P < 4:
            assert np.all(~np.isfinite(got) | (got > 1e37)) and np.all(~np.isfinite(ref) | (ref > 1e37))
        else:
            assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5 and (np.abs(got - ref) / np.maximum(ref, 1e-30)).max() < 1e-5
