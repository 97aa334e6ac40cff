"""CPU: independent cross-checks of the oracle that share no code with it."""
import math

import numpy as np
import pytest
import torch

from conftest import make_scene, rel_err


def _f64(sc):
    return {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}


def test_canonical_logf_within_one_ulp(oracle32):
    xs = np.concatenate([np.linspace(1.0, 255.0, 4001), np.exp(np.linspace(-20, 20, 2001)), [1.0, 2.0, 0.5, 254.99998]]).astype(np.float32)
    for x in xs:
        got, ref = oracle32.logf(float(x)), math.log(float(x))
        ulp = np.spacing(np.float32(abs(ref))) if ref != 0 else np.float32(1e-45)
        assert abs(got - ref) <= 1.0 * float(ulp) + 1e-45, (x, got, ref)


def test_higher_msb(oracle32):
    for n, want in [(1, 1), (2, 2), (3, 2), (1200, 11), (8160, 13), (32400, 15), (2 ** 31, 32)]:
        assert oracle32.lib.orc_higher_msb(n) == want


def test_backward_matches_finite_differences(oracle64):
    """Analytic backward (restated backward.cu) vs central differences of the restated forward, in float64."""
    W, H = 64, 48
    raw, sc, camd, cam = make_scene("random", 60, W, H, 3, 3)
    sc = _f64(sc)
    rng = np.random.default_rng(0)
    wimg = rng.standard_normal((3, H, W))

    def loss(s):
        f = oracle64.forward(s, camd)
        return float((f["color"] * wimg).sum()), f

    _, f0 = loss(sc)
    g = oracle64.backward(sc, camd, f0, wimg)
    names = dict(means="dL_dmean3D", scales="dL_dscale", rots="dL_drot", opac="dL_dopacity", dc="dL_ddc", shs="dL_dsh")
    for k, gk in names.items():
        an = g[gk].reshape(sc[k].shape)
        errs = []
        for _ in range(25):
            ix = tuple(int(rng.integers(0, s)) for s in sc[k].shape)
            eps = 1e-6 * max(1.0, abs(sc[k][ix]))
            s2 = dict(sc)
            a = sc[k].copy(); a[ix] += eps; s2[k] = a
            lp, _ = loss(s2)
            a = sc[k].copy(); a[ix] -= eps; s2[k] = a
            lm, _ = loss(s2)
            errs.append(((lp - lm) / (2 * eps), an[ix]))
        errs = np.array(errs)
        rel = np.abs(errs[:, 0] - errs[:, 1]) / (np.abs(errs).max() + 1e-12)
        assert np.median(rel) < 1e-6 and (rel > 1e-4).sum() <= 1, (k, rel.max())


def test_float32_oracle_tracks_float64_oracle(oracle32, oracle64):
    W, H = 96, 64
    raw, sc, camd, cam = make_scene("random", 400, W, H, 3, 9)
    dL = np.random.default_rng(1).standard_normal((3, H, W)).astype(np.float32)
    f32 = oracle32.forward(sc, camd)
    f64 = oracle64.forward(_f64(sc), camd)
    if not np.array_equal(f32["bins"]["point_list"], f64["bins"]["point_list"]):
        pytest.skip("a threshold decision differs between fp32 and fp64 on this seed")
    assert rel_err(f32["color"], f64["color"]) < 1e-4
    g32 = oracle32.backward(sc, camd, f32, dL)
    g64 = oracle64.backward(_f64(sc), camd, f64, dL.astype(np.float64))
    for k in ("dL_dmean3D", "dL_dscale", "dL_dopacity", "dL_ddc", "dL_dsh", "dL_dmean2D", "dL_dconic"):
        assert rel_err(g32[k].reshape(-1), g64[k].reshape(-1)) < 2e-3, k


def test_blend_against_dense_pytorch_autograd(oracle64):
    """The blend stage (A.3/A.4) re-implemented in plain torch float64 with autograd: image and the four 2D gradients."""
    W, H = 32, 32
    raw, sc, camd, cam = make_scene("random", 150, W, H, 3, 17)
    sc = _f64(sc)
    f = oracle64.forward(sc, camd)
    pre, bins = f["pre"], f["bins"]
    m2 = torch.tensor(pre["means2D"], dtype=torch.float64, requires_grad=True)
    co = torch.tensor(pre["conic_opacity"], dtype=torch.float64, requires_grad=True)
    col = torch.tensor(pre["rgb"], dtype=torch.float64, requires_grad=True)
    gx = (W + 15) // 16
    img = torch.zeros(3, H, W, dtype=torch.float64)
    rows = []
    for y in range(H):
        for x in range(W):
            t = (y // 16) * gx + (x // 16)
            r0, r1 = bins["ranges"][t]
            T = torch.ones((), dtype=torch.float64)
            C = torch.zeros(3, dtype=torch.float64)
            for k in range(int(r0), int(r1)):
                g = int(bins["point_list"][k])
                dx, dy = m2[g, 0] - x, m2[g, 1] - y
                power = -0.5 * (co[g, 0] * dx * dx + co[g, 2] * dy * dy) - co[g, 1] * dx * dy
                if power.item() > 0:
                    continue
                alpha = torch.clamp(co[g, 3] * torch.exp(power), max=0.99)
                if alpha.item() < 1.0 / 255.0:
                    continue
                if (T * (1 - alpha)).item() < 1e-4:
                    break
                C = C + col[g] * alpha * T
                T = T * (1 - alpha)
            rows.append(C)
    img = torch.stack(rows).reshape(H, W, 3).permute(2, 0, 1)
    assert rel_err(img.detach().numpy(), f["color"]) < 1e-10
    w = torch.tensor(np.random.default_rng(3).standard_normal((3, H, W)))
    (img * w).sum().backward()
    g = oracle64.backward(sc, camd, f, w.numpy())
    # oracle dL_dmean2D is in NDC-scaled units (x 0.5 W, x 0.5 H): backward.cu:464-465,572-575
    assert rel_err(g["dL_dmean2D"][:, 0], m2.grad[:, 0].numpy() * 0.5 * W) < 1e-8
    assert rel_err(g["dL_dmean2D"][:, 1], m2.grad[:, 1].numpy() * 0.5 * H) < 1e-8
    gc = co.grad.numpy()
    dconic = g["dL_dconic"].reshape(-1, 4)
    assert rel_err(dconic[:, 0], gc[:, 0]) < 1e-8      # d/dA
    assert rel_err(dconic[:, 1], gc[:, 1] * 0.5) < 1e-8  # the kernel stores HALF of d/dB in .y (backward.cu:578)
    assert rel_err(dconic[:, 3], gc[:, 2]) < 1e-8      # d/dC in .w
    assert rel_err(g["dL_dopacity"][:, 0], gc[:, 3]) < 1e-8
    assert rel_err(g["dL_dcolor"], col.grad.numpy()) < 1e-8


def test_properties(oracle32):
    W, H = 128, 96
    raw, sc, camd, cam = make_scene("random", 3000, W, H, 3, 5)
    f = oracle32.forward(sc, camd)
    assert f["final_T"].min() >= 0 and f["final_T"].max() <= 1
    fn = oracle32.forward(sc, camd, no_color=True)
    np.testing.assert_array_equal(fn["final_T"], f["final_T"])          # no_color: identical transmittance
    np.testing.assert_array_equal(fn["pre"]["radii"], f["pre"]["radii"])
    assert float(np.abs(fn["color"]).max()) == 0.0
    # permutation invariance (up to equal-depth ties and fp32 summation order inside a pixel: none here)
    perm = np.random.default_rng(0).permutation(3000)
    scp = {k: (v[perm] if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    fp = oracle32.forward(scp, camd)
    assert rel_err(fp["color"], f["color"]) < 1e-5
    np.testing.assert_array_equal(fp["pre"]["radii"], f["pre"]["radii"][perm])
    # invisible Gaussians receive exactly zero gradient
    dL = np.random.default_rng(1).standard_normal((3, H, W)).astype(np.float32)
    g = oracle32.backward(sc, camd, f, dL)
    inv = f["pre"]["radii"] <= 0
    assert inv.any()
    for k, v in g.items():
        assert float(np.abs(v.reshape(3000, -1)[inv]).max()) == 0.0, k
    # keys sorted, stable
    keys, pl = f["bins"]["keys"], f["bins"]["point_list"].astype(np.int64)
    assert np.all(keys[1:] >= keys[:-1])
    eq = keys[1:] == keys[:-1]
    assert np.all(pl[1:][eq] > pl[:-1][eq])


def test_ssim_against_reference_conv2d_formula(oracle32):
    """fused-SSIM restatement vs the reference's own conv2d SSIM (loss_utils.h:40-128) and its autograd gradient."""
    rng = np.random.default_rng(0)
    B, CH, H, W = 1, 3, 40, 56
    a, b = rng.random((B, CH, H, W)).astype(np.float32), rng.random((B, CH, H, W)).astype(np.float32)
    g1 = torch.tensor([math.exp(-((x - 5) ** 2) / (2.0 * 1.5 * 1.5)) for x in range(11)], dtype=torch.float64)
    g1 = g1 / g1.sum()
    win = (g1[:, None] @ g1[None, :])[None, None].expand(CH, 1, 11, 11).contiguous()
    ta = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tb = torch.tensor(b, dtype=torch.float64)
    conv = lambda x: torch.nn.functional.conv2d(x, win, padding=5, groups=CH)
    mu1, mu2 = conv(ta), conv(tb)
    s1, s2, s12 = conv(ta * ta) - mu1 ** 2, conv(tb * tb) - mu2 ** 2, conv(ta * tb) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
    om, d1, d2, d3 = oracle32.ssim_forward(a, b)
    assert rel_err(om, m.detach().numpy()) < 2e-5
    dL = rng.standard_normal((B, CH, H, W))
    (m * torch.tensor(dL)).sum().backward()
    og = oracle32.ssim_backward(a, b, dL.astype(np.float32), d1, d2, d3)
    assert rel_err(og, ta.grad.numpy()) < 2e-4


def test_knn_against_kdtree(oracle32):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    pts = rng.standard_normal((2000, 3)).astype(np.float32) * np.array([5, 1, 3], np.float32)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    assert rel_err(oracle32.knn(pts), (d[:, 1:] ** 2).mean(1)) < 1e-5
    out = oracle32.knn(pts[:3])  # fewer than 4 points: FLT_MAX stands in (simple_knn.cu:155)
    assert np.all(~np.isfinite(out) | (out > 1e37))


def test_adam_formula(oracle32):
    rng = np.random.default_rng(0)
    N, M = 64, 4
    p, g = rng.standard_normal((N, M)).astype(np.float32), rng.standard_normal((N, M)).astype(np.float32)
    m, v = np.zeros((N, M), np.float32), np.zeros((N, M), np.float32)
    vis = rng.random(N) < 0.5
    p0 = p.copy()
    oracle32.adam(p, g, m, v, vis, 1e-2)
    m_ref, v_ref = 0.1 * g, 0.001 * g * g
    p_ref = p0 - 1e-2 * m_ref / (np.sqrt(v_ref) + 1e-15)
    assert rel_err(p[vis], p_ref[vis]) < 1e-6
    np.testing.assert_array_equal(p[~vis], p0[~vis])
    assert float(np.abs(m[~vis]).max()) == 0.0 and float(np.abs(v[~vis]).max()) == 0.0


@pytest.mark.parametrize("pose", ["se3_a", "se3_b", 6])
def test_rigid_motion_of_scene_and_camera_leaves_the_render_unchanged(oracle64, pose):
    """Pins the oracle's handling of a NON-IDENTITY pose mathematically, independent of the reference: moving the world rigidly (positions and
    Gaussian orientations) and the camera with it must reproduce the identity-pose render — image, final_T, radii, per-tile lists — and rotate the
    position gradients (dL/dxyz_world = R_wc dL/dxyz_cam), at SH degree 0 (no view dependence of the colour left).  A transposed or mis-indexed view
    matrix in transformPoint4x3 / the W of cov2D (auxiliary.h:70-78, forward.cu:101-104, backward.cu:185), or a wrong camera centre, breaks it.
    Double precision on double inputs, so the bar is tight."""
    from gaussian_lic_amd.camera import resolve_view, synthetic_camera
    from gaussian_lic_amd.synthetic import _quat_from_matrix, pixel_grad
    W, H, P = 128, 96, 1500
    raw, sc, camd0, cam0 = make_scene("random", P, W, H, 0, 5)
    R, t, _ = resolve_view(pose)
    cam1 = synthetic_camera(W, H, pose)
    sc0 = {k: (np.asarray(v, np.float64) if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    sc1 = dict(sc0)
    sc1["means"] = sc0["means"] @ R.T + t
    a, b = _quat_from_matrix(R), sc0["rots"]
    ar, ax, ay, az = a
    br, bx, by, bz = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    sc1["rots"] = np.stack([ar * br - ax * bx - ay * by - az * bz, ar * bx + ax * br + ay * bz - az * by,
                            ar * by - ax * bz + ay * br + az * bx, ar * bz + ax * by - ay * bx + az * br], 1)
    # the camera matrices in double as well (camera.py rounds them to fp32 as camera.h does: rebuild them from the pose here)
    camd1 = cam1.as_dict()
    V = np.eye(4); V[:3, :3] = R.T; V[:3, 3] = -R.T @ t
    Pm = np.asarray(cam0.projection_matrix, np.float64).T
    camd1["view"], camd1["proj"], camd1["campos"] = V.T.reshape(16).copy(), (Pm @ V).T.reshape(16).copy(), t.copy()
    f0, f1 = oracle64.forward(sc0, camd0), oracle64.forward(sc1, camd1)
    assert f0["num_rendered"] > 3000
    assert (f0["pre"]["radii"] != f1["pre"]["radii"]).sum() <= 1 and abs(f0["num_rendered"] - f1["num_rendered"]) <= 2
    assert rel_err(f1["color"], f0["color"]) < 1e-6 and rel_err(f1["final_T"], f0["final_T"]) < 1e-6
    dL = pixel_grad(H, W, seed=1).numpy()
    g0, g1 = oracle64.backward(sc0, camd0, f0, dL), oracle64.backward(sc1, camd1, f1, dL)
    assert rel_err(g1["dL_dmean3D"], g0["dL_dmean3D"] @ R.T) < 1e-6
    for k in ("dL_dscale", "dL_dopacity", "dL_ddc", "dL_dmean2D"):
        assert rel_err(g1[k], g0[k]) < 1e-6, k
    # ... and the identity-pose render is NOT reproduced when the camera alone moves (the test has teeth)
    f2 = oracle64.forward(sc0, camd1)
    assert rel_err(f2["color"], f0["color"]) > 1e-2


def test_get_rect_follows_the_reference_order(oracle32, oracle64):
    """getRect (auxiliary.h:46-56) adds the radius, then BLOCK_X, then subtracts 1 — `p.x + max_radius + BLOCK_X - 1` evaluated left to right in
    fp32.  p + radius + 15 is NOT the same function: for p.x = 216 - 2^-16 (= 215.99998474, a mean the reference's ndc2Pix produces) and radius 25,
    (p + 25) + 16 = 256.99998474 is a tie of the binade [256, 512) and rounds to 257.0 (even), so the rectangle ends at tile column 16 (exclusive);
    with + 15 the sum is 255.99998474 exactly and it ends at 15.  The reference's own kernels list the Gaussian in tile column 15 (fuzz scene 845806,
    tools/experiments/tile_count_mismatch_probe.py); round 1-6's restatement dropped it.  One Gaussian in about 3e7."""
    px = float(np.float32(216.0) - np.float32(2.0 ** -16))
    assert np.float32(px) == np.float32(215.99998474)
    assert oracle32.get_rect(px, 151.11269, 25, 20, 12) == (11, 7, 16, 11)
    # a neighbouring mean (one ulp further left: no tie) and the double-precision twin (no tie either) stop one column earlier
    assert oracle32.get_rect(float(np.float32(216.0) - np.float32(2.0 ** -15)), 151.11269, 25, 20, 12)[2] == 15
    assert oracle64.get_rect(px, 151.11269, 25, 20, 12)[2] == 15
    # the literal fp32 sequence in numpy
    s = ((np.float32(px) + np.float32(25.0)) + np.float32(16.0)) - np.float32(1.0)
    assert s == np.float32(256.0) and int(s / np.float32(16.0)) == 16
    assert int(((np.float32(px) + np.float32(25.0)) + np.float32(15.0)) / np.float32(16.0)) == 15
