"""-m gpu: reference-kernel parity at NON-IDENTITY camera poses (round-4 review, item 1).  Until round 5 every comparison with the reference's
own kernels (oracle/_ref/libref_hip.so) used the identity pose, where `transformPoint4x3` (auxiliary.h:70-78), the `W` matrix of cov2D
(forward.cu:101-104, backward.cu:185), `dir = pos - campos` (forward.cu:32, backward.cu:30) and the transposed storage of the camera matrices
(camera.h:86,109) are exercised only in their degenerate form.  Here, HIP path vs the reference's kernels on the same MI355X and inputs:

  * all eight config-4 views (yaw (k - 3.5) * 4 deg, x = (k - 3.5) * 0.25 m) at 2M Gaussians / 1920x1080;
  * four general SE(3) poses (camera.SE3_POSES: yaw, pitch AND roll of 20-40 deg, translation on all three axes -> dense W, campos != 0), two with
    the scene moved into the pose's frame (everything visible), two looking at the identity-frame scene from the side (Gaussians leave the image);
  * a case whose visible Gaussians are clamp-masked by the frustum limits (forward.cu:91-94; backward.cu:177-178,248-249 zero their x / y chain):
    large Gaussians that reach the image from beyond the 15 % margins — the count is asserted, not assumed;
  * scale_modifier != 1 (renderer.h:36, forward.cu:120-149, backward.cu:257-310) at a posed camera;
  * 40 fuzz scenes that draw a random pose (and, for some, a scale_modifier and an extent scale) each.

Bars as in the identity-pose suites, no allowance in the default (strict) arithmetic: radii, tiles_touched, per-tile lists, ranges, means2D / depth /
conic / opacity / SH colour bit-exact; image, final_T, n_contrib bit-identical; every gradient with ZERO elements beyond 1e-4 of the tensor's max-abs.
The fast arithmetic (opt-in) runs on the same scenes with its threshold-flip allowance.  `python tests/parity_report.py --poses` prints the per-view
table (profiles/r05_parity_poses.{json,log})."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FAST_OVER_PPM = 10.0


def _need_ref():
    from oracle.ref_build import refkernels
    if not refkernels.available():
        pytest.skip("oracle/_ref/libref_hip.so not built")


def assert_strict_parity(res, fast_small=False):
    from refcompare import GRADS, assert_path
    assert_path(res)   # (the grouping compare() forced is the one that ran)
    for mode in ("fast", "strict"):
        if mode not in res:
            continue
        st = res[mode]
        assert st["radii_mismatch"] == 0, (mode, st["radii_mismatch"])
        assert st["tiles_touched_mismatch"] == 0, (mode, st["tiles_touched_mismatch"])
        assert st["R"] == res["ref"]["R"]
        assert st["point_list_equal"] and st["ranges_equal"], mode
        assert st["means2D_bit_equal"] and st["depths_bit_equal"] and st["conic_opacity_bit_equal"] and st["rgb_bit_equal"], mode
    st = res["strict"]
    assert st["color"]["bit_equal"] and st["final_T"]["bit_equal"] and st["n_contrib_mismatch"] == 0, (st["color"], st["final_T"], st["n_contrib_mismatch"])
    for k in GRADS:
        # zero elements beyond 1e-4 — except where fp32 cannot deliver 1e-4: refcompare's conditioning probe (small scenes only) counts the over-elements
        # at which the sequential fp32 C oracle itself is more than 1e-5 of the tensor's max-abs away from the double-precision oracle
        # (catastrophic cancellation in the cov2D chain of a Gaussian almost touching the camera: the reference's own atomics reorder the same sums),
        # and except where the reference's OWN run-to-run scatter is what crosses the bar: its backward sums with atomicAdd, the same element of case 20 is
        # 6.4e-5 ... 1.03e-4 away from the (deterministic) HIP value in twelve runs of the reference's kernels (profiles/r06ag_pose_case20_*.log), so an
        # element over the bar against the first run is held against up to eight more and counts only when it is over against all of them AND further from
        # the double-precision oracle than the reference's own run is (at that element the HIP value is 3.3e-5 from fp64, the reference 7.1e-5).  All kinds
        # are printed by summarize() above.
        assert st[k]["over"] == st[k].get("over_excused", st[k].get("over_ill_conditioned", 0)) and st[k]["max_rel"] < 1e-3, (k, st[k])
    if "fast" in res:
        st = res["fast"]
        if fast_small:   # small scenes: a handful of threshold flips, the image never off by more than a contribution
            assert st["color"]["over"] <= max(8, 1e-4 * st["color"]["n"]) and st["color"]["max_rel"] < 5e-2, st["color"]
            assert st["n_contrib_mismatch"] <= max(8, 1e-4 * st["pixels"])
        else:
            for k in ("color", "final_T") + GRADS:
                assert st[k]["over"] <= max(4, FAST_OVER_PPM * 1e-6 * st[k]["n"]), (k, st[k])
            assert st["n_contrib_mismatch"] <= max(4, FAST_OVER_PPM * 1e-6 * st["pixels"])


@pytest.mark.parametrize("k", range(8))
def test_config4_view_matches_reference_kernels_full_size(k):
    """BASELINE config 4's eight cameras, each at the full 2M / 1080p size."""
    _need_ref()
    from refcompare import compare, summarize
    res = compare("random", 2000128, 1920, 1080, 3, 0, view=k)
    print("\n" + summarize(res))
    assert_strict_parity(res)


@pytest.mark.parametrize("pose", ["se3_a", "se3_b", "se3_c", "se3_d"])
@pytest.mark.parametrize("kind,P", [("random", 2000128), ("lidar", 500224)])
def test_general_se3_pose_matches_reference_kernels_full_size(kind, P, pose):
    """General SE(3) poses with pitch and roll: dense W, campos != 0 (camera.SE3_POSES), 1920x1080, both scene kinds."""
    _need_ref()
    from refcompare import compare, summarize
    res = compare(kind, P, 1920, 1080, 3, 0, view=pose)
    print("\n" + summarize(res))
    assert res["ref"]["visible"] > 0.05 * P
    assert_strict_parity(res)


@pytest.mark.parametrize("pose,sigma_scale,P,W,H", [("se3_c", 4.0, 100096, 320, 180), ("se3_d", 3.0, 100096, 480, 270), (5, 5.0, 100096, 320, 180),
                                                    ("se3_c", 3.0, 200192, 640, 360)])
def test_clamp_masked_gaussians_match_reference_kernels(pose, sigma_scale, P, W, H):
    """Visible Gaussians beyond the lim window: forward.cu:91-94 clamps t.x / t.z, t.y / t.z for the Jacobian, backward.cu:177-178,248-249 zero
    the clamped chain.  Large extents on small images make such Gaussians reach the image from beyond the 15 % margins; the test asserts that more
    than a thousand of them are visible (6390 / 1600 / 2566 / 1143 by the CPU oracle).  The posed fuzz scenes add cases with up to 24 000."""
    _need_ref()
    from refcompare import compare, summarize
    res = compare("random", P, W, H, 3, 7, view=pose, sigma_scale=sigma_scale)
    print("\n" + summarize(res))
    assert res["ref"]["clamp_masked_visible"] >= 1000, res["ref"]
    assert_strict_parity(res)


@pytest.mark.parametrize("pose,scale_modifier,kind", [("se3_a", 0.7, "random"), ("se3_c", 1.6, "random"), (None, 0.7, "lidar"), (2, 0.5, "lidar")])
def test_scale_modifier_matches_reference_kernels(pose, scale_modifier, kind):
    """scale_modifier != 1 (renderer.h:36): cov3D = R diag((mod * s)^2) R^T (forward.cu:120-149) and its backward (backward.cu:257-310:
    dL_dscale carries the factor once more)."""
    _need_ref()
    from refcompare import compare, summarize
    res = compare(kind, 500224, 1920, 1080, 3, 3, view=pose, scale_modifier=scale_modifier)
    print("\n" + summarize(res))
    assert_strict_parity(res)


N_FUZZ = 40


def fuzz_cases(n=N_FUZZ, seed0=20260927):
    rng = np.random.default_rng(seed0)
    out = []
    for i in range(n):
        kind = "random" if rng.random() < 0.7 else "lidar"
        P = 256 * int(rng.choice([1, 2, 7, 40, 100, 300]))
        W = int(rng.choice([33, 64, 100, 160, 320, 333, 640]))
        H = int(rng.choice([17, 48, 90, 97, 180, 360]))
        deg = int(rng.integers(0, 4))
        seed = int(rng.integers(0, 10 ** 6))
        ypr = tuple(float(round(v, 2)) for v in rng.uniform(-40.0, 40.0, 3))
        t = tuple(float(round(v, 3)) for v in rng.uniform(-1.5, 1.5, 3))
        view = dict(ypr=ypr, t=t, place=bool(rng.random() < 0.5))
        scale_modifier = float(rng.choice([1.0, 1.0, 0.7, 1.4]))
        sigma_scale = float(rng.choice([1.0, 1.0, 2.5]))
        out.append((kind, P, W, H, deg, seed, view, sigma_scale, scale_modifier))
    return out


FUZZ = fuzz_cases()


def test_fuzz_case_list_has_rolled_and_pitched_poses():
    assert sum(1 for c in FUZZ if abs(c[6]["ypr"][1]) >= 20 and abs(c[6]["ypr"][2]) >= 20) >= 6
    assert sum(1 for c in FUZZ if c[8] != 1.0) >= 8 and sum(1 for c in FUZZ if not c[6]["place"]) >= 10


@pytest.mark.parametrize("case", range(N_FUZZ))
def test_fuzz_scene_at_random_pose_is_bit_identical_to_the_reference_kernels(case):
    _need_ref()
    from refcompare import compare, summarize
    kind, P, W, H, deg, seed, view, sigma_scale, scale_modifier = FUZZ[case]
    binning, morton = (("atomic", True), ("radix", False), ("atomic", False), ("radix", True))[case % 4]   # forced and verified (assert_path); round 6
    res = compare(kind, P, W, H, deg, seed, view=view, sigma_scale=sigma_scale, scale_modifier=scale_modifier, binning=binning, morton=morton)
    print("\n" + summarize(res))
    assert_strict_parity(res, fast_small=True)


GOLDEN_POSED = ["random_1536_160x120_d3_se3a", "lidar_1536_160x120_d3_se3c_mod07", "random_1024_128x96_d2_se3d_clamp"]


@pytest.mark.parametrize("name", GOLDEN_POSED)
def test_hip_matches_posed_golden(name):
    """The committed vectors the reference's kernels produced at general poses (tests/golden/, oracle/ref_build/make_golden.py): the HIP path on
    the stored inputs.  Integer stages and the image bit-exact, gradients within 1e-4."""
    import ast
    import os
    import torch
    from conftest import rel_err
    from gaussian_lic_amd import rasterizer as rz
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    if not os.path.exists(path):
        pytest.skip(name + ".npz not generated yet")
    z = np.load(path)
    meta = ast.literal_eval(str(z["meta"]))
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    means, scales, rots, opac, dc, shs = (t(z["in_" + k]) for k in ("means", "scales", "rots", "opac", "dc", "shs"))
    opac = opac.reshape(-1, 1)
    tanx, tany, lxn, lxp, lyn, lyp = (float(v) for v in z["cam_scalars"])
    mod = float(meta.get("scale_modifier", 1.0))
    empty = torch.empty(0, device=dev)
    view, proj, campos = t(z["cam_view"]).reshape(4, 4), t(z["cam_proj"]).reshape(4, 4), t(z["cam_campos"])
    bg = torch.zeros(3, device=dev)
    out = rz.rasterize_gaussians(bg, means, empty, opac, scales, rots, mod, empty, view, proj, tanx, tany, meta["H"], meta["W"], lxn, lxp, lyn, lyp,
                                 dc, shs, meta["deg"], campos, False, False, False)
    R, B, color, final_T, radii, geom, binning, img, sample = out
    assert R == meta["R"]
    np.testing.assert_array_equal(radii.cpu().numpy(), z["radii"])
    np.testing.assert_array_equal(color.cpu().numpy(), z["color"])
    np.testing.assert_array_equal(final_T.cpu().numpy(), z["final_T"])
    g = rz.rasterize_gaussians_backward(bg, means, radii, empty, scales, rots, mod, empty, view, proj, tanx, tany, lxn, lxp, lyn, lyp, t(z["in_dL_dpix"]),
                                        dc, shs, meta["deg"], campos, geom, R, binning, img, B, sample, meta["lambda_erank"], False)
    names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale", "dL_drot"]
    for n, gt in zip(names, g):
        if n == "dL_drot":
            scale = max(np.abs(z["dL_drot"]).max(), np.abs(z["dL_dscale"]).max() * z["in_scales"].max())
            assert np.abs(gt.cpu().numpy() - z[n]).max() / scale < 1e-4
        else:
            assert rel_err(gt.cpu().numpy().reshape(-1), z[n].reshape(-1)) < 1e-4, n
