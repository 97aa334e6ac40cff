"""Camera gradient (gslic_rasterize_backward_camera; the "cam" of the north-star).  The reference returns NO gradient for its camera inputs
(src/rasterizer/rasterizer.cpp:171-182), so the pin is mathematical: the analytic gradient of the DOUBLE-precision oracle w.r.t. every
entry of viewmatrix / projmatrix / campos must agree with central finite differences of the double-precision oracle's own forward
(loss = <dL/dimage, image>), on scenes whose discrete decisions (culling, tile lists, alpha cuts) do not move within the step;
the HIP kernels are then held to the float oracle."""
import numpy as np
import pytest
import torch

from conftest import make_scene


def _loss(orc, sc, cam, dL):
    f = orc.forward(sc, cam)
    return float((np.asarray(f["color"], np.float64) * dL).sum()), f


def test_oracle_camera_gradient_matches_finite_differences(oracle64):
    W, H = 64, 48
    raw, sc, camd, cam = make_scene("random", 60, W, H, 3, 4)
    sc = {k: (np.asarray(v, np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v) for k, v in sc.items()}
    camd = dict(camd)
    for k in ("view", "proj", "campos"):
        camd[k] = np.asarray(camd[k], np.float64).copy()
    rng = np.random.default_rng(0)
    dL = rng.standard_normal((3, H, W))
    l0, f0 = _loss(oracle64, sc, camd, dL)
    g = oracle64.backward(sc, camd, f0, dL, camera_grads=True)
    checked = 0
    for name, key, idxs in (("dL_dviewmatrix", "view", [4 * c + r for c in range(4) for r in range(3)]),
                            ("dL_dprojmatrix", "proj", [4 * c + r for c in range(4) for r in (0, 1, 3)]),
                            ("dL_dcampos", "campos", [0, 1, 2])):
        scale = max(float(np.abs(g[name]).max()), 1e-12)
        for i in idxs:
            h = 1e-6 * max(1.0, abs(float(camd[key].reshape(-1)[i])))
            cp, cm = dict(camd), dict(camd)
            cp[key] = camd[key].copy(); cp[key].reshape(-1)[i] += h
            cm[key] = camd[key].copy(); cm[key].reshape(-1)[i] -= h
            lp, fp = _loss(oracle64, sc, cp, dL)
            lm, fm = _loss(oracle64, sc, cm, dL)
            if fp["num_rendered"] != f0["num_rendered"] or fm["num_rendered"] != f0["num_rendered"]:
                continue  # a tile decision moved inside the step: the finite difference straddles a discontinuity
            fd = (lp - lm) / (2 * h)
            assert abs(fd - float(g[name][i])) <= 2e-4 * scale + 1e-7, (name, i, fd, float(g[name][i]))
            checked += 1
        # rows that carry no gradient stay exactly zero
        for i in range(g[name].size):
            if i not in idxs:
                assert g[name][i] == 0.0
    assert checked >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("P,W,H,deg,seed", [(3000, 160, 120, 3, 3), (20000, 320, 240, 3, 8), (5000, 70, 50, 0, 9),
                                             (60, 64, 48, 0, 4), (40, 64, 48, 1, 5)])   # P < 64 on the generic path: the partial rows used to overlap the outputs
def test_hip_camera_gradient_matches_oracle(oracle32, P, W, H, deg, seed):
    from gpu_helpers import hip_forward
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene("random", P, W, H, deg, seed)
    dL = pixel_grad(H, W, seed=1)
    ref_f = oracle32.forward(sc, camd)
    ref = oracle32.backward(sc, camd, ref_f, dL.numpy(), camera_grads=True)
    f = hip_forward(raw, cam)
    act, rs = f["act"], f["rs"]
    dev = act["means"].device
    e = torch.empty(0, device=dev)
    geom, binning, img, sample = f["bufs"]
    out = rz.rasterize_gaussians_backward(rs.bg, act["means"], f["radii"], e, act["scales"], act["rots"], 1.0, e, rs.viewmatrix, rs.projmatrix,
                                          rs.tanfovx, rs.tanfovy, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos, dL.to(dev), act["dc"],
                                          act["shs"], act["D"], rs.campos, geom, f["R"], binning, img, f["B"], sample, 0.0, False,
                                          camera_grads=True)
    plain = rz.rasterize_gaussians_backward(rs.bg, act["means"], f["radii"], e, act["scales"], act["rots"], 1.0, e, rs.viewmatrix, rs.projmatrix,
                                            rs.tanfovx, rs.tanfovy, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos, dL.to(dev), act["dc"],
                                            act["shs"], act["D"], rs.campos, geom, f["R"], binning, img, f["B"], sample, 0.0, False)
    for a, b in zip(out[:9], plain):   # the ordinary gradients are the same (another instantiation of the kernel: fma contraction may differ)
        if a.numel():
            assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-30)
    for got, name in zip(out[9:], ("dL_dviewmatrix", "dL_dprojmatrix", "dL_dcampos")):
        r = np.asarray(ref[name], np.float64)
        gth = got.cpu().numpy().astype(np.float64)
        scale = max(float(np.abs(r).max()), 1e-30)
        assert float(np.abs(gth - r).max()) / scale < 2e-4, (name, gth, r)
        assert np.all(gth[r == 0.0] == 0.0)
    # run-to-run bit reproducibility of the reduction
    again = rz.rasterize_gaussians_backward(rs.bg, act["means"], f["radii"], e, act["scales"], act["rots"], 1.0, e, rs.viewmatrix, rs.projmatrix,
                                            rs.tanfovx, rs.tanfovy, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos, dL.to(dev), act["dc"],
                                            act["shs"], act["D"], rs.campos, geom, f["R"], binning, img, f["B"], sample, 0.0, False,
                                            camera_grads=True)
    for a, b in zip(out[9:], again[9:]):
        assert torch.equal(a, b)


def _camera_f64(cam, R_cw, t_cw):
    """The camera dict of `cam` (intrinsics, limits) at the world-to-camera pose [R_cw | t_cw], every matrix in float64 (camera.h:70-110 without
    its float members): what finite differences of the double-precision oracle need."""
    V = np.eye(4); V[:3, :3] = R_cw; V[:3, 3] = t_cw
    Pm = np.asarray(cam.projection_matrix, np.float64).T
    d = dict(cam.as_dict())
    d["view"] = np.ascontiguousarray(V.T).reshape(16).copy()
    d["proj"] = np.ascontiguousarray((Pm @ V).T).reshape(16).copy()
    d["campos"] = (-R_cw.T @ t_cw).copy()
    return d


@pytest.mark.parametrize("view,extra,seed", [(0, (0.0, 0.0, 0.0), 4), (6, (0.06, -0.04, 0.03), 5), (7, (-0.05, 0.0, 0.08), 6)])
def test_pose_gradient_chain_matches_finite_differences(oracle64, view, extra, seed):
    """Camera.pose_gradient — the chain from (dL/dviewmatrix, dL/dprojmatrix, dL/dcampos) to the six coordinates of a left se(3) increment of
    the pose — against central finite differences of the double-precision oracle's loss over xi, at rotated and translated poses (yaw of the
    rig + an extra pitch / roll), including Gaussians whose cov2D Jacobian is clamp-masked (|t.x / t.z| beyond lim: backward.cu:225-233)."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd.camera import se3_exp, synthetic_camera
    W, H = 64, 48
    raw, sc, camd0, _ = make_scene("random", 80, W, H, 3, seed)
    sc = {k: (np.asarray(v, np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v) for k, v in sc.items()}
    cam = synthetic_camera(W, H, view)
    cam.apply_pose_increment([0.0, 0.0, 0.0, *extra])
    R_cw, t_cw = cam.R_wc.T.copy(), -cam.R_wc.T @ cam.t_wc
    camd = _camera_f64(cam, R_cw, t_cw)
    rng = np.random.default_rng(1)
    dL = rng.standard_normal((3, H, W))
    l0, f0 = _loss(oracle64, sc, camd, dL)
    g = oracle64.backward(sc, camd, f0, dL, camera_grads=True)
    # the clamp case is present: a visible Gaussian whose camera-space direction lies outside the frustum limits
    vis = f0["pre"]["radii"] > 0
    p = np.asarray(sc["means"], np.float64)[vis]
    tcam = p @ R_cw.T + t_cw
    ratio_x, ratio_y = tcam[:, 0] / tcam[:, 2], tcam[:, 1] / tcam[:, 2]
    clamped = int(((ratio_x < camd["limx_neg"]) | (ratio_x > camd["limx_pos"]) | (ratio_y < camd["limy_neg"]) | (ratio_y > camd["limy_pos"])).sum())
    if view in (0, 7):
        assert clamped > 0, "this pose was chosen to have clamp-masked Gaussians"
    cam.world_view_transform = camd["view"].reshape(4, 4)          # (float64 matrices for the chain)
    analytic = cam.pose_gradient(g["dL_dviewmatrix"], g["dL_dprojmatrix"], g["dL_dcampos"])
    scale = max(float(np.abs(analytic).max()), 1e-12)
    checked = 0
    def fd(i, h):
        ls, ok = [], True
        for sgn in (+1.0, -1.0):
            xi = np.zeros(6); xi[i] = sgn * h
            E = se3_exp(xi)
            cd = _camera_f64(cam, E[:3, :3] @ R_cw, E[:3, :3] @ t_cw + E[:3, 3])
            l, f = _loss(oracle64, sc, cd, dL)
            ok = ok and f["num_rendered"] == f0["num_rendered"]
            ls.append(l)
        return (ls[0] - ls[1]) / (2 * h), ok
    for i in range(6):
        # the loss is piecewise smooth in the pose (depth-order swaps, alpha cuts): two step sizes that disagree straddle a jump — skipped
        f1, ok1 = fd(i, 1e-7)
        f2, ok2 = fd(i, 2.5e-8)
        if not (ok1 and ok2) or abs(f1 - f2) > 1e-3 * scale:
            continue
        assert abs(f1 - analytic[i]) <= 5e-4 * scale + 1e-7, (i, f1, analytic[i], clamped)
        checked += 1
    assert checked >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("view,P,W,H", [(0, 20000, 320, 240), (7, 3000, 160, 120)])
def test_hip_pose_gradient_matches_oracle_and_descends(oracle32, view, P, W, H):
    """trainer.pose_gradient (forward -> loss kernels -> gslic_rasterize_backward_camera -> Camera.pose_gradient) at a rotated pose with
    clamp-masked Gaussians: equal to the oracle's chained gradient, and a small step against it lowers the loss."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import trainer
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import gt_image
    raw, sc, _camd, _cam = make_scene("random", P, W, H, 3, 12)
    dev = torch.device("cuda:0")
    cam = synthetic_camera(W, H, view).to_device(dev)
    camd = cam.as_dict()
    model = trainer.GaussianModel(raw, dev)
    gt, bg = gt_image(H, W).to(dev), torch.zeros(3, device=dev)
    g, terms = trainer.pose_gradient(model, cam, gt, bg)
    # the oracle on the same loss: dL/dimage from the loss kernels' definition, then the same chain
    f = oracle32.forward(sc, camd)
    n = float(f["color"].size)
    gtn = gt.cpu().numpy()
    m, d1, d2, d3 = oracle32.ssim_forward(f["color"][None], gtn[None])
    dL = (0.8 / n) * np.sign(f["color"] - gtn).astype(np.float32) + oracle32.ssim_backward(f["color"][None], gtn[None], np.full_like(m, -0.2 / n), d1, d2, d3)[0]
    ref = oracle32.backward(sc, camd, f, dL, camera_grads=True)
    want = cam.pose_gradient(ref["dL_dviewmatrix"], ref["dL_dprojmatrix"], ref["dL_dcampos"])
    assert float(np.abs(g - want).max()) <= 2e-3 * max(float(np.abs(want).max()), 1e-30), (g, want)
    # the joint map + pose step computes the same camera gradient in the same backward that produces the parameter gradients
    model_b = trainer.GaussianModel(raw, dev); model_b.training_setup()
    cam_b = synthetic_camera(W, H, view).to_device(dev)
    _t, _v, g_joint = trainer.training_step_with_pose(model_b, cam_b, gt, bg, pose_lr=0.0)
    assert float(np.abs(g_joint - g).max()) <= 1e-5 * max(float(np.abs(g).max()), 1e-30)
    assert not torch.equal(model_b.xyz, model.xyz)                  # ... and the map was updated
    loss0 = 0.8 * float(terms[0]) + 0.2 * (1.0 - float(terms[1]))
    step = 1e-3 / max(float(np.linalg.norm(g)), 1e-30)            # a 1e-3 (m, rad) move along -gradient
    cam.apply_pose_increment(-step * g).to_device(dev)
    _g2, terms2 = trainer.pose_gradient(model, cam, gt, bg)
    loss1 = 0.8 * float(terms2[0]) + 0.2 * (1.0 - float(terms2[1]))
    assert loss1 < loss0, (loss0, loss1)
