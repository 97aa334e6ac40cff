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
