"""-m gpu: the backward split for the N > 1 exchange.  gslic_rasterize_backward_rgb writes the clamp-masked colour gradient instead of
dL_ddc / dL_dsh; gslic_sh_grad_from_rgb rebuilds the summed rows of several views from those 3-float vectors.  Both against the plain
per-view gslic_rasterize_backward outputs: bit for bit."""
import numpy as np
import pytest
import torch

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _views(P, W, H, deg, n):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import pixel_grad
    dev = torch.device("cuda:0")
    raw, _sc, _camd, _cam = make_scene("random", P, W, H, deg, 23)
    t = {k: raw[k].to(dev).contiguous() for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")}
    e, bg = torch.empty(0, device=dev), torch.zeros(3, device=dev)
    out = []
    for k in range(n):
        cam = synthetic_camera(W, H, k).to_device(dev)
        dL = pixel_grad(H, W, seed=1 + k).to(dev)
        fw = rz.rasterize_gaussians(bg, t["xyz"], e, t["opacity"], t["scaling"], t["rotation"], 1.0, e, cam.d_world_view_transform,
                                    cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy), H, W, float(cam.limx_neg), float(cam.limx_pos),
                                    float(cam.limy_neg), float(cam.limy_pos), t["features_dc"], t["features_rest"], deg, cam.d_camera_center, False, False,
                                    False, raw_params=True)
        R, B, _img, _fT, radii, geom, binning, img, sample = fw
        args = (bg, t["xyz"], radii, e, t["scaling"], t["rotation"], 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx),
                float(cam.tanfovy), float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dL, t["features_dc"],
                t["features_rest"], deg, cam.d_camera_center, geom, R, binning, img, B, sample, 0.0, False)
        dense = rz.rasterize_gaussians_backward(*args, raw_params=True)
        names = ("xyz", "opacity", "scaling", "rotation")
        o = {"xyz": torch.empty(P, 3, device=dev), "opacity": torch.empty(P, 1, device=dev), "scaling": torch.empty(P, 3, device=dev),
             "rotation": torch.empty(P, 4, device=dev)}
        rgb = torch.full((P, 3), float("nan"), device=dev)
        rz.rasterize_gaussians_backward(*args, raw_params=True, out=o, rgb_out=rgb)
        out.append(dict(cam=cam, dense=dense, split=o, rgb=rgb, names=names, radii=radii))
    return t, out


@pytest.mark.parametrize("deg", [3, 1, 0])
def test_backward_rgb_and_rebuild_equal_the_dense_backward(deg):
    from gaussian_lic_amd import rasterizer as rz
    P, W, H, n = 20000, 320, 240, 3
    t, views = _views(P, W, H, deg, n)
    dev = t["xyz"].device
    for v in views:
        _m2d, _col, d_op, d_xyz, _cov, d_dc, d_sh, d_sc, d_rot = v["dense"]
        for name, ref in (("xyz", d_xyz), ("opacity", d_op), ("scaling", d_sc), ("rotation", d_rot)):
            assert torch.equal(v["split"][name], ref), name          # the four all-reduced groups: the same numbers
        assert torch.equal(v["rgb"] * 0.28209479177387814, d_dc.view(P, 3))    # dL_ddc = SH_C0 * (what is shipped)
        assert float(v["rgb"][v["radii"] <= 0].abs().max()) == 0.0            # invisible Gaussians: exact zeros
    rgb_all = torch.stack([v["rgb"] for v in views]).contiguous()
    campos_all = torch.stack([v["cam"].d_camera_center for v in views]).contiguous()
    M = t["features_rest"].shape[1]
    ddc = torch.full((P, 1, 3), float("nan"), device=dev)
    dsh = torch.full((P, M, 3), float("nan"), device=dev)
    rz.sh_grad_from_rgb(t["xyz"], campos_all, rgb_all, deg, ddc, dsh)
    want_dc = (views[0]["dense"][5] + views[1]["dense"][5]) + views[2]["dense"][5]      # view order, like the kernel
    assert torch.equal(ddc, want_dc)
    if M:
        want_sh = (views[0]["dense"][6] + views[1]["dense"][6]) + views[2]["dense"][6]
        assert torch.equal(dsh, want_sh)
        assert float(want_sh.abs().max()) > 0 or deg == 0
    # hosts that only see the reference's gradient tensors ship dL_ddc instead: dRGB recovered to 1 ulp
    ddc2, dsh2 = torch.empty_like(ddc), torch.empty_like(dsh)
    dc_all = torch.stack([v["dense"][5].view(P, 3) for v in views]).contiguous()
    rz.sh_grad_from_rgb(t["xyz"], campos_all, dc_all, deg, ddc2, dsh2, input_is_ddc=True)
    assert torch.equal(ddc2, want_dc)
    if M:
        s = float(want_sh.abs().max())
        assert float((dsh2 - want_sh).abs().max()) <= 1e-6 * max(s, 1e-30)


@pytest.mark.parametrize("deg,P", [(3, 20000), (3, 20037), (0, 5000)])
def test_rebuild_with_adam_inside_equals_rebuild_then_adam(deg, P):
    """gslic_sh_grad_from_rgb_adam (rows rebuilt and consumed by the masked Adam in one kernel) against gslic_sh_grad_from_rgb followed by
    SparseGaussianAdam.step(only=[1, 2]): parameters and both moments bit for bit, over three steps; P = 20037 leaves a partial last block."""
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd import trainer
    W, H, n = 320, 240, 2
    t, views = _views(P, W, H, deg, n)
    dev = t["xyz"].device
    raw = {k: v.cpu() for k, v in t.items()}
    raw["sh_degree"] = deg
    rgb_all = torch.stack([v["rgb"] for v in views]).contiguous()
    campos_all = torch.stack([v["cam"].d_camera_center for v in views]).contiguous()
    vis = (views[0]["radii"] > 0) | (views[1]["radii"] > 0)
    models = []
    for fused in (False, True):
        m = trainer.GaussianModel({k: (v.clone() if torch.is_tensor(v) else v) for k, v in raw.items()}, dev)
        m.training_setup()
        for step in range(3):
            m.optimizer.set_visibility_and_N(vis, P)
            scale = 1.0 + 0.5 * step        # different gradients every step so that the moments matter
            if fused:
                m.optimizer.step_sh_from_rgb(m.xyz.detach(), campos_all, (rgb_all * scale).contiguous(), deg)
            else:
                ddc, dsh = torch.empty_like(m.features_dc), torch.empty_like(m.features_rest)
                rz.sh_grad_from_rgb(m.xyz.detach(), campos_all, (rgb_all * scale).contiguous(), deg, ddc, dsh)
                m.optimizer.step([None, ddc, dsh, None, None, None], only=[1, 2])
        models.append(m)
    a, b = models
    for i in (1, 2):
        assert torch.equal(a.optimizer.params[i], b.optimizer.params[i]), i
        if a.optimizer.params[i].numel():
            assert torch.equal(a.optimizer.state[i]["exp_avg"], b.optimizer.state[i]["exp_avg"]), i
            assert torch.equal(a.optimizer.state[i]["exp_avg_sq"], b.optimizer.state[i]["exp_avg_sq"]), i
    if deg > 0:   # (with an empty sh tensor the SH backward is skipped altogether, like the reference's `if (shs)`, backward.cu:352: zero gradients)
        assert not torch.equal(a.features_dc, raw["features_dc"].to(dev))     # the update did something
