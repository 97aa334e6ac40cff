"""-m gpu: a map whose rows are stored in Morton order (trainer.GaussianModel(order="morton"): visibility-coherent memory for the per-Gaussian
kernels) renders, trains, grows and exports exactly as the same map in insertion order.  The only order-dependent step of the path is the tie
rule of the sort — the reference lists Gaussians of equal (tile, depth bits) by ascending index (stable sort of the index-ordered emission,
rasterizer_impl.cu:395-424) — and the forward applies it on the rows' ORIGINAL indices (gslic_raster_params.tie_rank).  The scenes here have
their depths quantised so that thousands of Gaussians tie."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(P, W, H, seed, quantum=0.5, sigma=1.0):
    import math
    from gaussian_lic_amd.synthetic import random_scene
    raw = random_scene(P, W, H, sh_degree=3, seed=seed)
    if sigma != 1.0:   # larger Gaussians: longer per-tile lists (the per-tile depth sort's long-list paths: LDS workgroup kernel, global scratch)
        raw["scaling"] = (raw["scaling"] + math.log(sigma)).contiguous()
    z = raw["xyz"][:, 2]
    zq = torch.where(z > 0.3, (z / quantum).round().clamp_min(1.0) * quantum, z)   # identity camera: depth = z -> many exact depth ties
    raw["xyz"] = torch.stack([raw["xyz"][:, 0] * zq / z, raw["xyz"][:, 1] * zq / z, zq], 1).contiguous()
    return raw


def _models(raw, dev, **kw):
    from gaussian_lic_amd import trainer
    a = trainer.GaussianModel({k: (v.clone() if torch.is_tensor(v) else v) for k, v in raw.items()}, dev, **kw)
    b = trainer.GaussianModel({k: (v.clone() if torch.is_tensor(v) else v) for k, v in raw.items()}, dev, order="morton", **kw)
    a.training_setup(); b.training_setup()
    return a, b


def _same_map(a, b):
    order = b.original_order()
    for n in a.NAMES:
        x, y = getattr(a, n).detach(), getattr(b, n).detach()[order]
        assert torch.equal(x, y), n
    for n in a.NAMES:
        assert torch.equal(a._m[n][:a.P], b._m[n][:b.P][order]) and torch.equal(a._v[n][:a.P], b._v[n][:b.P][order]), n


def test_morton_order_is_a_nontrivial_permutation_with_ties_in_the_scene():
    from gaussian_lic_amd import trainer
    raw = _scene(30000, 320, 192, 5)
    perm = trainer.morton_order(raw["xyz"])
    assert sorted(perm.tolist()) == list(range(30000)) and (perm != torch.arange(30000)).float().mean() > 0.99
    z = raw["xyz"][:, 2]
    assert (z > 0.3).sum() - torch.unique(z[z > 0.3]).numel() > 20000     # (the depths tie)


@pytest.mark.parametrize("P,W,H,seed,sigma", [(30000, 320, 192, 5, 1.0), (200192, 960, 540, 6, 1.0), (60000, 160, 96, 7, 3.0)])
def test_morton_model_renders_and_trains_bit_identically(P, W, H, seed, sigma):
    from gaussian_lic_amd import trainer
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.rasterizer import render
    from gaussian_lic_amd.synthetic import gt_image
    dev = torch.device("cuda:0")
    a, b = _models(_scene(P, W, H, seed, sigma=sigma), dev)
    cam = synthetic_camera(W, H).to_device(dev)
    gt, bg = gt_image(H, W).to(dev), torch.zeros(3, device=dev)
    if sigma != 1.0:   # (the case is there for its long lists: 60 tiles, thousands of instances each — beyond what one wave and what LDS sorts)
        from gaussian_lic_amd import rasterizer as rz
        e = torch.empty(0, device=dev)
        with torch.no_grad():
            R = rz.rasterize_gaussians(bg, a.xyz.detach(), e, a.opacity.detach(), a.scaling.detach(), a.rotation.detach(), 1.0, e, cam.d_world_view_transform,
                                       cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy), H, W, float(cam.limx_neg), float(cam.limx_pos),
                                       float(cam.limy_neg), float(cam.limy_pos), a.features_dc.detach(), a.features_rest.detach(), 3, cam.d_camera_center,
                                       False, False, False, raw_params=True)[0]
        assert R > 4096 * 1.5 * ((W + 15) // 16) * ((H + 15) // 16), R
    with torch.no_grad():
        ia, Ta, _, va, ra = render(cam, a, bg)
        ib, Tb, _, vb, rb = render(cam, b, bg)
    order = b.original_order()
    assert torch.equal(ia, ib) and torch.equal(Ta, Tb) and torch.equal(ra, rb[order]) and torch.equal(va, vb[order])
    # the tie rule has teeth: without it the permuted map lists tied Gaussians in storage order and pixels change
    tie, b._tie = b._tie, None
    with torch.no_grad():
        ic = render(cam, b, bg)[0]
    b._tie = tie
    assert not torch.equal(ia, ic)
    for _ in range(5):
        la, _ = trainer.training_step_fused(a, cam, gt, bg)
        lb, _ = trainer.training_step_fused(b, cam, gt, bg)
    torch.cuda.synchronize()
    assert float(torch.as_tensor(la).sum()) == float(torch.as_tensor(lb).sum())
    _same_map(a, b)
    # ... and through the reference's operator API (autograd node + separate Adam) as well
    trainer.training_step(a, cam, gt, bg); trainer.training_step(b, cam, gt, bg)
    _same_map(a, b)


def test_morton_model_graphed_step_and_extend_and_save_map(tmp_path):
    from gaussian_lic_amd import io_ply, trainer
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import gt_image, lidar_scene
    dev = torch.device("cuda:0")
    W, H = 320, 192
    raw = _scene(30000, W, H, 9)
    u_pix = raw["xyz"][:, 0] * (0.675 * W) / raw["xyz"][:, 2].abs().clamp_min(0.2) + 0.4857 * W
    keep = u_pix < 0.7 * W      # the map does not cover the right 30 % of the image yet: that is where extend() inserts LiDAR points
    raw = {k: (v[keep].contiguous() if torch.is_tensor(v) else v) for k, v in raw.items()}
    P = int(raw["xyz"].shape[0])
    a, b = _models(raw, dev, capacity=2 * P, resort_fraction=None)   # (no automatic re-sort: the appended rows' place is asserted below; resort() is exercised at the end)
    cam = synthetic_camera(W, H).to_device(dev)
    gt, bg = gt_image(H, W).to(dev), torch.zeros(3, device=dev)
    ga, gb = trainer.GraphedStep(a, cam, gt, bg, check_every=0, use_graph=True), trainer.GraphedStep(b, cam, gt, bg, check_every=0, use_graph=True)
    for _ in range(3):
        ga.step(); gb.step()
    assert ga.check() == 0 and gb.check() == 0
    del ga, gb
    _same_map(a, b)
    # extend(): the same LiDAR frame appended to both maps -> the same rows, behind the sorted block, original index = row
    frame = lidar_scene(4000, W, H, sh_degree=3, seed=77)
    pts = frame["xyz"].to(dev)
    col = (frame["features_dc"].reshape(-1, 3) * 0.28209479177387814 + 0.5).to(dev)
    rsp = frame["xyz"][:, 2].contiguous().to(dev)
    Rcw = torch.from_numpy(cam.world_view_transform[:3, :3].T.copy())
    tcw = torch.from_numpy(cam.world_view_transform[3, :3].copy())
    intr = (float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy))
    ka, kb = a.extend(cam, pts, col, rsp, Rcw, tcw, intr), b.extend(cam, pts, col, rsp, Rcw, tcw, intr)
    assert ka == kb and ka > 0 and a.P == b.P == P + ka
    assert torch.equal(b.tie_rank[P:].cpu(), torch.arange(P, P + kb, dtype=torch.int32))
    _same_map(a, b)
    for _ in range(2):
        trainer.training_step_fused(a, cam, gt, bg); trainer.training_step_fused(b, cam, gt, bg)
    _same_map(a, b)
    # saveMap: byte-identical files (gaussian.cpp:306-397 writes rows in the map's order: the permuted model writes its ORIGINAL order)
    pa, pb = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    io_ply.save_map(a, pa); io_ply.save_map(b, pb)
    assert open(pa, "rb").read() == open(pb, "rb").read()
    # resort(): the grown map back in Morton order on the device — parameters, moments and tie_rank permuted together: still the same map, it keeps
    # training bit-identically, and saveMap still writes the original order
    tail_before = b.tie_rank[P:].clone()
    perm = b.resort()
    assert sorted(perm.cpu().tolist()) == list(range(b.P)) and not torch.equal(b.tie_rank[P:], tail_before)
    assert sorted(b.tie_rank.cpu().tolist()) == list(range(b.P))
    _same_map(a, b)
    for _ in range(2):
        trainer.training_step_fused(a, cam, gt, bg); trainer.training_step_fused(b, cam, gt, bg)
    _same_map(a, b)
    io_ply.save_map(b, pb)
    assert open(pa, "rb").read() != open(pb, "rb").read()      # (two more steps: the file moved ...)
    io_ply.save_map(a, pa)
    assert open(pa, "rb").read() == open(pb, "rb").read()      # (... to the same bytes)


def test_extend_resorts_a_morton_map_once_the_appended_tail_is_large():
    """rows appended by extend() sit behind the sorted block, so the layout's coherence decays as a SLAM map grows.  With
    resort_fraction (default 0.1) extend() re-sorts the map on the device once the unsorted tail passes that share: same map bit for bit as the
    insertion-order model through appends, re-sorts and training steps; the sorts' cost is recorded (model.sort_ms)."""
    from gaussian_lic_amd import trainer
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import gt_image, lidar_scene
    dev = torch.device("cuda:0")
    W, H = 320, 192
    raw = _scene(30000, W, H, 11)
    u_pix = raw["xyz"][:, 0] * (0.675 * W) / raw["xyz"][:, 2].abs().clamp_min(0.2) + 0.4857 * W
    keep = u_pix < 0.6 * W
    raw = {k: (v[keep].contiguous() if torch.is_tensor(v) else v) for k, v in raw.items()}
    a, b = _models(raw, dev)
    assert b.resort_fraction == 0.1 and a.resort_fraction is None
    cam = synthetic_camera(W, H).to_device(dev)
    gt, bg = gt_image(H, W).to(dev), torch.zeros(3, device=dev)
    Rcw = torch.from_numpy(cam.world_view_transform[:3, :3].T.copy())
    tcw = torch.from_numpy(cam.world_view_transform[3, :3].copy())
    intr = (float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy))
    sorts = len(b.sort_ms)
    for f in range(4):
        frame = lidar_scene(1500, W, H, sh_degree=3, seed=80 + f)
        pts = frame["xyz"].to(dev)
        col = (frame["features_dc"].reshape(-1, 3) * 0.28209479177387814 + 0.5).to(dev)
        rsp = frame["xyz"][:, 2].contiguous().to(dev)
        assert a.extend(cam, pts, col, rsp, Rcw, tcw, intr) == b.extend(cam, pts, col, rsp, Rcw, tcw, intr)
        for _ in range(2):
            trainer.training_step_fused(a, cam, gt, bg); trainer.training_step_fused(b, cam, gt, bg)
        _same_map(a, b)
    assert len(b.sort_ms) > sorts and b._sorted_P > int(keep.sum())       # (at least one automatic re-sort happened on the way)
    assert (b.P - b._sorted_P) <= 0.1 * b.P
