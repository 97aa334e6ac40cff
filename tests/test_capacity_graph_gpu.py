"""-m gpu: the capacity-mode forward (gslic_rasterize_forward_capacity: no host round trip, caller-owned buffers) and the hipGraph
training step built on it.  The callback forward (the reference's contract, rasterize_points.cu:40-48 + rasterizer_impl.cu:398,442)
is the yardstick: same kernels, so everything must be bit-identical."""
import numpy as np
import pytest
import torch

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _inputs(raw, cam, dev):
    from gaussian_lic_amd.synthetic import activate
    act = {k: (v.to(dev).contiguous() if torch.is_tensor(v) else v) for k, v in activate(raw).items()}
    vm = torch.from_numpy(cam.world_view_transform).to(dev)
    pm = torch.from_numpy(cam.full_proj_transform).to(dev)
    cp = torch.from_numpy(cam.camera_center).to(dev)
    scal = (float(cam.tanfovx), float(cam.tanfovy), float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos))
    return act, vm, pm, cp, scal


@pytest.mark.parametrize("kind,P,W,H,deg,seed", [("random", 30000, 320, 240, 3, 3), ("lidar", 20000, 200, 150, 3, 4), ("random", 2000, 70, 50, 0, 5)])
def test_capacity_forward_backward_equal_callback_path(kind, P, W, H, deg, seed):
    from gpu_helpers import hip_backward, hip_forward
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd.synthetic import pixel_grad
    dev = torch.device("cuda:0")
    raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed)
    ref = hip_forward(raw, cam)
    dL = pixel_grad(H, W, seed=1)
    gref = hip_backward(ref, dL)
    act, vm, pm, cp, scal = _inputs(raw, cam, dev)
    bg, e = torch.zeros(3, device=dev), torch.empty(0, device=dev)
    for slack in (1.0, 1.7):   # exactly full, and with room to spare (launches sized for more than there is)
        bufs = rz.CapacityBuffers(P, W, H, int(ref["R"] * slack) + (0 if slack == 1.0 else 999), int(ref["B"] * slack) + (0 if slack == 1.0 else 7), dev)
        out = rz.rasterize_gaussians_capacity(bufs, bg, act["means"], act["opac"], act["scales"], act["rots"], 1.0, vm, pm, *scal, act["dc"],
                                              act["shs"], act["D"], cp)
        cR, cB, color, final_T, radii, geom, binning, img, sample = out
        R, B, bits, good = bufs.read_status()
        assert (R, B, bits, good) == (ref["R"], ref["B"], 0, 1)
        assert cR >= ref["R"] and cB >= ref["B"]
        assert torch.equal(color, ref["color"]) and torch.equal(final_T, ref["final_T"]) and torch.equal(radii, ref["radii"])
        g = rz.rasterize_gaussians_backward(bg, act["means"], radii, e, act["scales"], act["rots"], 1.0, e, vm, pm, scal[0], scal[1], *scal[2:],
                                            dL.to(dev), act["dc"], act["shs"], act["D"], cp, geom, cR, binning, img, cB, sample, 0.0, False)
        names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale", "dL_drot"]
        for n, t in zip(names, g):
            np.testing.assert_array_equal(t.cpu().numpy(), gref[n], err_msg=f"{n} slack {slack}")


def test_capacity_overflow_turns_the_step_into_a_noop():
    from gpu_helpers import hip_forward
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd.synthetic import pixel_grad
    dev = torch.device("cuda:0")
    P, W, H = 20000, 320, 240
    raw, sc, camd, cam = make_scene("random", P, W, H, 3, 6)
    ref = hip_forward(raw, cam)
    act, vm, pm, cp, scal = _inputs(raw, cam, dev)
    bg, e = torch.zeros(3, device=dev), torch.empty(0, device=dev)
    dL = pixel_grad(H, W, seed=1).to(dev)
    for cap_R, cap_B, bit in ((ref["R"] // 2, ref["B"], 1), (ref["R"] + 10, ref["B"] // 2, 2)):
        bufs = rz.CapacityBuffers(P, W, H, cap_R, cap_B, dev)
        cR, cB, color, final_T, radii, geom, binning, img, sample = rz.rasterize_gaussians_capacity(
            bufs, bg, act["means"], act["opac"], act["scales"], act["rots"], 1.0, vm, pm, *scal, act["dc"], act["shs"], act["D"], cp)
        R, B, bits, good = bufs.read_status()
        assert R == ref["R"] and (bits & bit) and good == 0          # the real instance count is still reported: the host sizes the retry from it
        out = {k: torch.full(s, 7.0, device=dev) for k, s in (("xyz", (P, 3)), ("features_dc", (P, 1, 3)), ("features_rest", (P, 15, 3)),
                                                               ("opacity", (P, 1)), ("scaling", (P, 3)), ("rotation", (P, 4)))}
        rz.rasterize_gaussians_backward(bg, act["means"], radii, e, act["scales"], act["rots"], 1.0, e, vm, pm, scal[0], scal[1], *scal[2:], dL,
                                        act["dc"], act["shs"], act["D"], cp, geom, cR, binning, img, cB, sample, 0.0, False, out=out)
        torch.cuda.synchronize()
        for k, t in out.items():
            assert bool((t == 7.0).all()), f"{k} was written although the forward had overflowed"


@pytest.mark.parametrize("use_graph", [True, False])
def test_graphed_step_equals_eager_steps(use_graph):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import trainer
    from gaussian_lic_amd.synthetic import gt_image
    dev = torch.device("cuda:0")
    P, W, H, steps = 40000, 320, 240, 12
    raw, sc, camd, cam = make_scene("random", P, W, H, 3, 8)
    cam.to_device(dev)
    gt = gt_image(H, W, seed=3).to(dev)
    bg = torch.zeros(3, device=dev)

    def fresh():
        m = trainer.GaussianModel(raw, dev)
        m.training_setup()
        return m
    eager = fresh()
    for _ in range(steps):
        terms_e, _vis = trainer.training_step_fused(eager, cam, gt, bg)
    for headroom, expect_recapture in ((1.25, False), (0.9, True)):   # 0.9: the first capture is too small on purpose
        model = fresh()
        gs = trainer.GraphedStep(model, cam, gt, bg, headroom=headroom, check_every=5, use_graph=use_graph)
        if expect_recapture:
            gs.cap_R, gs.cap_B = int(gs.cap_R * 0.5), int(gs.cap_B * 0.5)
            gs._capture()
        for _ in range(steps):
            terms_g = gs.step()
        gs.check()
        assert (gs.recaptures > 0) == expect_recapture
        for n in model.NAMES:
            assert torch.equal(getattr(model, n).detach(), getattr(eager, n).detach()), (n, headroom)
            assert torch.equal(model._m[n][:P], eager._m[n][:P]) and torch.equal(model._v[n][:P], eager._v[n][:P]), (n, headroom)
        assert torch.equal(terms_g, terms_e)


@pytest.mark.parametrize("use_graph", [True, False])
def test_graphed_step_repeats_overflowed_steps_with_their_own_views(use_graph):
    """A different camera and target every step, capacity buffers sized so that SOME views overflow: every view must be trained exactly
    once — the views that fitted in issue order, then the ones that did not (each with ITS camera and ground truth, not the last one's)
    — i.e. the parameters equal an eager run over that order bit for bit.  Also: the status words report the did-not-fit steps."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd import trainer
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import gt_image
    dev = torch.device("cuda:0")
    P, W, H = 40000, 320, 240
    raw, sc, camd, cam0 = make_scene("random", P, W, H, 3, 8)
    cams = [synthetic_camera(W, H, k).to_device(dev) for k in range(8)]
    gts = [gt_image(H, W, seed=10 + k).to(dev) for k in range(8)]
    bg = torch.zeros(3, device=dev)

    def fresh():
        m = trainer.GaussianModel(raw, dev)
        m.training_setup()
        return m
    # instance counts of the eight views on the initial map: pick a capacity between the smallest and the largest
    probe = fresh()
    e = torch.empty(0, device=dev)
    Rs = []
    with torch.no_grad():
        for c in cams:
            Rs.append(rz.rasterize_gaussians(bg, probe.xyz.detach(), e, probe.opacity.detach(), probe.scaling.detach(), probe.rotation.detach(), 1.0, e,
                                             c.d_world_view_transform, c.d_full_proj_transform, float(c.tanfovx), float(c.tanfovy), H, W,
                                             float(c.limx_neg), float(c.limx_pos), float(c.limy_neg), float(c.limy_pos), probe.features_dc.detach(),
                                             probe.features_rest.detach(), 3, c.d_camera_center, False, False, False, raw_params=True)[0])
    assert max(Rs) > min(Rs) + 64
    cap_R = (max(Rs) + sorted(Rs)[len(Rs) // 2]) // 2      # the larger views do not fit
    fits = [r <= cap_R - 64 for r in Rs]
    assert any(fits) and not all(fits)
    model = fresh()
    first_fit = fits.index(True)
    gs = trainer.GraphedStep(model, cams[first_fit], gts[first_fit], bg, check_every=0, cap_R=cap_R, use_graph=use_graph)
    order = list(range(8))
    for k in order:
        gs.step(cams[k], gts[k])
    issued, mask, max_R, _max_B = gs.bufs.read_window()
    assert issued == 8 and max_R >= max(Rs) - 4096
    failed = [k for k in order if (mask >> k) & 1]
    assert failed and all(not fits[k] or Rs[k] > cap_R - 4096 for k in failed)   # (counts drift a little as the map trains)
    repeated = gs.check()
    assert repeated == len(failed) and gs.recaptures >= 1
    eager = fresh()
    for k in [k for k in order if k not in failed] + failed:
        trainer.training_step_fused(eager, cams[k], gts[k], bg)
    for n in model.NAMES:
        assert torch.equal(getattr(model, n).detach(), getattr(eager, n).detach()), n
        assert torch.equal(model._m[n][:P], eager._m[n][:P]) and torch.equal(model._v[n][:P], eager._v[n][:P]), n


def test_graphed_step_repeats_with_the_pose_of_the_step_when_one_camera_object_moves():
    """The host keeps ONE Camera object and moves it in place between steps (pose refinement, a reused staging object): a step that has to
    be repeated must run on the pose it was ISSUED with (snapshotted by value), not on the pose the object holds at check time; and a target
    tensor that was overwritten in place after its step makes the repeat fail loudly instead of training on the wrong data."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd import trainer
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import gt_image
    dev = torch.device("cuda:0")
    P, W, H = 40000, 320, 240
    raw, sc, camd, cam0 = make_scene("random", P, W, H, 3, 8)
    poses = [synthetic_camera(W, H, k) for k in range(6)]
    gts = [gt_image(H, W, seed=20 + k).to(dev) for k in range(6)]
    bg = torch.zeros(3, device=dev)

    def fresh():
        m = trainer.GaussianModel(raw, dev)
        m.training_setup()
        return m
    probe, e, Rs = fresh(), torch.empty(0, device=dev), []
    with torch.no_grad():
        for c in poses:
            c.to_device(dev)
            Rs.append(rz.rasterize_gaussians(bg, probe.xyz.detach(), e, probe.opacity.detach(), probe.scaling.detach(), probe.rotation.detach(), 1.0, e,
                                             c.d_world_view_transform, c.d_full_proj_transform, float(c.tanfovx), float(c.tanfovy), H, W,
                                             float(c.limx_neg), float(c.limx_pos), float(c.limy_neg), float(c.limy_pos), probe.features_dc.detach(),
                                             probe.features_rest.detach(), 3, c.d_camera_center, False, False, False, raw_params=True)[0])
    cap_R = (max(Rs) + sorted(Rs)[len(Rs) // 2]) // 2
    fits = [r <= cap_R - 64 for r in Rs]
    assert any(fits) and not all(fits)
    moving = synthetic_camera(W, H, fits.index(True)).to_device(dev)       # the one object the host moves around
    model = fresh()
    gs = trainer.GraphedStep(model, moving, gts[fits.index(True)], bg, check_every=0, cap_R=cap_R, use_graph=True)
    for k in range(6):
        moving.set_pose(poses[k].R_wc, poses[k].t_wc)
        moving.to_device(dev)
        gs.step(moving, gts[k])
    _issued, mask, _mr, _mb = gs.bufs.read_window()
    failed = [k for k in range(6) if (mask >> k) & 1]
    assert failed
    moving.set_pose(poses[0].R_wc, poses[0].t_wc); moving.to_device(dev)    # wherever the object points now must not matter
    assert gs.check() == len(failed)
    eager = fresh()
    for k in [k for k in range(6) if k not in failed] + failed:
        trainer.training_step_fused(eager, poses[k], gts[k], bg)
    for n in model.NAMES:
        assert torch.equal(getattr(model, n).detach(), getattr(eager, n).detach()), n
    # a target modified in place after its step was issued: the repeat refuses
    model2 = fresh()
    gs2 = trainer.GraphedStep(model2, poses[fits.index(True)], gts[fits.index(True)], bg, check_every=0, cap_R=cap_R, use_graph=True)
    k_bad = fits.index(False)
    staging = gts[k_bad].clone()
    gs2.step(poses[k_bad], staging)
    staging.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        gs2.check()
