"""-m gpu: SURVEY.md section 8d's training-config gate — "final PSNR-to-GT within 0.05 dB of the oracle run on a down-scaled instance".
The same optimisation (render forward -> 0.8 L1 + 0.2 (1 - SSIM) -> backward -> masked Adam on the six groups, the reference's learning
rates) runs for 20 steps on the HIP kernels and on the CPU oracle (bench.py:_cpu_step — the oracle leg of `cpu_baseline`), from the same
start; compared: the loss of every step, the final image's PSNR to the target, the final parameters."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _psnr(a, b):
    return 10.0 * np.log10(1.0 / float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


@pytest.mark.parametrize("kind,P,W,H,lr_scale", [("random", 30000, 320, 180, 1.0), ("lidar", 30000, 320, 180, 1.0), ("random", 30000, 320, 180, 0.01)])
def test_twenty_training_steps_track_the_oracle(kind, P, W, H, lr_scale):
    sys.path.insert(0, ROOT)
    import bench
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib, loss as loss_utils, rasterizer as rz
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import activate, gt_image, lidar_scene, random_scene, to_numpy
    from oracle.oracle import Oracle
    STEPS = 20
    lrs = [lr_scale * v for v in (1.6e-4, 2.5e-3, 2.5e-3 / 20.0, 5e-2, 5e-3, 1e-3)]   # gaussian.cpp:399-418 with config/fastlivo.yaml (x 0.01: bench.py's rates)
    raw = (random_scene if kind == "random" else lidar_scene)(P, W, H, sh_degree=3, seed=3)
    cam = synthetic_camera(W, H)
    gt = gt_image(H, W, seed=2)
    keys = ("means", "dc", "shs", "opac", "scales", "rots")

    # ---- oracle run (the algorithm of bench.py's cpu_baseline leg: Adam directly on the arrays the renderer reads)
    orc = Oracle(np.float32)
    sc = to_numpy(activate(raw))
    state = {k: (np.array(sc[k], np.float32, copy=True), np.zeros_like(sc[k], np.float32), np.zeros_like(sc[k], np.float32)) for k in keys}   # (copies: .numpy() aliases the torch tensors of `raw`)
    for k in keys:
        sc[k] = state[k][0]
    camd, gtn = cam.as_dict(), gt.numpy()
    o_loss = []
    for _ in range(STEPS):
        f, _g, _dL = bench._cpu_step(orc, sc, camd, gtn, state, lrs)
        m = orc.ssim_forward(f["color"][None], gtn[None])[0]
        o_loss.append(0.8 * float(np.abs(f["color"] - gtn).mean()) + 0.2 * (1.0 - float(m.mean())))
    o_img = orc.forward(sc, camd)["color"]

    # ---- HIP run: forward / loss kernels / backward / one-launch Adam through the C-ABI, same arrays, same order
    dev = torch.device("cuda:0")
    act = activate(raw)
    prm = {k: act[k].to(dev).contiguous().clone() for k in keys}
    mom = {k: (torch.zeros_like(prm[k]), torch.zeros_like(prm[k])) for k in keys}
    cam.to_device(dev)
    gtd = gt.to(dev)
    e, bg = torch.empty(0, device=dev), torch.zeros(3, device=dev)
    fl = loss_utils.FusedLoss(0.2)
    h_loss = []

    def forward():
        return rz.rasterize_gaussians(bg, prm["means"], e, prm["opac"], prm["scales"], prm["rots"], 1.0, e, cam.d_world_view_transform,
                                      cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy), H, W, float(cam.limx_neg), float(cam.limx_pos),
                                      float(cam.limy_neg), float(cam.limy_pos), prm["dc"], prm["shs"], 3, cam.d_camera_center, False, False, False)
    for _ in range(STEPS):
        R, B, image, _fT, radii, geom, binning, img, sample = forward()
        dL, terms = fl.forward_backward(image, gtd)
        h_loss.append(float(fl.value(terms)))
        g = rz.rasterize_gaussians_backward(bg, prm["means"], radii, e, prm["scales"], prm["rots"], 1.0, e, cam.d_world_view_transform,
                                            cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy), float(cam.limx_neg), float(cam.limx_pos),
                                            float(cam.limy_neg), float(cam.limy_pos), dL, prm["dc"], prm["shs"], 3, cam.d_camera_center, geom, R, binning,
                                            img, B, sample, 0.0, False)
        grads = dict(means=g[3], dc=g[5], shs=g[6], opac=g[2], scales=g[7], rots=g[8])
        vis = (radii > 0).contiguous()
        groups = [_lib.AdamGroup(prm[k].data_ptr(), grads[k].contiguous().data_ptr(), mom[k][0].data_ptr(), mom[k][1].data_ptr(), lr, prm[k].numel() // P)
                  for k, lr in zip(keys, lrs)]
        arr = (_lib.AdamGroup * len(groups))(*groups)
        _lib.check(_lib.lib().gslic_adam_update_groups(arr, len(groups), _lib.ptr(vis), 0.9, 0.999, 1e-15, P, _lib.current_stream_ptr()))
    h_img = forward()[2].cpu().numpy()

    # ---- the gate
    assert o_loss[-1] < o_loss[0]                                           # the optimisation does move
    # step by step: the first six steps agree to 1e-4 at both sets of rates; later Adam's normalised update (|step| = lr whatever the gradient's
    # size) turns last-bit differences of near-zero gradients into lr-sized differences of single parameters and the two runs drift apart slowly.
    # The loss kernels carry the reference's bits (csrc/ssim.hip is compiled with -ffp-contract=off; test_vs_reference_kernels_gpu.py holds them
    # to the reference kernels bit for bit), so what separates the runs is the oracle's host-libm exp in the blend, not instruction scheduling.
    np.testing.assert_allclose(h_loss[:6], o_loss[:6], rtol=1e-4)
    np.testing.assert_allclose(h_loss, o_loss, rtol=5e-3)
    p_h, p_o = _psnr(h_img, gtn), _psnr(o_img, gtn)
    assert abs(p_h - p_o) < 0.05, (p_h, p_o)                                 # final PSNR-to-GT within 0.05 dB
    if lr_scale < 1.0:
        # at bench.py's rates the two runs stay together element by element; at the full rates Adam's sign-like steps (opacity moves by
        # 0.05 per step whatever the gradient's size) amplify last-bit differences into locally different trajectories — same loss, same
        # PSNR to the target, different images — which is a property of the optimiser, not of the kernels
        assert _psnr(h_img, o_img) > 60.0
        for k in keys:
            a, b = prm[k].cpu().numpy().reshape(-1), state[k][0].reshape(-1)
            err = np.abs(a - b) / max(float(np.abs(b).max()), 1e-30)
            assert float(np.median(err)) < 1e-5 and float(np.quantile(err, 0.999)) < 2e-3, (k, float(np.median(err)), float(np.quantile(err, 0.999)), float(err.max()))
