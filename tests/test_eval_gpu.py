"""-m gpu: the evaluation path (evaluateVisualQuality, gaussian.cpp:751-789; psnr / ssim of
loss_utils.h:35-128) on the device against the CPU oracle: per-view PSNR and SSIM of the HIP render vs the oracle's render of the
same scene, for both SSIM formulations (fused-SSIM kernel, LibTorch conv2d)."""
import numpy as np
import pytest
import torch

from conftest import make_scene

pytestmark = pytest.mark.gpu


def test_evaluate_visual_quality_matches_oracle(oracle32):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import loss, trainer
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import activate, gt_image, to_numpy
    P, W, H = 20000, 320, 240
    dev = torch.device("cuda:0")
    raw, _sc, _camd, _cam = make_scene("random", P, W, H, 3, 51)
    model = trainer.GaussianModel(raw, dev)
    sc = to_numpy(activate(raw))
    cams, gts, ref_psnr, ref_ssim = [], [], [], []
    for view in (1, 5):
        cam = synthetic_camera(W, H, view)
        gt = gt_image(H, W, seed=60 + view) * 1.2 - 0.1             # exercises the clamp of gaussian.cpp:759
        f = oracle32.forward(sc, cam.as_dict())
        img = np.clip(f["color"].astype(np.float32), 0.0, 1.0)
        g = np.clip(gt.numpy(), 0.0, 1.0)
        mse = float(((img.astype(np.float64) - g) ** 2).mean())
        ref_psnr.append(10.0 * np.log10(1.0 / mse))
        ref_ssim.append(float(oracle32.ssim_forward(img[None], g[None], train=False)[0].astype(np.float64).mean()))
        cams.append(cam.to_device(dev))
        gts.append(gt.to(dev))
    bg = torch.zeros(3, device=dev)
    for fused in (True, False):
        p, s = loss.evaluate_visual_quality(model, cams, gts, bg, fused=fused)
        assert p.is_cuda and s.is_cuda
        assert abs(float(p) - float(np.mean(ref_psnr))) < 2e-3, (fused, float(p), ref_psnr)      # dB
        assert abs(float(s) - float(np.mean(ref_ssim))) < 2e-5, (fused, float(s), ref_ssim)
