"""Loss operators — mirrors src/loss_utils.h (l1_loss :30-33, FusedSSIMMap :130-193, psnr :35-39) on top of
gslic_fusedssim_forward/_backward (replacing src/fused-ssim/ssim.cu)."""
import torch

from . import _lib

C1 = 0.01 ** 2
C2 = 0.03 ** 2


def fusedssim(C1_, C2_, img1, img2, train):
    """fusedssim(C1, C2, img1, img2, train) -> (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)  (ssim.h:7-15)."""
    img1, img2 = img1.contiguous(), img2.contiguous()
    B, CH, H, W = img1.shape
    m = torch.empty_like(img1)
    if train:
        d = [torch.empty_like(img1) for _ in range(3)]
    else:
        d = [torch.empty(0, device=img1.device) for _ in range(3)]
    p = _lib.ptr
    _lib.check(_lib.lib().gslic_fusedssim_forward(B, CH, H, W, float(C1_), float(C2_), p(img1), p(img2), p(m), p(d[0]), p(d[1]),
                                                  p(d[2]), _lib.current_stream_ptr()))
    return m, d[0], d[1], d[2]


def fusedssim_backward(C1_, C2_, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    """fusedssim_backward(...) -> dL_dimg1  (ssim.h:17-26)."""
    img1, img2, dL_dmap = img1.contiguous(), img2.contiguous(), dL_dmap.contiguous()
    B, CH, H, W = img1.shape
    out = torch.empty_like(img1)
    p = _lib.ptr
    _lib.check(_lib.lib().gslic_fusedssim_backward(B, CH, H, W, float(C1_), float(C2_), p(img1), p(img2), p(dL_dmap), p(dm_dmu1),
                                                   p(dm_dsigma1_sq), p(dm_dsigma12), p(out), _lib.current_stream_ptr()))
    return out


class FusedSSIMMap(torch.autograd.Function):
    """loss_utils.h:130-187 with padding == "same"."""

    @staticmethod
    def forward(ctx, C1_, C2_, img1, img2):
        m, d1, d2, d3 = fusedssim(C1_, C2_, img1, img2, True)
        ctx.save_for_backward(img1.detach(), img2, d1, d2, d3)
        ctx.C = (C1_, C2_)
        return m

    @staticmethod
    def backward(ctx, dL_dmap):
        img1, img2, d1, d2, d3 = ctx.saved_tensors
        grad = fusedssim_backward(ctx.C[0], ctx.C[1], img1, img2, dL_dmap, d1, d2, d3)
        return None, None, grad, None


def fused_ssim(img1, img2):
    """loss_utils.h:189-193."""
    return FusedSSIMMap.apply(C1, C2, img1, img2).mean()


def l1_loss(network_output, gt):
    """loss_utils.h:30-33."""
    return torch.abs(network_output - gt).mean()


def psnr(img1, img2):
    """loss_utils.h:35-39."""
    mse = torch.pow(img1 - img2, 2).mean()
    return 10.0 * torch.log10(1.0 / mse)


def _gaussian_window(window_size, sigma, channel, like):
    """gaussian() + create_window() of loss_utils.h:41-74 (float math like the reference)."""
    import math
    import numpy as np
    # std::exp(-temp * temp / (2.0f * sigma * sigma)) with a float argument (loss_utils.h:50-51): the quotient is rounded to float BEFORE the exponential
    # (expf: correctly rounded here, like the double exp of the float argument rounded once); tests/golden/eval_*.npz hold the reference's window
    den = np.float32(2.0) * np.float32(sigma) * np.float32(sigma)
    g = torch.tensor([float(np.float32(math.exp(float(np.float32(-((x - window_size // 2) ** 2)) / den)))) for x in range(window_size)], dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous().to(like.device).type_as(like)


def ssim(img1, img2, window_size=11, size_average=True):
    """Evaluation SSIM — loss_utils.h:80-128 (conv2d with an 11x11 Gaussian window, zero padding); used by
    evaluateVisualQuality (gaussian.cpp:721-831).  LibTorch conv2d stays the host's op, exactly as in the reference."""
    channel = img1.size(-3)
    window = _gaussian_window(window_size, 1.5, channel, img1)
    pad = window_size // 2
    conv = lambda x: torch.nn.functional.conv2d(x, window, padding=pad, groups=channel)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = conv(img1 * img1) - mu1_sq
    sigma2_sq = conv(img2 * img2) - mu2_sq
    sigma12 = conv(img1 * img2) - mu1_mu2
    c1, c2 = 0.01 * 0.01, 0.03 * 0.03
    ssim_map = ((2 * mu1_mu2 + c1) * (2 * sigma12 + c2)) / ((mu1_sq + mu2_sq + c1) * (sigma1_sq + sigma2_sq + c2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def evaluate_visual_quality(model, cameras, gt_images, bg, fused=True):
    """PSNR / SSIM over a set of views — the metric part of evaluateVisualQuality (gaussian.cpp:751-789): render, clamp to [0,1],
    psnr + SSIM per image, averaged over the views.  fused=True evaluates the SSIM map with the fused-SSIM kernel
    (gslic_fusedssim_forward, train = 0: the same 11-tap sigma-1.5 window with zero padding as loss_utils.h:41-128, separable, one
    launch) instead of five LibTorch conv2d calls; fused=False is the reference's formulation verbatim.  Returns device scalars
    (mean PSNR, mean SSIM).  (LPIPS needs the TorchScript AlexNet blob the reference does not ship.)"""
    from .rasterizer import render
    ps, ss = [], []
    with torch.no_grad():
        for cam, gt in zip(cameras, gt_images):
            img = torch.clamp(render(cam, model, bg)[0], 0.0, 1.0)
            gt = torch.clamp(gt, 0.0, 1.0)                      # gaussian.cpp:759
            ps.append(psnr(img, gt))
            if fused:
                ss.append(fusedssim(C1, C2, img.unsqueeze(0).contiguous(), gt.unsqueeze(0).contiguous(), False)[0].mean())
            else:
                ss.append(ssim(img.unsqueeze(0), gt.unsqueeze(0)))
    return torch.stack(ps).mean(), torch.stack(ss).mean()


class _L1SsimLoss(torch.autograd.Function):
    """(1 - lambda) * l1_loss + lambda * (1 - fused_ssim) as ONE autograd node on gslic_l1_ssim_loss_forward / _backward: what
    gaussian.cpp:685-691 builds from l1_loss (loss_utils.h:30-33: sub, abs, mean), fused_ssim (:130-193: the map kernel + mean) and four
    scalar ops, with their backward nodes — two kernels forward, one backward.  dL/dimage equals the chain's bit for bit
    (tests/test_vs_reference_kernels_gpu.py::test_fused_loss_gradient_is_the_reference_chain_bit_for_bit) times the upstream scalar."""

    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        fl = FusedLoss(lambda_dssim)
        ctx.fl, ctx.shape = fl, image.shape
        img3, gt3 = image.reshape(image.shape[-3:]), gt.reshape(gt.shape[-3:])
        dL, terms = fl.forward_backward(img3, gt3)
        ctx.save_for_backward(dL)
        return fl.value(terms)

    @staticmethod
    def backward(ctx, g):
        (dL,) = ctx.saved_tensors
        return (dL * g).reshape(ctx.shape), None, None


def l1_ssim_loss(image, gt, lambda_dssim=0.2):
    """Optional one-call replacement for the loss lines of optimize() (gaussian.cpp:685-691): `loss = l1_ssim_loss(rendered_image, gt_image,
    lambda_dssim)`.  image, gt: [3,H,W] (or [1,3,H,W])."""
    return _L1SsimLoss.apply(image, gt, float(lambda_dssim))


class FusedLoss:
    """loss = (1-lambda) * L1 + lambda * (1 - SSIM) of optimize() (gaussian.cpp:685-691) computed by two kernels
    (gslic_l1_ssim_loss_forward / _backward) with no LibTorch elementwise ops and no autograd graph."""

    def __init__(self, lambda_dssim=0.2):
        self.lambda_dssim = float(lambda_dssim)
        self._shape, self._buf = None, None

    def _scratch(self, img):
        if self._shape != tuple(img.shape) or self._buf[0].device != img.device:
            n = int(_lib.lib().gslic_loss_partials_count(1, img.shape[0], img.shape[1], img.shape[2]))
            self._buf = [torch.empty_like(img) for _ in range(4)] + [torch.empty(n, device=img.device), torch.empty(2, device=img.device)]
            self._shape = tuple(img.shape)
        return self._buf

    def forward_backward(self, image, gt):
        """image, gt: [3,H,W].  Returns (dL/dimage, terms) with terms = device tensor [mean|img-gt|, mean ssim]."""
        image, gt = image.contiguous(), gt.contiguous()
        CH, H, W = image.shape
        d1, d2, d3, dL, partials, terms = self._scratch(image)
        p, L = _lib.ptr, _lib.lib()
        # (two launches: the reduction of the forward's partial sums rides on the backward kernel; gslic_l1_ssim_loss_forward + _backward are three)
        _lib.check(L.gslic_l1_ssim_loss_forward_backward(1, CH, H, W, C1, C2, self.lambda_dssim, p(image), p(gt), p(d1), p(d2), p(d3), p(partials),
                                                         p(terms), p(dL), _lib.current_stream_ptr()))
        return dL, terms

    def value(self, terms):
        return (1.0 - self.lambda_dssim) * terms[0] + self.lambda_dssim * (1.0 - terms[1])
