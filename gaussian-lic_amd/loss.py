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
