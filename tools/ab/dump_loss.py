"""Dump the loss kernels' outputs for a fixed input (same-box A/B of two library builds: GSLIC_HIP_LIB selects the build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gaussian_lic_amd
from gaussian_lic_amd import loss as loss_utils
out = sys.argv[1]
res = {}
for (H, W) in ((180, 320), (1080, 1920), (97, 333)):
    g = torch.Generator().manual_seed(H)
    a = torch.rand(3, H, W, generator=g).cuda(); b = torch.rand(3, H, W, generator=g).cuda()
    fl = loss_utils.FusedLoss(0.2)
    dL, terms = fl.forward_backward(a, b)
    res[f"dL_{H}"] = dL.cpu().numpy(); res[f"terms_{H}"] = terms.cpu().numpy()
    m = loss_utils.fused_ssim(a[None], b[None]) if hasattr(loss_utils, "fused_ssim") else None
np.savez(out, **res)
print("saved", out)
