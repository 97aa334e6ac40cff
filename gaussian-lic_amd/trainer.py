"""Training step of the hot path — mirrors optimize() (src/gaussian.cpp:640-719) and the parts of GaussianModel it
touches (getters :147-175, trainingSetup :399-418), plus the one exchange step the north star adds for N > 1 GPUs:
a single all-reduce of the Gaussian-gradient slab (+ a max-reduce of the visibility mask) per optimiser step.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).  Every rank
holds a full replica of the parameters and Adam state, renders a different camera view, and applies the identical
sparse-Adam update, so replicas stay bit-identical without a broadcast.
"""
import torch

from . import loss as loss_utils
from .optim import SparseGaussianAdam
from .rasterizer import render

# config/fastlivo.yaml:18-22 (position, feature, opacity, scaling, rotation) and lambda_dssim
DEFAULT_LRS = dict(position_lr=1.6e-4, feature_lr=2.5e-3, opacity_lr=5e-2, scaling_lr=5e-3, rotation_lr=1e-3)
LAMBDA_DSSIM = 0.2


class GaussianModel:
    """Parameters + activations of src/gaussian.{h,cpp} that the hot path touches (no map management)."""

    def __init__(self, raw, device, lambda_erank=0.0):
        self.sh_degree = int(raw["sh_degree"])
        self.lambda_erank = float(lambda_erank)
        mk = lambda t: t.to(device).contiguous().requires_grad_(True)
        self.xyz, self.features_dc, self.features_rest = mk(raw["xyz"]), mk(raw["features_dc"]), mk(raw["features_rest"])
        self.opacity, self.scaling, self.rotation = mk(raw["opacity"]), mk(raw["scaling"]), mk(raw["rotation"])

    # gaussian.cpp:147-175
    def get_xyz(self): return self.xyz
    def get_features_dc(self): return self.features_dc
    def get_features_rest(self): return self.features_rest
    def get_opacity(self): return torch.sigmoid(self.opacity)
    def get_scaling(self): return torch.exp(self.scaling)
    def get_rotation(self): return torch.nn.functional.normalize(self.rotation)

    def parameters(self):
        """Group order of trainingSetup (gaussian.cpp:399-418)."""
        return [self.xyz, self.features_dc, self.features_rest, self.opacity, self.scaling, self.rotation]

    def training_setup(self, lrs=None):
        c = dict(DEFAULT_LRS)
        c.update(lrs or {})
        group_lrs = [c["position_lr"], c["feature_lr"], c["feature_lr"] / 20.0, c["opacity_lr"], c["scaling_lr"], c["rotation_lr"]]
        self.optimizer = SparseGaussianAdam(self.parameters(), group_lrs, eps=1e-15)
        return self.optimizer


def _dist_on():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def allreduce_gradients(grads, visible):
    """The per-step exchange: SUM of the concatenated gradient slab [P x (11+3K)] and MAX (= OR) of the visibility
    bytes.  One collective each; returns (list of reduced gradient views, reduced visibility)."""
    flat = torch.cat([g.reshape(-1) for g in grads])
    vis = visible.to(torch.uint8)
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    torch.distributed.all_reduce(vis, op=torch.distributed.ReduceOp.MAX)
    out, off = [], 0
    for g in grads:
        n = g.numel()
        out.append(flat[off:off + n].view_as(g))
        off += n
    return out, vis.bool()


def training_step(model, camera, gt_image, bg, lambda_dssim=LAMBDA_DSSIM, do_step=True):
    """One iteration of optimize()'s loop (gaussian.cpp:674-716): render -> 0.8*L1 + 0.2*(1-SSIM) -> backward ->
    sparse Adam.  Returns (loss tensor, visible mask)."""
    image, _final_T, _pts, visible, _radii = render(camera, model, bg)
    Ll1 = loss_utils.l1_loss(image, gt_image)
    ssim_value = loss_utils.fused_ssim(image.unsqueeze(0), gt_image.unsqueeze(0))
    loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim_value)
    loss.backward()
    if do_step:
        params = model.parameters()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        if _dist_on():
            grads, visible = allreduce_gradients(grads, visible)
        model.optimizer.set_visibility_and_N(visible, model.xyz.size(0))
        model.optimizer.step(grads)
        model.optimizer.zero_grad(True)
    return loss.detach(), visible


def render_fwd_bwd(model, camera, dL_dimage, bg):
    """Bare rendered view forward + backward (the metric's "(fwd+bwd)"), no loss kernels, no optimiser."""
    image, _final_T, _pts, visible, _radii = render(camera, model, bg)
    image.backward(dL_dimage)
    for p in model.parameters():
        p.grad = None
    return visible
