"""SparseGaussianAdam — mirrors src/optim_utils.h:26-142 and adamUpdate (rasterize_points.cu:248-273).

step() applies the visibility-masked, bias-correction-free Adam of adam.cu:26-37 to every parameter group in
ONE kernel launch (gslic_adam_update_groups) and without the per-group grad.clone() of optim_utils.h:130.
"""
import ctypes

import torch

from . import _lib


def adam_update(param, param_grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M):
    """adamUpdate(param, grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M) — in place."""
    for t in (param, param_grad, exp_avg, exp_avg_sq):
        assert t.is_contiguous() and t.dtype == torch.float32
    vis = visible.contiguous()
    assert vis.dtype in (torch.bool, torch.uint8)
    p = _lib.ptr
    _lib.check(_lib.lib().gslic_adam_update(p(param), p(param_grad), p(exp_avg), p(exp_avg_sq), p(vis), float(lr), float(b1),
                                            float(b2), float(eps), int(N), int(M), _lib.current_stream_ptr()))


class SparseGaussianAdam:
    """Six single-tensor parameter groups with their own learning rates (gaussian.cpp:399-418); state keyed by
    group index; `step` counter kept but unused, like the reference (optim_utils.h:135)."""

    def __init__(self, params, lrs, eps=1e-15, betas=(0.9, 0.999)):
        self.params = list(params)
        self.lrs = [float(x) for x in lrs]
        self.eps, self.betas = float(eps), betas
        self.state = [None] * len(self.params)
        self.visibility, self.N = None, 0

    def rebind(self, params, exp_avgs=None, exp_avg_sqs=None):
        """Point the groups at new parameter tensors (after densificationPostfix, gaussian.cpp:426-497, re-keys the state
        map).  Moments given by the caller replace the stored ones (they already hold the old rows + zeros for the new)."""
        self.params = list(params)
        for i in range(len(self.params)):
            if exp_avgs is not None:
                step = self.state[i]["step"] if self.state[i] else 0
                self.state[i] = dict(step=step, exp_avg=exp_avgs[i], exp_avg_sq=exp_avg_sqs[i])

    def set_visibility_and_N(self, visibility, N):
        self.visibility, self.N = visibility, int(N)

    def _ensure_state(self, i):
        if self.state[i] is None:
            p = self.params[i]
            self.state[i] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
        return self.state[i]

    def step(self, grads=None, only=None, rows=None, grad_addr=None):
        """grads: optional list of gradient tensors (defaults to each param's .grad); only: optional subset of group indices
        (the pipelined N > 1 exchange updates a group as soon as its gradients have been reduced).
        rows = (p0, p1): update Gaussians [p0, p1) only — the chunked exchange; grad_addr = {group index: device address of the chunk's
        gradient block, row p0 first} when the chunk's gradients do not live at their rows of `grads`."""
        groups, keep = [], []
        p0, p1 = (0, self.N) if rows is None else (int(rows[0]), int(rows[1]))
        for i, prm in enumerate(self.params):
            if only is not None and i not in only:
                continue
            g = grads[i] if grads is not None else prm.grad
            if (g is None and not (grad_addr and i in grad_addr)) or prm.numel() == 0:   # an empty group (features_rest [P,0,3] at SH degree 0) is a no-op, as in the reference
                continue
            st = self._ensure_state(i)
            M = prm.numel() // self.N
            if grad_addr and i in grad_addr:
                g_ptr = int(grad_addr[i])
            else:
                g = g.contiguous()
                keep.append(g)
                g_ptr = g.data_ptr() + 4 * M * p0
            off = 4 * M * p0
            groups.append(_lib.AdamGroup(prm.data_ptr() + off, g_ptr, st["exp_avg"].data_ptr() + off, st["exp_avg_sq"].data_ptr() + off,
                                         self.lrs[i], M))
            if rows is None or p1 == self.N:
                st["step"] += 1
        if not groups or p1 <= p0:
            return
        arr = (_lib.AdamGroup * len(groups))(*groups)
        vis = self.visibility.contiguous()
        _lib.check(_lib.lib().gslic_adam_update_groups(arr, len(groups), ctypes.c_void_p(vis.data_ptr() + p0), self.betas[0], self.betas[1], self.eps,
                                                       p1 - p0, _lib.current_stream_ptr()))

    def fused_descriptor(self, visible_out=None):
        """gslic_adam_fused for gslic_rasterize_backward_adam: the six groups' parameters and moments, learning rates, betas, eps.
        visible_out: optional uint8 [P] device tensor the backward also fills with `radii > 0` (kept alive by the caller)."""
        d = _lib.AdamFused()
        d.visible_out = None if visible_out is None else visible_out.data_ptr()
        for i, prm in enumerate(self.params):
            st = self._ensure_state(i)
            d.param[i] = prm.data_ptr() if prm.numel() else None
            d.exp_avg[i] = st["exp_avg"].data_ptr() if prm.numel() else None
            d.exp_avg_sq[i] = st["exp_avg_sq"].data_ptr() if prm.numel() else None
            d.lr[i] = self.lrs[i]
        d.b1, d.b2, d.eps = self.betas[0], self.betas[1], self.eps
        return d

    def step_sh_from_rgb(self, means3D, campos_all, rgb_all, degree, n_views=None, view_stride=0, rows=None):
        """Groups 1 and 2 (features_dc, features_rest) of an N > 1 step: their summed gradients are rebuilt from the views' exchanged colour
        gradients and consumed by the masked Adam update in the same kernel (gslic_sh_grad_from_rgb_adam) — bit-identical to rebuilding the
        rows and calling step(only=[1, 2]), without writing and re-reading 192 B per Gaussian.  set_visibility_and_N() first.
        view_stride > 0 (floats): the views' blocks sit view_stride apart inside an all-gathered payload (see rasterizer.sh_grad_from_rgb)."""
        P = means3D.size(0)
        n = rgb_all.size(0) if n_views is None else int(n_views)
        M = self.params[2].size(1) if self.params[2].numel() else 0
        if not view_stride:
            assert rgb_all.is_contiguous() and campos_all.is_contiguous() and tuple(rgb_all.shape) == (n, P, 3)
        d = self.fused_descriptor()
        vis = self.visibility.contiguous()
        means3D = means3D.contiguous()
        vp = ctypes.c_void_p
        if rows is None:
            p0, nrow = 0, P
        else:   # Gaussians [p0, p1) only (p0 a multiple of 64): rgb_all / campos_all then hold THAT chunk's gathered payload, row p0 first
            p0, nrow = int(rows[0]), int(rows[1]) - int(rows[0])
            assert p0 % 64 == 0
            for i, w in ((1, 3), (2, 3 * M)):
                if d.param[i]:
                    d.param[i] += 4 * w * p0; d.exp_avg[i] += 4 * w * p0; d.exp_avg_sq[i] += 4 * w * p0
        if nrow > 0:
            _lib.check(_lib.lib().gslic_sh_grad_from_rgb_adam(nrow, int(degree), M, n, vp(means3D.data_ptr() + 12 * p0), _lib.ptr(campos_all), _lib.ptr(rgb_all), 0,
                                                              vp(vis.data_ptr() + p0), ctypes.byref(d), None, None, int(view_stride), _lib.current_stream_ptr()))
        if rows is None or int(rows[1]) == P:
            for i in (1, 2):
                if self.state[i] is not None:
                    self.state[i]["step"] += 1

    def step_all_from_exchange(self, means3D, campos_all, rgb_all, degree, n_views, view_stride, vis_all, vis_stride, vis_out, small_grads):
        """The whole optimiser step of an N > 1 rank in one launch (gslic_sh_grad_from_rgb_adam_all): the views' masks are OR-ed in the kernel
        (vis_all: view 0's mask inside the all-gathered payload, vis_stride bytes to the next view's; the OR lands in vis_out [P] uint8),
        dL_ddc / dL_dsh are rebuilt from the gathered colour gradients and consumed by the masked Adam of features_dc / features_rest, and
        the four small groups are updated from `small_grads` = (dL_dxyz, dL_dopacity, dL_dscaling, dL_drotation), the all-reduced slab views."""
        P = means3D.size(0)
        M = self.params[2].size(1) if self.params[2].numel() else 0
        d = self.fused_descriptor()
        gx, go, gs, gr = small_grads
        _lib.check(_lib.lib().gslic_sh_grad_from_rgb_adam_all(P, int(degree), M, int(n_views), _lib.ptr(means3D.contiguous()), _lib.ptr(campos_all), _lib.ptr(rgb_all), 0,
                                                              _lib.ptr(vis_all), int(vis_stride), _lib.ptr(vis_out), ctypes.byref(d), _lib.ptr(gx), _lib.ptr(go),
                                                              _lib.ptr(gs), _lib.ptr(gr), int(view_stride), _lib.current_stream_ptr()))
        self.count_step()

    def count_step(self):
        for st in self.state:
            if st is not None:
                st["step"] += 1

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
