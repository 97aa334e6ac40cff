"""Host-side mirror of the reference's rasterizer operator interface on top of the C-ABI.

Same names and argument meaning as the reference's L2/L3 layers:
  rasterize_gaussians / rasterize_gaussians_backward  <- RasterizeGaussiansCUDA / ...BackwardCUDA
                                                         (src/rasterizer/rasterize_points.h:25-82)
  GaussianRasterizationSettings, GaussianRasterizerFunction, GaussianRasterizer
                                                      <- src/rasterizer/rasterizer.h:27-114, rasterizer.cpp:21-216
  render                                              <- src/rasterizer/renderer.cpp:21-88
torch is used for device memory, streams and autograd plumbing only; all arithmetic runs in libgslic_hip.so.
"""
import ctypes
from dataclasses import dataclass

import torch

from . import _lib


@dataclass
class GaussianRasterizationSettings:
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    limx_neg: float
    limx_pos: float
    limy_neg: float
    limy_pos: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor   # [4,4] world_view_transform_ (stored transposed, camera.h:86)
    projmatrix: torch.Tensor   # [4,4] full_proj_transform_
    sh_degree: int
    campos: torch.Tensor       # [3]
    prefiltered: bool = False
    debug: bool = False
    no_color: bool = False
    lambda_erank: float = 0.0
    tie_rank: torch.Tensor = None   # not in the reference: int32 [P], set by render() for a model that keeps its rows in a permuted order (_params)


def _params(P, D, M, H, W, tanfovx, tanfovy, lxn, lxp, lyn, lyp, scale_modifier, prefiltered, debug, no_color, raw=False, tie_rank=None):
    """tie_rank (forward only): device int32 [P], the rows' indices in the map's ORIGINAL order when the host stores them permuted
    (gslic_raster_params.tie_rank); the caller keeps the tensor alive for the duration of the call."""
    if tie_rank is not None:
        assert tie_rank.is_contiguous() and tie_rank.dtype == torch.int32 and tie_rank.numel() >= int(P)
    return _lib.RasterParams(int(P), int(D), int(M), int(W), int(H), float(tanfovx), float(tanfovy), float(lxn), float(lxp),
                             float(lyn), float(lyp), float(scale_modifier), int(bool(prefiltered)), int(bool(debug)),
                             int(bool(no_color)), int(bool(raw)), None if tie_rank is None else ctypes.c_void_p(tie_rank.data_ptr()))


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                        projmatrix, tan_fovx, tan_fovy, image_height, image_width, limx_neg, limx_pos, limy_neg, limy_pos,
                        dc, sh, degree, campos, prefiltered, debug, no_color=False, raw_params=False, tie_rank=None):
    """RasterizeGaussiansCUDA (rasterize_points.cu:50-149): returns
    (num_rendered, num_buckets, out_color, out_final_T, radii, geomBuffer, binningBuffer, imgBuffer, sampleBuffer).
    raw_params=True (not in the reference): opacity / scales / rotations are the RAW parameters, activated inside the kernels.
    tie_rank (not in the reference): see _params — a map stored in a permuted row order renders as the unpermuted one."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:77-80
    L = _lib.lib()
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    M = sh.size(1) if sh is not None and sh.size(0) != 0 else 0
    out_color = torch.zeros(3, H, W, dtype=torch.float32, device=dev) if (P == 0 or no_color) else \
        torch.empty(3, H, W, dtype=torch.float32, device=dev)
    out_final_T = torch.zeros(H, W, dtype=torch.float32, device=dev) if P == 0 else torch.empty(H, W, dtype=torch.float32, device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    allocs = [_lib.TensorAllocator(dev) for _ in range(4)]  # geom, binning, img, sample
    R, B = ctypes.c_int32(0), ctypes.c_int32(0)
    if P != 0:
        means3D, dc, opacity, scales, rotations = map(_f32c, (means3D, dc, opacity, scales, rotations))
        sh_c = _f32c(sh) if M > 0 else None
        viewmatrix, projmatrix, campos, background = map(_f32c, (viewmatrix, projmatrix, campos, background))
        prm = _params(P, degree, M, H, W, tan_fovx, tan_fovy, limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier,
                      prefiltered, debug, no_color, raw_params, tie_rank)
        p = _lib.ptr
        _lib.check(L.gslic_rasterize_forward(
            ctypes.byref(prm), allocs[0].cb, None, allocs[1].cb, None, allocs[2].cb, None, allocs[3].cb, None,
            p(background), p(means3D), p(dc), p(sh_c), p(colors), p(opacity), p(scales), p(rotations), p(cov3D_precomp),
            p(viewmatrix), p(projmatrix), p(campos), p(out_color), p(out_final_T), p(radii),
            ctypes.byref(R), ctypes.byref(B), _lib.current_stream_ptr()))
    return (R.value, B.value, out_color, out_final_T, radii, allocs[0].tensor, allocs[1].tensor, allocs[2].tensor, allocs[3].tensor)


class CapacityBuffers:
    """Caller-owned scratch + outputs of gslic_rasterize_forward_capacity for one (P, W, H): nothing is allocated per step, every
    address is stable, so the step can be captured in a hipGraph.  cap_R / cap_B: how many instances / checkpoint buckets fit."""

    def __init__(self, P, W, H, cap_R, cap_B, device, no_color=False):
        L = _lib.lib()
        self.P, self.W, self.H, self.no_color = int(P), int(W), int(H), bool(no_color)
        mk = lambda n: torch.empty(int(n), dtype=torch.uint8, device=device)
        self.geom = mk(L.gslic_geom_bytes(self.P))
        self.img = mk(L.gslic_img_bytes(self.W, self.H))
        self.binning = mk(L.gslic_binning_bytes(int(cap_R), int(no_color)))
        self.sample = mk(L.gslic_sample_bytes(int(cap_B)) if not no_color else 0)
        self.status = torch.zeros(8, dtype=torch.int32, device=device)
        self.color = torch.zeros(3, self.H, self.W, dtype=torch.float32, device=device)
        self.final_T = torch.empty(self.H, self.W, dtype=torch.float32, device=device)
        self.radii = torch.empty(self.P, dtype=torch.int32, device=device)
        self.cap_R = self.cap_B = 0   # filled by the first forward (what the library derives from the buffer sizes)

    def read_status(self):
        """(R, B, bits, forwards that completed) of the status words — synchronises."""
        r, b, bits, good = (int(v) for v in self.status.cpu().tolist()[:4])
        return r & 0xffffffff, b & 0xffffffff, bits, good & 0xffffffff

    def read_window(self):
        """(forwards issued, bit mask of the issue indices mod 32 that did not complete, largest R, largest B) since the words were zeroed
        — synchronises."""
        w = [int(v) & 0xffffffff for v in self.status.cpu().tolist()]
        return w[4], w[5], w[6], w[7]


def rasterize_gaussians_capacity(bufs, background, means3D, opacity, scales, rotations, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                 limx_neg, limx_pos, limy_neg, limy_pos, dc, sh, degree, campos, raw_params=False, tie_rank=None):
    """gslic_rasterize_forward_capacity: RasterizeGaussiansCUDA without a host round trip, into caller-owned buffers.  The tensors must
    already be contiguous fp32 (no copies are made: addresses have to be stable for graph capture).  Returns the same 9-tuple as
    rasterize_gaussians with (cap_R, cap_B) in place of (R, B) — pass them on to rasterize_gaussians_backward unchanged."""
    L = _lib.lib()
    P = means3D.size(0)
    assert P == bufs.P and P > 0
    M = sh.size(1) if sh is not None and sh.size(0) != 0 else 0
    for t in (means3D, dc, opacity, scales, rotations, viewmatrix, projmatrix, campos):
        assert t.is_contiguous() and t.dtype == torch.float32
    prm = _params(P, degree, M, bufs.H, bufs.W, tan_fovx, tan_fovy, limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier, False, False,
                  bufs.no_color, raw_params, tie_rank)
    p = _lib.ptr
    cR, cB = ctypes.c_int32(0), ctypes.c_int32(0)
    bp = lambda t: ctypes.c_void_p(t.data_ptr()) if t.numel() else None
    _lib.check(L.gslic_rasterize_forward_capacity(
        ctypes.byref(prm), bp(bufs.geom), bufs.geom.numel(), bp(bufs.binning), bufs.binning.numel(), bp(bufs.img), bufs.img.numel(),
        bp(bufs.sample), bufs.sample.numel(), p(background), p(means3D), p(dc), p(sh if M > 0 else None), None, p(opacity), p(scales),
        p(rotations), None, p(viewmatrix), p(projmatrix), p(campos), p(bufs.color), p(bufs.final_T), p(bufs.radii), ctypes.byref(cR),
        ctypes.byref(cB), p(bufs.status), _lib.current_stream_ptr()))
    bufs.cap_R, bufs.cap_B = cR.value, cB.value
    return (cR.value, cB.value, bufs.color, bufs.final_T, bufs.radii, bufs.geom, bufs.binning, bufs.img, bufs.sample)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                 projmatrix, tan_fovx, tan_fovy, limx_neg, limx_pos, limy_neg, limy_pos, dL_dout_color, dc, sh,
                                 degree, campos, geomBuffer, R, binningBuffer, imageBuffer, B, sampleBuffer, lambda_erank, debug,
                                 raw_params=False, out=None, adam=None, camera_grads=False, rgb_out=None, rows=None, skip_blend=False,
                                 out_addr=None, payload=None):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:151-246): returns (dL_dmeans2D, dL_dcolors_precomp,
    dL_dopacities, dL_dmeans3D, dL_dcov3Ds_precomp, dL_ddc, dL_dsh, dL_dscales, dL_drotations).
    raw_params=True: scales / rotations are raw and dL_dopacities / dL_dscales / dL_drotations are w.r.t. the raw parameters.
    camera_grads=True (gslic_rasterize_backward_camera; no reference counterpart): three more tensors are appended —
    dL_dviewmatrix [16], dL_dprojmatrix [16], dL_dcampos [3], element order of the inputs."""
    L = _lib.lib()
    dev = means3D.device
    P, H, W = means3D.size(0), dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh is not None and sh.size(0) != 0 else 0
    mk = (lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)) if P != 0 else \
        (lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev))
    if adam is not None:
        # single-GPU fast path: Adam applied inside the backward kernel, no gradient tensors at all (gslic_rasterize_backward_adam)
        means3D, dc, scales, rotations, dL = map(_f32c, (means3D, dc, scales, rotations, dL_dout_color))
        sh_c = _f32c(sh) if M > 0 else None
        viewmatrix, projmatrix, campos, background = map(_f32c, (viewmatrix, projmatrix, campos, background))
        prm = _params(P, degree, M, H, W, tan_fovx, tan_fovy, limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier, False, debug, False, True)
        p = _lib.ptr
        _lib.check(L.gslic_rasterize_backward_adam(
            ctypes.byref(prm), int(R), int(B), p(background), p(means3D), p(dc), p(sh_c), p(colors), p(scales), p(rotations),
            p(cov3D_precomp), p(viewmatrix), p(projmatrix), p(campos), p(radii.contiguous()),
            ctypes.c_void_p(geomBuffer.data_ptr()), ctypes.c_void_p(binningBuffer.data_ptr()), ctypes.c_void_p(imageBuffer.data_ptr()),
            ctypes.c_void_p(sampleBuffer.data_ptr()), p(dL), None, None, None, None, None, None, float(lambda_erank), ctypes.byref(adam),
            _lib.current_stream_ptr()))
        return None
    if rgb_out is not None:
        # N > 1 exchange (gslic_rasterize_backward_rgb): the clamp-masked colour gradient [P,3] is written instead of dL_ddc / dL_dsh; the
        # four other parameter gradients go into the caller's storage (`out`), nothing else is materialised
        assert out is not None
        if P != 0:
            means3D, dc, scales, rotations, dL = map(_f32c, (means3D, dc, scales, rotations, dL_dout_color))
            sh_c = _f32c(sh) if M > 0 else None
            viewmatrix, projmatrix, campos, background = map(_f32c, (viewmatrix, projmatrix, campos, background))
            prm = _params(P, degree, M, H, W, tan_fovx, tan_fovy, limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier, False, debug, False, raw_params)
            p = _lib.ptr
            if rows is not None:
                # chunked (gslic_rasterize_backward_rgb_rows): Gaussians [rows[0], rows[1]) only; the kernels index the gradient pointers by the
                # ABSOLUTE Gaussian index, so a caller that keeps its chunks in separate blocks passes `out_addr` = {name: address of row 0}
                addr = out_addr or {k: out[k].data_ptr() for k in ("opacity", "xyz", "scaling", "rotation")}
                rgb_addr = (out_addr or {}).get("rgb", rgb_out.data_ptr())
                vp = ctypes.c_void_p
                _lib.check(L.gslic_rasterize_backward_rgb_rows(
                    ctypes.byref(prm), int(R), int(B), p(background), p(means3D), p(dc), p(sh_c), p(colors), p(scales), p(rotations),
                    p(cov3D_precomp), p(viewmatrix), p(projmatrix), p(campos), p(radii.contiguous()),
                    vp(geomBuffer.data_ptr()), vp(binningBuffer.data_ptr()), vp(imageBuffer.data_ptr()), vp(sampleBuffer.data_ptr()), p(dL),
                    vp(addr["opacity"]), vp(addr["xyz"]), vp(rgb_addr), vp(addr["scaling"]), vp(addr["rotation"]),
                    float(lambda_erank), int(rows[0]), int(rows[1]), int(bool(skip_blend)), _lib.current_stream_ptr()))
                return None
            pay_vis, pay_campos = (payload or (None, None))   # the rest of the rank's all-gather payload, written by the kernel (no compare / copy launches)
            _lib.check(L.gslic_rasterize_backward_rgb_payload(
                ctypes.byref(prm), int(R), int(B), p(background), p(means3D), p(dc), p(sh_c), p(colors), p(scales), p(rotations),
                p(cov3D_precomp), p(viewmatrix), p(projmatrix), p(campos), p(radii.contiguous()),
                ctypes.c_void_p(geomBuffer.data_ptr()), ctypes.c_void_p(binningBuffer.data_ptr()), ctypes.c_void_p(imageBuffer.data_ptr()),
                ctypes.c_void_p(sampleBuffer.data_ptr()), p(dL), p(out["opacity"]), p(out["xyz"]), p(rgb_out), p(out["scaling"]), p(out["rotation"]),
                float(lambda_erank), p(pay_vis), p(pay_campos), _lib.current_stream_ptr()))
        return None
    if out is not None:
        # caller-provided gradient storage (e.g. views of one flat slab for a zero-copy all-reduce); the tensors the host
        # discards (means2D, conic, colors_precomp, cov3D: rasterizer.cpp:171-182) are not materialised at all
        dL_dmeans3D, dL_ddc, dL_dsh = out["xyz"], out["features_dc"], out["features_rest"]
        dL_dopacities, dL_dscales, dL_drotations = out["opacity"], out["scaling"], out["rotation"]
        dL_dmeans2D = dL_dcolors = dL_dconic = dL_dcov3D = None
    else:
        dL_dmeans3D, dL_dmeans2D, dL_dcolors = mk(P, 3), mk(P, 3), mk(P, 3)
        dL_dconic, dL_dopacities, dL_dcov3D = mk(P, 2, 2), mk(P, 1), mk(P, 6)
        dL_ddc, dL_dsh, dL_dscales, dL_drotations = mk(P, 1, 3), mk(P, M, 3), mk(P, 3), mk(P, 4)
    if P != 0:
        means3D, dc, scales, rotations, dL = map(_f32c, (means3D, dc, scales, rotations, dL_dout_color))
        sh_c = _f32c(sh) if M > 0 else None
        viewmatrix, projmatrix, campos, background = map(_f32c, (viewmatrix, projmatrix, campos, background))
        prm = _params(P, degree, M, H, W, tan_fovx, tan_fovy, limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier,
                      False, debug, False, raw_params)
        p = _lib.ptr
        common = (ctypes.byref(prm), int(R), int(B), p(background), p(means3D), p(dc), p(sh_c), p(colors), p(scales), p(rotations),
                  p(cov3D_precomp), p(viewmatrix), p(projmatrix), p(campos), p(radii.contiguous()),
                  ctypes.c_void_p(geomBuffer.data_ptr()), ctypes.c_void_p(binningBuffer.data_ptr()),
                  ctypes.c_void_p(imageBuffer.data_ptr()), ctypes.c_void_p(sampleBuffer.data_ptr()), p(dL),
                  p(dL_dmeans2D), p(dL_dconic), p(dL_dopacities), p(dL_dcolors), p(dL_dmeans3D), p(dL_dcov3D), p(dL_ddc),
                  p(dL_dsh), p(dL_dscales), p(dL_drotations), float(lambda_erank))
        if camera_grads:
            cam = (torch.empty(16, device=dev), torch.empty(16, device=dev), torch.empty(3, device=dev))
            _lib.check(L.gslic_rasterize_backward_camera(*common, p(cam[0]), p(cam[1]), p(cam[2]), _lib.current_stream_ptr()))
        else:
            _lib.check(L.gslic_rasterize_backward(*common, _lib.current_stream_ptr()))
    elif camera_grads:
        cam = (torch.zeros(16, device=dev), torch.zeros(16, device=dev), torch.zeros(3, device=dev))
    res = (dL_dmeans2D, dL_dcolors, dL_dopacities, dL_dmeans3D, dL_dcov3D, dL_ddc, dL_dsh, dL_dscales, dL_drotations)
    return res + cam if camera_grads else res


class GaussianRasterizerFunction(torch.autograd.Function):
    """rasterizer.cpp:21-183.  forward returns (color, radii, final_T); only d/dcolor flows back."""

    @staticmethod
    def forward(ctx, means3D, means2D, dc, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        (R, B, color, final_T, radii, geom, binning, img, sample) = rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.limx_neg, rs.limx_pos, rs.limy_neg,
            rs.limy_pos, dc, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, rs.no_color, tie_rank=rs.tie_rank)
        ctx.rs, ctx.R, ctx.B = rs, R, B
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, dc, sh, geom, binning, img, sample)
        ctx.mark_non_differentiable(radii, final_T)
        return color, radii, final_T

    @staticmethod
    def backward(ctx, dL_dcolor, _dL_dradii, _dL_dfinal_T):
        rs = ctx.rs
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, dc, sh, geom, binning, img, sample = ctx.saved_tensors
        (dL_dmeans2D, dL_dcolors_precomp, dL_dopacities, dL_dmeans3D, dL_dcov3Ds_precomp, dL_ddc, dL_dsh, dL_dscales,
         dL_drotations) = rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos, dL_dcolor, dc, sh,
            rs.sh_degree, rs.campos, geom, ctx.R, binning, img, ctx.B, sample, rs.lambda_erank, False)
        return (dL_dmeans3D, dL_dmeans2D, dL_ddc, dL_dsh, None, dL_dopacities, dL_dscales, dL_drotations, None, None)


class GaussianRasterizer(torch.nn.Module):
    """rasterizer.cpp:185-216."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, dc, shs, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        dev = means3D.device
        colors_precomp = torch.empty(0, device=dev)   # rasterizer.cpp:200-201: always empty
        cov3D_precomp = torch.empty(0, device=dev)
        return GaussianRasterizerFunction.apply(means3D, means2D, dc, shs, colors_precomp, opacities, scales, rotations,
                                                cov3D_precomp, self.raster_settings)


class RawGaussianRasterizerFunction(torch.autograd.Function):
    """The autograd node behind render(): same forward / backward as GaussianRasterizerFunction (rasterizer.cpp:21-183), but on the model's
    RAW leaf tensors — sigmoid(opacity_), exp(scaling_), normalize(rotation_) of gaussian.cpp:147-175 run inside preprocess / preprocess_bwd
    (raw_params = 1), so the six LibTorch elementwise launches of the activations, the ten of their backward nodes and the zeros_like of the
    screen-space points disappear from the step.  Gradients are w.r.t. the raw tensors: exactly what autograd would have chained to.
    Only the six gradients the host's optimiser reads are materialised (rasterizer.cpp:171-182 discards the other four)."""

    @staticmethod
    def forward(ctx, xyz, dc, sh, opacity_raw, scaling_raw, rotation_raw, rs):
        e = torch.empty(0, device=xyz.device)
        (R, B, color, final_T, radii, geom, binning, img, sample) = rasterize_gaussians(
            rs.bg, xyz, e, opacity_raw, scaling_raw, rotation_raw, rs.scale_modifier, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
            rs.image_height, rs.image_width, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos, dc, sh, rs.sh_degree, rs.campos, rs.prefiltered,
            rs.debug, rs.no_color, raw_params=True, tie_rank=rs.tie_rank)
        ctx.rs, ctx.R, ctx.B = rs, R, B
        ctx.save_for_backward(xyz, dc, sh, opacity_raw, scaling_raw, rotation_raw, radii, geom, binning, img, sample)
        ctx.mark_non_differentiable(radii, final_T)
        return color, radii, final_T

    @staticmethod
    def backward(ctx, dL_dcolor, _dL_dradii, _dL_dfinal_T):
        rs = ctx.rs
        xyz, dc, sh, opacity_raw, scaling_raw, rotation_raw, radii, geom, binning, img, sample = ctx.saved_tensors
        e = torch.empty(0, device=xyz.device)
        out = dict(xyz=torch.empty_like(xyz), features_dc=torch.empty_like(dc), features_rest=torch.empty_like(sh), opacity=torch.empty_like(opacity_raw),
                   scaling=torch.empty_like(scaling_raw), rotation=torch.empty_like(rotation_raw))   # (every row is written by the kernel)
        rasterize_gaussians_backward(rs.bg, xyz, radii, e, scaling_raw, rotation_raw, rs.scale_modifier, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                                     rs.tanfovy, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos, dL_dcolor, dc, sh, rs.sh_degree, rs.campos, geom, ctx.R,
                                     binning, img, ctx.B, sample, rs.lambda_erank, False, raw_params=True, out=out)
        return out["xyz"], out["features_dc"], out["features_rest"], out["opacity"], out["scaling"], out["rotation"], None


def _raw_leaves(model):
    """The six raw parameter tensors of a GaussianModel (xyz_, features_dc_, features_rest_, opacity_, scaling_, rotation_: gaussian.h:153-158),
    or None when the model only offers the activated accessors."""
    # An explicit marker, not duck typing: a model whose attributes of these names hold ACTIVATED values would otherwise be
    # activated a second time.  trainer.GaussianModel sets raw_parameter_leaves = True; any other model opts in by providing raw_leaves().
    if callable(getattr(model, "raw_leaves", None)):
        return tuple(model.raw_leaves())
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")
    if getattr(model, "raw_parameter_leaves", False) and all(torch.is_tensor(getattr(model, n, None)) for n in names):
        return tuple(getattr(model, n) for n in names)
    return None


def render(camera, model, bg_color, no_color=False, scaling_modifier=1.0, raw=None):
    """renderer.cpp:21-88 — same signature, same five results (image, final_T, screenspace_points, visible, radii).  `camera` =
    gaussian_lic_amd.camera.Camera with device tensors attached via to_device(); `model` exposes get_xyz / get_opacity / get_scaling /
    get_rotation / get_features_dc / get_features_rest, sh_degree, lambda_erank.

    raw (default: on whenever the model exposes its raw leaf tensors; GSLIC_RENDER_RAW=0 turns it off): the renderer feeds the RAW parameters to
    one autograd node whose kernels apply the activations (RawGaussianRasterizerFunction) instead of calling getOpacity() / getScaling() /
    getRotation() (renderer.cpp:57-63).  Image and gradients equal the operator path's up to fp32 rounding of the activation chain
    (tests/test_fused_gpu.py).  screenspace_points is then a zero-stride view of one zero row: the reference allocates and zero-fills
    [P,3] for a gradient (dL_dmeans2D) no caller of render() reads (gaussian.cpp:506,683,757,797); raw=False restores it."""
    import os
    leaves = _raw_leaves(model)
    if raw is None:
        raw = leaves is not None and os.environ.get("GSLIC_RENDER_RAW", "1") != "0"
    rs = GaussianRasterizationSettings(
        camera.image_height, camera.image_width, float(camera.tanfovx), float(camera.tanfovy), float(camera.limx_neg),
        float(camera.limx_pos), float(camera.limy_neg), float(camera.limy_pos), bg_color, scaling_modifier,
        camera.d_world_view_transform, camera.d_full_proj_transform, model.sh_degree, camera.d_camera_center, False, False,
        no_color, model.lambda_erank, getattr(model, "tie_rank", None))
    if raw:
        if leaves is None:
            raise TypeError("render(raw=True): the model does not expose its raw parameter leaves (set raw_parameter_leaves = True on a model whose "
                            "xyz / features_dc / features_rest / opacity / scaling / rotation attributes are the PRE-activation tensors, or provide raw_leaves())")
        xyz, dc, rest, opacity, scaling, rotation = leaves
        image, radii, final_T = RawGaussianRasterizerFunction.apply(xyz, dc, rest, opacity, scaling, rotation, rs)
        screenspace_points = torch.zeros(1, 3, dtype=xyz.dtype, device=xyz.device).expand(xyz.shape[0], 3)
        return image, final_T, screenspace_points, radii > 0, radii
    xyz = model.get_xyz()
    screenspace_points = torch.zeros_like(xyz, requires_grad=True)
    rasterizer = GaussianRasterizer(rs)
    image, radii, final_T = rasterizer(xyz, screenspace_points, model.get_opacity(), model.get_features_dc(),
                                       model.get_features_rest(), None, model.get_scaling(), model.get_rotation(), None)
    return image, final_T, screenspace_points, radii > 0, radii


def sh_grad_from_rgb(means3D, campos_all, rgb_all, degree, dL_ddc, dL_dsh, input_is_ddc=False, n_views=None, view_stride=0):
    """gslic_sh_grad_from_rgb: dL_ddc [P,1,3] and dL_dsh [P,M,3] summed over the views from the views' masked colour gradients
    rgb_all [n_views,P,3] and camera centres campos_all [n_views,3] (all device fp32, contiguous); written in place.
    view_stride > 0 (floats): rgb_all / campos_all point at view 0's block inside an all-gathered payload, n_views blocks view_stride apart."""
    P = means3D.size(0)
    n = rgb_all.size(0) if n_views is None else int(n_views)
    M = dL_dsh.size(1) if dL_dsh is not None and dL_dsh.numel() else 0
    assert dL_ddc.is_contiguous() and (M == 0 or dL_dsh.is_contiguous())
    if not view_stride:
        assert rgb_all.is_contiguous() and campos_all.is_contiguous() and tuple(rgb_all.shape) == (n, P, 3) and tuple(campos_all.shape) == (n, 3)
    p = _lib.ptr
    _lib.check(_lib.lib().gslic_sh_grad_from_rgb(P, int(degree), M, n, p(_f32c(means3D)), p(campos_all), p(rgb_all), int(bool(input_is_ddc)), p(dL_ddc),
                                                 p(dL_dsh) if M else None, int(view_stride), _lib.current_stream_ptr()))


def debug_export(settings, P, M, R, B, geom, binning, img, sample, what=("tiles_touched", "point_list", "ranges")):
    """Test-only: copy stage boundaries out of the opaque scratch buffers (gslic_debug_export)."""
    L = _lib.lib()
    rs = settings
    dev = geom.device
    H, W = rs.image_height, rs.image_width
    T = ((W + 15) // 16) * ((H + 15) // 16)
    prm = _params(P, rs.sh_degree, M, H, W, rs.tanfovx, rs.tanfovy, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos,
                  rs.scale_modifier, False, False, rs.no_color)
    out = {}
    mk = dict(
        tiles_touched=lambda: torch.zeros(P, dtype=torch.int32, device=dev),
        means2D=lambda: torch.zeros(P, 2, device=dev), depths=lambda: torch.zeros(P, device=dev),
        conic_opacity=lambda: torch.zeros(P, 4, device=dev), rgb=lambda: torch.zeros(P, 3, device=dev),
        sorted_keys=lambda: torch.zeros(max(R, 1), dtype=torch.int64, device=dev),
        point_list=lambda: torch.zeros(max(R, 1), dtype=torch.int32, device=dev),
        ranges=lambda: torch.zeros(T, 2, dtype=torch.int32, device=dev),
        n_contrib=lambda: torch.zeros(H, W, dtype=torch.int32, device=dev),
        max_contrib=lambda: torch.zeros(T, dtype=torch.int32, device=dev))
    order = ["tiles_touched", "means2D", "depths", "conic_opacity", "rgb", "sorted_keys", "point_list", "ranges", "n_contrib",
             "max_contrib"]
    args = []
    for k in order:
        if k in what:
            out[k] = mk[k]()
            args.append(ctypes.c_void_p(out[k].data_ptr()))
        else:
            args.append(None)
    bp = lambda t: ctypes.c_void_p(t.data_ptr()) if t.numel() else None
    _lib.check(L.gslic_debug_export(ctypes.byref(prm), int(R), int(B), bp(geom), bp(binning), bp(img), bp(sample), *args,
                                    _lib.current_stream_ptr()))
    torch.cuda.synchronize()
    for k in ("sorted_keys", "point_list"):
        if k in out:
            out[k] = out[k][:R]
    return out
