"""Helpers shared by the -m gpu tests: run the HIP path through the C-ABI front-end on seeded scenes."""
import numpy as np
import torch

from gaussian_lic_amd import rasterizer as rz
from gaussian_lic_amd.synthetic import activate


def settings_from(cam, deg, dev, no_color=False, lambda_erank=0.0, debug=False, scale_modifier=1.0):
    return rz.GaussianRasterizationSettings(
        cam.image_height, cam.image_width, float(cam.tanfovx), float(cam.tanfovy), float(cam.limx_neg), float(cam.limx_pos),
        float(cam.limy_neg), float(cam.limy_pos), torch.zeros(3, device=dev), float(scale_modifier),
        torch.from_numpy(cam.world_view_transform).to(dev), torch.from_numpy(cam.full_proj_transform).to(dev), deg,
        torch.from_numpy(cam.camera_center).to(dev), False, debug, no_color, lambda_erank)


def hip_forward(raw, cam, no_color=False, export=(), debug=False, scale_modifier=1.0, tie_rank=None, act=None):
    """tie_rank: int32 device tensor [P] — the rows' original indices when the map's rows are given in a permuted order
    (gslic_raster_params.tie_rank: what trainer.GaussianModel(order="morton") passes).
    act: the ACTIVATED scene (synthetic.activate's dict) instead of `raw` — a caller that permutes rows activates first and permutes the activated
    tensors, so that both sides of a comparison see the same bits whatever the host's LibTorch does at its chunk boundaries."""
    dev = torch.device("cuda:0")
    act = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in (activate(raw) if act is None else act).items()}
    rs = settings_from(cam, act["D"], dev, no_color, debug=debug, scale_modifier=scale_modifier)
    empty = torch.empty(0, device=dev)
    out = rz.rasterize_gaussians(rs.bg, act["means"], empty, act["opac"], act["scales"], act["rots"], rs.scale_modifier, empty, rs.viewmatrix,
                                 rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.limx_neg, rs.limx_pos,
                                 rs.limy_neg, rs.limy_pos, act["dc"], act["shs"], act["D"], rs.campos, False, debug, no_color, tie_rank=tie_rank)
    R, B, color, final_T, radii, geom, binning, img, sample = out
    res = dict(R=R, B=B, color=color, final_T=final_T, radii=radii, bufs=(geom, binning, img, sample), act=act, rs=rs)
    if export:
        P = act["means"].shape[0]
        M = act["shs"].shape[1] if act["shs"].numel() else 0
        res["dbg"] = rz.debug_export(rs, P, M, R, B, geom, binning, img, sample, what=tuple(export))
    torch.cuda.synchronize()
    return res


def hip_backward(fwd, dL_dpix, lambda_erank=0.0):
    act, rs = fwd["act"], fwd["rs"]
    dev = act["means"].device
    empty = torch.empty(0, device=dev)
    geom, binning, img, sample = fwd["bufs"]
    g = rz.rasterize_gaussians_backward(rs.bg, act["means"], fwd["radii"], empty, act["scales"], act["rots"], rs.scale_modifier, empty,
                                        rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.limx_neg, rs.limx_pos,
                                        rs.limy_neg, rs.limy_pos, dL_dpix.to(dev), act["dc"], act["shs"], act["D"], rs.campos,
                                        geom, fwd["R"], binning, img, fwd["B"], sample, lambda_erank, False)
    torch.cuda.synchronize()
    names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale", "dL_drot"]
    return {n: t.cpu().numpy() for n, t in zip(names, g)}


def npy(t):
    return t.detach().cpu().numpy()


def hip_extend_rows(points, colors, depths_rsp, R_cw, t_cw, intr, W, H, final_T, M=15, scaling_scale=1.0):
    """gslic_extend_select + gslic_extend_emit on a given transmittance image (what trainer.GaussianModel.extend runs after its no_color
    render): returns (k, dict of the six new-Gaussian row tensors as numpy, ascending point index)."""
    import ctypes
    from gaussian_lic_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    pts, col, rsp, Rc, tc, fT = t(points), t(colors), t(depths_rsp), t(R_cw), t(t_cw), t(final_T)
    n = int(pts.shape[0])
    fx, fy, cx, cy = (float(v) for v in intr)
    scratch = _lib.TensorAllocator(dev)
    flags, pos, count = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int32(0)
    p = _lib.ptr
    _lib.check(L.gslic_extend_select(n, p(pts), p(rsp), p(Rc), p(tc), fx, fy, cx, cy, int(W), int(H), p(fT), scratch.cb, None, ctypes.byref(flags),
                                     ctypes.byref(pos), ctypes.byref(count), _lib.current_stream_ptr()))
    k = count.value
    rows = dict(xyz=torch.empty(k, 3, device=dev), dc=torch.empty(k, 1, 3, device=dev), rest=torch.empty(k, M, 3, device=dev),
                opacity=torch.empty(k, 1, device=dev), scaling=torch.empty(k, 3, device=dev), rotation=torch.empty(k, 4, device=dev))
    if k:
        q = lambda x: ctypes.c_void_p(x.data_ptr()) if x.numel() else None
        _lib.check(L.gslic_extend_emit(n, flags, pos, p(pts), p(col), p(rsp), float(scaling_scale), (fx + fy) / 2.0, int(M), q(rows["xyz"]), q(rows["dc"]),
                                       q(rows["rest"]), q(rows["opacity"]), q(rows["scaling"]), q(rows["rotation"]), _lib.current_stream_ptr()))
    torch.cuda.synchronize()
    return k, {key: v.cpu().numpy() for key, v in rows.items()}
