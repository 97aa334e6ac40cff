"""Helpers shared by the -m gpu tests: run the HIP path through the C-ABI front-end on seeded scenes."""
import numpy as np
import torch

from gaussian_lic_amd import rasterizer as rz
from gaussian_lic_amd.synthetic import activate


def settings_from(cam, deg, dev, no_color=False, lambda_erank=0.0, debug=False):
    return rz.GaussianRasterizationSettings(
        cam.image_height, cam.image_width, float(cam.tanfovx), float(cam.tanfovy), float(cam.limx_neg), float(cam.limx_pos),
        float(cam.limy_neg), float(cam.limy_pos), torch.zeros(3, device=dev), 1.0,
        torch.from_numpy(cam.world_view_transform).to(dev), torch.from_numpy(cam.full_proj_transform).to(dev), deg,
        torch.from_numpy(cam.camera_center).to(dev), False, debug, no_color, lambda_erank)


def hip_forward(raw, cam, no_color=False, export=(), debug=False):
    dev = torch.device("cuda:0")
    act = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in activate(raw).items()}
    rs = settings_from(cam, act["D"], dev, no_color, debug=debug)
    empty = torch.empty(0, device=dev)
    out = rz.rasterize_gaussians(rs.bg, act["means"], empty, act["opac"], act["scales"], act["rots"], 1.0, empty, rs.viewmatrix,
                                 rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.limx_neg, rs.limx_pos,
                                 rs.limy_neg, rs.limy_pos, act["dc"], act["shs"], act["D"], rs.campos, False, debug, no_color)
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
    g = rz.rasterize_gaussians_backward(rs.bg, act["means"], fwd["radii"], empty, act["scales"], act["rots"], 1.0, empty,
                                        rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.limx_neg, rs.limx_pos,
                                        rs.limy_neg, rs.limy_pos, dL_dpix.to(dev), act["dc"], act["shs"], act["D"], rs.campos,
                                        geom, fwd["R"], binning, img, fwd["B"], sample, lambda_erank, False)
    torch.cuda.synchronize()
    names = ["dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale", "dL_drot"]
    return {n: t.cpu().numpy() for n, t in zip(names, g)}


def npy(t):
    return t.detach().cpu().numpy()
