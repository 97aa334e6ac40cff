"""Plain-torch restatement of the identity the N > 1 exchange rests on (test infrastructure): computeColorFromSH's backward
(/root/reference/src/rasterizer/cuda_rasterizer/backward.cu:27-136) is the outer product of direction-only coefficients c_k(dir)
(the factors multiplying sh[k] in forward.cu:37-66) with the clamp-masked colour gradient dRGB:
    dL_ddc = SH_C0 * dRGB,   dL_dsh[k] = c_k(normalize(p - campos)) * dRGB.
Used by the CPU tests (against the oracle's backward and inside the gloo exchange test) — never by the product."""
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


def sh_coefs(D, d):
    """d: [P,3] unit directions -> [P,15] coefficients (zeros above the active degree D)."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    c = torch.zeros(d.shape[0], 15, dtype=d.dtype)
    if D > 0:
        c[:, 0], c[:, 1], c[:, 2] = -SH_C1 * y, SH_C1 * z, -SH_C1 * x
    if D > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        c[:, 3], c[:, 4], c[:, 5] = SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2 * zz - xx - yy)
        c[:, 6], c[:, 7] = SH_C2[3] * xz, SH_C2[4] * (xx - yy)
    if D > 2:
        c[:, 8], c[:, 9] = SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * xy * z
        c[:, 10], c[:, 11] = SH_C3[2] * y * (4 * zz - xx - yy), SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy)
        c[:, 12], c[:, 13] = SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy)
        c[:, 14] = SH_C3[6] * x * (xx - 3 * yy)
    return c


def rows_one_view(means, campos, rgb, D, M):
    """(dL_ddc [P,1,3], dL_dsh [P,M,3]) of ONE view from its masked colour gradient rgb [P,3]."""
    d = means - campos.view(1, 3)
    d = d / d.norm(dim=1, keepdim=True)
    c = sh_coefs(D, d)[:, :min(M, 15)]
    sh = torch.zeros(means.shape[0], M, 3, dtype=means.dtype)
    sh[:, :c.shape[1]] = c.unsqueeze(2) * rgb.unsqueeze(1)
    return (SH_C0 * rgb).view(-1, 1, 3), sh


def rows_from_rgb(means, campos_all, rgb_all, D, M):
    """Sum over the views, in view order."""
    dc = torch.zeros(means.shape[0], 1, 3, dtype=means.dtype)
    sh = torch.zeros(means.shape[0], M, 3, dtype=means.dtype)
    for v in range(rgb_all.shape[0]):
        a, b = rows_one_view(means, campos_all[v], rgb_all[v], D, M)
        dc, sh = dc + a, sh + b
    return dc, sh
