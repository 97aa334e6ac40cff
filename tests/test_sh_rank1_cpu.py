"""CPU: the identity behind the N > 1 exchange (gslic_rasterize_backward_rgb + gslic_sh_grad_from_rgb), checked on the ORACLE's backward
(oracle/gs_oracle.c follows backward.cu:27-136): dL_dsh is the outer product of direction-only coefficients with the clamp-masked colour
gradient, and dL_ddc is SH_C0 times the same vector — so shipping 3 floats per Gaussian and view carries all 48."""
import numpy as np
import pytest
import torch

from conftest import make_scene
from sh_rank1_ref import SH_C0, rows_one_view


@pytest.mark.parametrize("deg", [3, 1])
def test_oracle_sh_gradient_is_rank_one(deg):
    from gaussian_lic_amd.synthetic import pixel_grad
    from oracle.oracle import Oracle
    W, H, P = 160, 120, 3000
    raw, sc, camd, cam = make_scene("random", P, W, H, deg, 11)
    orc = Oracle(np.float32)
    ref = orc.forward(sc, camd)
    g = orc.backward(sc, camd, ref, pixel_grad(H, W, seed=1).numpy())
    ddc = torch.from_numpy(np.asarray(g["dL_ddc"], np.float64).reshape(P, 3))
    dsh = torch.from_numpy(np.asarray(g["dL_dsh"], np.float64).reshape(P, -1, 3))
    M = dsh.shape[1]
    means = torch.from_numpy(np.asarray(sc["means"], np.float64).reshape(P, 3))
    campos = torch.from_numpy(np.asarray(camd["campos"], np.float64).reshape(3))
    dc_r, sh_r = rows_one_view(means, campos, ddc / SH_C0, deg, M)
    scale = float(dsh.abs().max())
    assert scale > 0
    assert float((sh_r - dsh).abs().max()) <= 2e-6 * scale
    assert (ddc != 0).any(dim=1).sum() > P // 10   # a real share of the Gaussians carries colour gradient
    # above the active degree the rows are exact zeros on both sides
    nk = (deg + 1) ** 2 - 1
    assert float(dsh[:, nk:].abs().max() if nk < M else 0.0) == 0.0
