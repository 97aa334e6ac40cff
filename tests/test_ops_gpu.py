"""-m gpu: Adam / fused-SSIM / simple-knn HIP kernels (through the C-ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,M", [(1000, 3), (1000, 45), (777, 1), (513, 4)])
def test_adam_parity(oracle32, N, M):
    from gaussian_lic_amd import optim
    rng = np.random.default_rng(0)
    p, g = rng.standard_normal((N, M)).astype(np.float32), rng.standard_normal((N, M)).astype(np.float32)
    m, v = (0.1 * rng.standard_normal((N, M))).astype(np.float32), (0.01 * rng.random((N, M))).astype(np.float32)
    vis = rng.random(N) < 0.6
    dev = "cuda:0"
    tp, tg, tm, tv = (torch.from_numpy(x.copy()).to(dev) for x in (p, g, m, v))
    tvis = torch.from_numpy(vis).to(dev)
    optim.adam_update(tp, tg, tm, tv, tvis, 1.6e-4, 0.9, 0.999, 1e-15, N, M)
    rp, rm, rv = p.copy(), m.copy(), v.copy()
    oracle32.adam(rp, g, rm, rv, vis, 1.6e-4)
    assert rel_err(tp.cpu().numpy(), rp) < 1e-6 and rel_err(tm.cpu().numpy(), rm) < 1e-6 and rel_err(tv.cpu().numpy(), rv) < 1e-6
    # invisible rows untouched, bit for bit
    np.testing.assert_array_equal(tp.cpu().numpy()[~vis], p[~vis])
    np.testing.assert_array_equal(tm.cpu().numpy()[~vis], m[~vis])
    np.testing.assert_array_equal(tv.cpu().numpy()[~vis], v[~vis])


@pytest.mark.parametrize("B,CH,H,W", [(1, 3, 48, 64), (2, 3, 37, 45), (1, 1, 5, 7), (1, 3, 270, 480)])
def test_fused_ssim_parity(oracle32, B, CH, H, W):
    from gaussian_lic_amd import loss
    rng = np.random.default_rng(1)
    a, b = rng.random((B, CH, H, W)).astype(np.float32), rng.random((B, CH, H, W)).astype(np.float32)
    dL = rng.standard_normal((B, CH, H, W)).astype(np.float32)
    dev = "cuda:0"
    ta, tb, tdl = (torch.from_numpy(x).to(dev) for x in (a, b, dL))
    m, d1, d2, d3 = loss.fusedssim(0.01 ** 2, 0.03 ** 2, ta, tb, True)
    rm, r1, r2, r3 = oracle32.ssim_forward(a, b)
    for got, ref in ((m, rm), (d1, r1), (d2, r2), (d3, r3)):
        assert rel_err(got.cpu().numpy(), ref) < 1e-4
    gi = loss.fusedssim_backward(0.01 ** 2, 0.03 ** 2, ta, tb, tdl, d1, d2, d3)
    ri = oracle32.ssim_backward(a, b, dL, r1, r2, r3)
    assert rel_err(gi.cpu().numpy(), ri) < 1e-4
    m2, e1, e2, e3 = loss.fusedssim(0.01 ** 2, 0.03 ** 2, ta, tb, False)
    assert e1.numel() == 0 and torch.equal(m2, m)


@pytest.mark.parametrize("P", [1, 3, 4, 1000, 5000, 100000])
def test_knn_parity(oracle32, P):
    from gaussian_lic_amd import knn
    rng = np.random.default_rng(2)
    pts = (rng.standard_normal((P, 3)) * np.array([10, 3, 7])).astype(np.float32)
    got = knn.distCUDA2(torch.from_numpy(pts).to("cuda:0")).cpu().numpy()
    if P <= 5000:
        ref = oracle32.knn(pts)
    else:  # brute force on a sample of the queries (the oracle is O(P^2))
        from scipy.spatial import cKDTree
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
        ref = (d[:, 1:] ** 2).mean(1)
    if P < 4:
        assert np.all(~np.isfinite(got) | (got > 1e37))
    else:
        assert rel_err(got, ref) < 1e-5
