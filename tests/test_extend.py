"""extend() point selection + in-place append: oracle properties on CPU, HIP parity on GPU."""
import numpy as np
import pytest
import torch

from conftest import make_scene, rel_err


def _lidar_frame(n, W, H, seed, fx, fy, cx, cy):
    """n synthetic LiDAR returns in the camera frame of the identity pose, several per pixel on purpose, a few behind/outside."""
    rng = np.random.default_rng(seed)
    u, v = rng.uniform(-0.1 * W, 1.1 * W, n), rng.uniform(-0.1 * H, 1.1 * H, n)
    u[: n // 4] = np.floor(u[: n // 4] / 8) * 8 + 0.5      # pile points onto shared pixels
    v[: n // 4] = np.floor(v[: n // 4] / 8) * 8 + 0.5
    z = rng.uniform(1.0, 40.0, n)
    z[rng.random(n) < 0.01] *= -1.0
    pts = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], 1).astype(np.float32)
    rsp = np.where(rng.random(n) < 0.02, -1.0, np.abs(z)).astype(np.float32)
    col = rng.random((n, 3)).astype(np.float32)
    return pts, col, rsp


def test_oracle_extend_select_properties(oracle32):
    W, H, n = 96, 64, 5000
    fx = fy = 0.675 * W
    cx, cy = 0.4857 * W, 0.5215 * H
    pts, col, rsp = _lidar_frame(n, W, H, 0, fx, fy, cx, cy)
    T = np.random.default_rng(1).random((H, W)).astype(np.float32)
    T[:, : W // 2] = 0.001                                  # left half already opaque: alpha = 0.999 >= 0.99
    keep = oracle32.extend_select(pts, rsp, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), fx, fy, cx, cy, W, H, T)
    # brute-force restatement with a python dict, like the reference's unordered_map
    best = {}
    for i in range(n):
        x, y, z = pts[i]
        xf = np.floor(np.float32(np.float32(x * np.float32(fx)) / z) + np.float32(cx))
        yf = np.floor(np.float32(np.float32(y * np.float32(fy)) / z) + np.float32(cy))
        key = (float(xf), float(yf))
        if key not in best or z < best[key][1]:
            best[key] = (i, z)
    want = np.zeros(n, bool)
    for (xf, yf), (i, z) in best.items():
        if 0 <= xf < W and 0 <= yf < H and rsp[i] > 0 and (np.float32(1.0) - T[int(yf), int(xf)]) < np.float32(0.99):
            want[i] = True
    np.testing.assert_array_equal(keep, want)
    assert keep.sum() > 100
    xs = np.floor((pts[keep, 0] * fx) / pts[keep, 2] + cx)
    assert xs.min() >= W // 2                               # nothing lands on the opaque half
    rows = oracle32.extend_emit(keep, pts, col, rsp, 1.0, 0.5 * (fx + fy), 15)
    assert rows["xyz"].shape[0] == keep.sum() and np.allclose(rows["rotation"], [1, 0, 0, 0])
    assert np.allclose(rows["opacity"], np.log(0.1 / 0.9)) and float(np.abs(rows["rest"]).max()) == 0.0
    assert np.allclose(rows["dc"][:, 0], (col[keep] - 0.5) / 0.28209479177387814, atol=1e-6)


GOLDEN_FRAMES = ["extend_identity_96x64", "extend_rot90_160x120", "extend_ties_33x17"]


def _golden_frame(name):
    """tests/golden/extend_*.npz: a LiDAR frame and what the REFERENCE's own extend() (gaussian.cpp:499-638, compiled unmodified on CPU
    LibTorch by oracle/ref_build/make_extend_golden.py) appended for it — the survivor set (ascending frame indices; the reference's own
    order is unordered_map iteration order) and the six new-Gaussian tensors in that order."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    R_wc, t_wc = g["R_wc"], g["t_wc"]
    R_cw = R_wc.T                                    # gaussian.cpp:527-530, in double like the reference's Eigen types, then cast (:531-539)
    t_cw = -R_cw @ t_wc
    return g, R_cw.astype(np.float32), t_cw.astype(np.float32)


@pytest.mark.parametrize("name", GOLDEN_FRAMES)
def test_oracle_extend_equals_reference_golden(oracle32, name):
    """The oracle's selection and new-Gaussian rows against the reference's own extend(): frames with several returns per pixel,
    equal-depth ties, points off the image / behind the camera, non-positive sensor depths, alpha exactly at the 0.99 cut."""
    g, R_cw, t_cw = _golden_frame(name)
    fx, fy, cx, cy = (float(v) for v in g["intr"])
    W, H = (int(v) for v in g["size"])
    keep = oracle32.extend_select(g["points"], g["depths_rsp"], R_cw, t_cw, fx, fy, cx, cy, W, H, g["final_T"])
    np.testing.assert_array_equal(np.nonzero(keep)[0], g["keep"])
    assert g["keep"].size > 100
    rows = oracle32.extend_emit(keep, g["points"], g["colors"], g["depths_rsp"], 1.0, 0.5 * (fx + fy), 15)
    for key in ("xyz", "rest", "rotation"):
        np.testing.assert_array_equal(rows[key].reshape(g["row_" + key].shape), g["row_" + key])
    for key in ("dc", "opacity", "scaling"):        # (c - 0.5) / C0, log(0.1 / 0.9), log(range / focal): libm vs LibTorch's vectorised log
        assert rel_err(rows[key].reshape(g["row_" + key].shape), g["row_" + key]) < 1e-6, key


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_FRAMES)
def test_hip_extend_equals_reference_golden(name):
    """gslic_extend_select / gslic_extend_emit (the 64-bit atomic-min z-buffer that replaces the reference's CPU hash map) against the
    reference's own extend(): identical survivor set, identical rows."""
    from gpu_helpers import hip_extend_rows
    g, R_cw, t_cw = _golden_frame(name)
    W, H = (int(v) for v in g["size"])
    k, rows = hip_extend_rows(g["points"], g["colors"], g["depths_rsp"], R_cw, t_cw, g["intr"], W, H, g["final_T"])
    assert k == g["keep"].size
    np.testing.assert_array_equal(rows["xyz"], g["row_xyz"])            # ascending index on both sides: same set <=> same rows
    np.testing.assert_array_equal(rows["rest"], g["row_rest"])
    np.testing.assert_array_equal(rows["rotation"], g["row_rotation"])
    for key in ("dc", "opacity", "scaling"):
        assert rel_err(rows[key], g["row_" + key]) < 1e-6, key


@pytest.mark.gpu
def test_extend_matches_oracle_and_appends_in_place(oracle32):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import trainer
    from gaussian_lic_amd.rasterizer import render
    from gaussian_lic_amd.synthetic import gt_image
    W, H, P, n = 320, 240, 2000, 30000   # sparse model: most pixels are still transparent
    raw, sc, camd, cam = make_scene("random", P, W, H, 3, 51)
    dev = torch.device("cuda:0")
    cam.to_device(dev)
    model = trainer.GaussianModel(raw, dev)              # capacity == P: the append must grow the storage
    model.training_setup()
    bg = torch.zeros(3, device=dev)
    gt = gt_image(H, W).to(dev)
    for _ in range(2):
        trainer.training_step(model, cam, gt, bg)        # non-trivial Adam moments before the append
    before = {k: getattr(model, k).detach().clone() for k in model.NAMES}
    m_before = [s["exp_avg"].clone() for s in model.optimizer.state]
    fx, fy, cx, cy = float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy)
    pts, col, rsp = _lidar_frame(n, W, H, 3, fx, fy, cx, cy)
    with torch.no_grad():
        final_T = render(cam, model, bg, no_color=True)[1].cpu().numpy()
    k = model.extend(cam, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(rsp).to(dev),
                     torch.eye(3), torch.zeros(3), (fx, fy, cx, cy))
    keep = oracle32.extend_select(pts, rsp, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), fx, fy, cx, cy, W, H, final_T)
    assert k == int(keep.sum()) and k > 1000
    rows = oracle32.extend_emit(keep, pts, col, rsp, 1.0, 0.5 * (fx + fy), 15)
    assert model.P == P + k and model.capacity >= P + k
    for name, key in (("xyz", "xyz"), ("features_dc", "dc"), ("features_rest", "rest"), ("opacity", "opacity"), ("scaling", "scaling"),
                      ("rotation", "rotation")):
        t = getattr(model, name).detach()
        assert torch.equal(t[:P], before[name]), name                              # old rows untouched
        got = t[P:].cpu().numpy()
        if name in ("xyz", "features_rest", "rotation"):
            np.testing.assert_array_equal(got, rows[key].reshape(got.shape))       # exact: copies / constants, ascending order
        else:
            assert rel_err(got, rows[key].reshape(got.shape)) < 1e-6, name
    for s, mb in zip(model.optimizer.state, m_before):                             # moments: old rows kept, new rows zero
        assert torch.equal(s["exp_avg"][:P], mb) and float(s["exp_avg"][P:].abs().max()) == 0.0
        assert s["exp_avg"].shape[0] == P + k and float(s["exp_avg_sq"][P:].abs().max()) == 0.0
    loss, vis = trainer.training_step(model, cam, gt, bg)                          # the grown model trains
    assert vis.shape[0] == P + k and torch.isfinite(loss)
    k2 = model.extend(cam, torch.from_numpy(pts).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(rsp).to(dev),
                      torch.eye(3), torch.zeros(3), (fx, fy, cx, cy))
    assert 0 <= k2 < k                                                              # pixels covered by the new Gaussians reject more points


@pytest.mark.gpu
def test_extend_edge_cases_and_interleaved_forwards():
    """extend() with an empty frame and with a frame that lands only outside the image inserts nothing and leaves the model untouched;
    two forwards issued back to back keep independent scratch, so their backwards can run in any order."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import trainer
    from gpu_helpers import hip_backward, hip_forward
    from gaussian_lic_amd.synthetic import pixel_grad
    W, H, P = 160, 120, 3000
    raw, sc, camd, cam = make_scene("random", P, W, H, 3, 61)
    dev = torch.device("cuda:0")
    cam.to_device(dev)
    model = trainer.GaussianModel(raw, dev)
    model.training_setup()
    fx, fy, cx, cy = float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy)
    before = model.xyz.detach().clone()
    e3, e1 = torch.empty(0, 3, device=dev), torch.empty(0, device=dev)
    assert model.extend(cam, e3, e3, e1, torch.eye(3), torch.zeros(3), (fx, fy, cx, cy)) == 0
    far = torch.tensor([[1e4, 1e4, 1.0], [-1e4, 0.0, 2.0], [0.0, 0.0, -5.0]], device=dev)   # off-image / behind the camera
    assert model.extend(cam, far, torch.rand(3, 3, device=dev), far[:, 2].abs().contiguous(), torch.eye(3), torch.zeros(3), (fx, fy, cx, cy)) == 0
    assert model.P == P and torch.equal(model.xyz.detach(), before)
    # interleaved forwards: A, B, then backward(B), backward(A) == the separate sequences
    raw2, _, _, cam2 = make_scene("random", 2000, 96, 64, 3, 62)
    dLa, dLb = pixel_grad(H, W, seed=1), pixel_grad(64, 96, seed=2)
    fa, fb = hip_forward(raw, cam), hip_forward(raw2, cam2)
    gb, ga = hip_backward(fb, dLb), hip_backward(fa, dLa)
    fa2 = hip_forward(raw, cam); ga2 = hip_backward(fa2, dLa)
    fb2 = hip_forward(raw2, cam2); gb2 = hip_backward(fb2, dLb)
    for k in ga:
        np.testing.assert_array_equal(ga[k], ga2[k])
        np.testing.assert_array_equal(gb[k], gb2[k])
