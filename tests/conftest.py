import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.oracle import Oracle, build
    build()
    return Oracle(np.float32)


@pytest.fixture(scope="session")
def oracle64():
    from oracle.oracle import Oracle, build
    build()
    return Oracle(np.float64)


def make_scene(kind, P, W, H, deg, seed=0, view=None, sigma_scale=1.0):
    """(raw torch params, activated numpy dict for the oracle, camera dict, Camera).
    view: None (identity pose) | k in 0..7 (config-4 views) | a name of camera.SE3_POSES | dict(ypr=, t=, place=) — see camera.resolve_view;
    with place=True the scene is moved rigidly into the pose's frame (synthetic.place_scene).  sigma_scale multiplies every Gaussian's extent
    (adds log(sigma_scale) to the raw scaling): large Gaussians reach the image from beyond the 15 % margins where the cov2D Jacobian is clamped."""
    import math
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd.camera import resolve_view, synthetic_camera
    from gaussian_lic_amd.synthetic import activate, lidar_scene, place_scene, random_scene, to_numpy
    cam = synthetic_camera(W, H, view)
    raw = (random_scene if kind == "random" else lidar_scene)(P, W, H, sh_degree=deg, seed=seed)
    pose = resolve_view(view)
    if pose is not None and pose[2]:
        raw = place_scene(raw, pose[0], pose[1])
    if sigma_scale != 1.0:
        raw["scaling"] = (raw["scaling"] + math.log(sigma_scale)).contiguous()
    return raw, to_numpy(activate(raw)), cam.as_dict(), cam


def clamp_masked_visible(sc, camd, radii):
    """Number of visible Gaussians (radii > 0) whose view-space t.x / t.z or t.y / t.z lies outside the lim window, i.e. whose cov2D Jacobian
    the reference clamps (forward.cu:91-94) and whose backward zeroes the x / y chain (backward.cu:177-178,248-249).  float64 estimate."""
    V = np.asarray(camd["view"], np.float64).reshape(4, 4).T      # stored transposed: element (r, c) at [4c + r]
    p = np.asarray(sc["means"], np.float64)
    t = p @ V[:3, :3].T + V[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        tx, ty = t[:, 0] / t[:, 2], t[:, 1] / t[:, 2]
    out = (tx < camd["limx_neg"]) | (tx > camd["limx_pos"]) | (ty < camd["limy_neg"]) | (ty > camd["limy_pos"])
    return int((out & (np.asarray(radii) > 0)).sum())


def rel_err(a, b):
    """max |a-b| relative to max |b| (the tolerance convention of BASELINE.md: 1e-4 of the tensor's max-abs)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close_flips(a, b, tol=1e-4, what="", max_flip_frac=2e-5, flip_bound=6e-3):
    """fp32 parity bar for outputs of the blend: every element within `tol` of the tensor's max-abs, EXCEPT threshold flips.
    The blend has hard cuts (alpha < 1/255 skips a Gaussian, forward.cu:437; T < 1e-4 stops a pixel, :439): two correct
    implementations whose exponent differs by one ulp decide differently for the rare (pixel, Gaussian) pair sitting exactly
    on a cut, which moves that pixel by up to alpha*T*c <= 1/255 of the colour scale.  At most `max_flip_frac` of the elements
    may exceed `tol`, and none may exceed `flip_bound` (1.5/255) of max-abs."""
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    if b.size == 0:
        return
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b) / scale
    bad = int((err > tol).sum())
    assert bad <= max(1, int(max_flip_frac * b.size)), f"{what}: {bad} of {b.size} elements off by more than {tol} (max {err.max():.3e})"
    assert err.max() <= flip_bound, f"{what}: max rel err {err.max():.3e} exceeds a single threshold contribution"
