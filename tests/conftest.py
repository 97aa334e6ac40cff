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


def make_scene(kind, P, W, H, deg, seed=0, view=None):
    """(raw torch params, activated numpy dict for the oracle, camera dict, Camera)"""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import activate, lidar_scene, random_scene, to_numpy
    cam = synthetic_camera(W, H, view)
    raw = (random_scene if kind == "random" else lidar_scene)(P, W, H, sh_degree=deg, seed=seed)
    return raw, to_numpy(activate(raw)), cam.as_dict(), cam


def rel_err(a, b):
    """max |a-b| relative to max |b| (the tolerance convention of BASELINE.md: 1e-4 of the tensor's max-abs)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
