"""-m gpu: the two ways the forward groups the (Gaussian, tile) instances by tile — the stable radix sort on the tile id and the block-aggregated
atomics of csrc/tile_bin.hip — give the reference's lists bit for bit (CPU oracle: a stable sort of 64-bit (tile << 32 | depth) keys of an
index-ordered emission, rasterizer_impl.cu:395-424), depth TIES included: the atomic path delivers a tile's instances in arrival order, so the
per-tile sort orders equal depths by a second key.  The scenes have their depths quantised (thousands of ties per tile) and, in one case,
lists beyond what one wave and what LDS sorts."""
import math

import numpy as np
import pytest
import torch

from conftest import make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture
def binning_mode():
    from gaussian_lic_amd import _lib
    old = _lib.set_binning_mode("auto")
    yield _lib.set_binning_mode
    _lib.set_binning_mode(old)


def _quantise(raw, sc, quantum):
    """identity camera: depth = z -> many exact depth ties (x, y scaled along so that the projections stay where they were)"""
    z = raw["xyz"][:, 2]
    zq = torch.where(z > 0.3, (z / quantum).round().clamp_min(1.0) * quantum, z)
    raw["xyz"] = torch.stack([raw["xyz"][:, 0] * zq / z, raw["xyz"][:, 1] * zq / z, zq], 1).contiguous()
    sc["means"] = raw["xyz"].numpy().copy()


CASES = [
    ("random", 40000, 320, 192, 0.5, 1.0, 3),      # ties, lists of a few hundred
    ("random", 40000, 320, 192, 0.0, 1.0, 4),      # no quantisation (the odd natural tie only)
    ("random", 60000, 160, 96, 0.5, 3.0, 7),       # ties in lists of thousands (the workgroup kernel: LDS and global-scratch paths)
    ("lidar", 30000, 333, 190, 0.25, 1.0, 5),      # ragged image
    ("random", 60000, 3840, 2400, 0.0, 1.0, 9),    # 36 000 tiles: the largest LDS histogram the atomic path takes (144 KB), 1024-thread workgroups
    ("random", 60000, 4096, 2400, 0.0, 1.0, 10),   # 38 400 tiles: beyond it — "atomic" sorts as well
]


@pytest.mark.parametrize("kind,P,W,H,quantum,sigma,seed", CASES)
def test_both_binning_paths_give_the_reference_lists(oracle32, binning_mode, kind, P, W, H, quantum, sigma, seed):
    from gpu_helpers import hip_forward, npy
    raw, sc, camd, cam = make_scene(kind, P, W, H, 3, seed, sigma_scale=sigma)
    if quantum > 0:
        _quantise(raw, sc, quantum)
    ref = oracle32.forward(sc, camd)
    if quantum > 0:
        k = ref["bins"]["keys"]
        assert int((k[1:] == k[:-1]).sum()) > ref["num_rendered"] // 4      # (the keys tie)
    if sigma > 1:
        r = ref["bins"]["ranges"].astype(np.int64)
        assert (r[:, 1] - r[:, 0]).max() > 4096
    got = {}
    for mode in ("radix", "atomic"):
        binning_mode(mode)
        f = hip_forward(raw, cam, export=("sorted_keys", "point_list", "ranges"))
        assert f["R"] == ref["num_rendered"]
        d = f["dbg"]
        np.testing.assert_array_equal(npy(d["ranges"]).astype(np.uint32), ref["bins"]["ranges"], err_msg=mode)
        np.testing.assert_array_equal(npy(d["sorted_keys"]).view(np.uint64), ref["bins"]["keys"], err_msg=mode)
        np.testing.assert_array_equal(npy(d["point_list"]).astype(np.uint32), ref["bins"]["point_list"], err_msg=mode)
        got[mode] = (npy(f["color"]), npy(f["final_T"]))
    assert np.array_equal(got["radix"][0], got["atomic"][0]) and np.array_equal(got["radix"][1], got["atomic"][1])


def test_backward_is_the_same_on_both_paths(binning_mode):
    """the emission slots (where the blend backward leaves an instance's partial row, and where the per-Gaussian backward finds it) travel through
    either grouping: every gradient bit-identical"""
    from gpu_helpers import hip_backward, hip_forward
    from gaussian_lic_amd.synthetic import pixel_grad
    raw, sc, camd, cam = make_scene("random", 50000, 480, 270, 3, 11)
    _quantise(raw, sc, 0.5)
    dL = pixel_grad(270, 480, seed=2)
    g = {}
    for mode in ("radix", "atomic"):
        binning_mode(mode)
        g[mode] = hip_backward(hip_forward(raw, cam), dL)
    for k in g["radix"]:
        assert np.array_equal(g["radix"][k], g["atomic"][k]), k


def test_auto_mode_follows_the_row_order(binning_mode):
    """auto: rows in Morton order -> the atomic path; rows in random order -> one probing forward on it, then the radix sort"""
    from gaussian_lic_amd import _lib, trainer
    from gpu_helpers import hip_forward
    P, W, H = 300000, 1920, 1080
    raw, sc, camd, cam = make_scene("random", P, W, H, 3, 3)
    order = trainer.morton_order(raw["xyz"])
    raw_m = {k: (v[order].contiguous() if torch.is_tensor(v) and v.shape[:1] == (P,) else v) for k, v in raw.items()}

    def launches(r, n):
        _lib.profile_enable(True)
        _lib.profile_reset()
        outs = [hip_forward(r, cam) for _ in range(n)]
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        return prof.get("tile_bin", (0, 0))[1], prof.get("sort_scatter", (0, 0))[1], outs

    binning_mode("auto")
    nb, ns, a = launches(raw_m, 3)
    assert nb == 3 and ns == 0, (nb, ns)
    binning_mode("auto")          # (forgets what was measured)
    nb, ns, b = launches(raw, 4)
    assert nb == 1 and ns == 2 * 3, (nb, ns)
    assert torch.equal(b[0]["color"], b[3]["color"])     # the probing forward and the radix ones: the same image
    assert a[0]["R"] == b[0]["R"]
