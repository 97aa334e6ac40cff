"""CPU: a list-order invariant of the binning (and what `tools/experiments/dead_cutoff.patch` relied on; measured and rejected, profiles/r04v).

The blend backward skips every 64-entry bucket of a tile that lies behind the tile's last contributor (backward.cu:428) and tells the
per-Gaussian backward which instances those are.  Today that is one flag byte per instance, scattered; the experiment replaces it by ONE key per
tile: a tile's list is sorted by (depth bits, Gaussian id) (rasterizer_impl.cu:86-128, 419-424: a stable radix sort of tile << 32 | depth over
instances emitted in id order), so "position >= first dead position" is the same set as "(depth bits, id) >= key of the first dead entry".
This test holds the oracle's lists to that equivalence, ties in depth included."""
import numpy as np
import pytest

from conftest import make_scene

BUCKET = 64


def _check(fwd):
    keys, plist = fwd["bins"]["keys"], fwd["bins"]["point_list"].astype(np.int64)
    ranges, mc = fwd["bins"]["ranges"].astype(np.int64), fwd["max_contrib"].astype(np.int64)
    depth_bits = (keys & np.uint64(0xffffffff)).astype(np.int64)
    n_dead = n_live = n_ties = 0
    for t in range(ranges.shape[0]):
        lo, hi = ranges[t]
        if hi <= lo:
            continue
        first_dead = lo + BUCKET * ((mc[t] + BUCKET - 1) // BUCKET)      # first list position of the tile's first dead bucket
        d, g = depth_bits[lo:hi], plist[lo:hi]
        assert np.all((d[1:] > d[:-1]) | ((d[1:] == d[:-1]) & (g[1:] > g[:-1]))), f"tile {t}: list not strictly sorted by (depth bits, id)"
        n_ties += int((d[1:] == d[:-1]).sum())
        by_position = np.arange(lo, hi) >= first_dead
        if first_dead < hi:
            cd, cg = depth_bits[first_dead], plist[first_dead]
            by_key = (d > cd) | ((d == cd) & (g >= cg))
        else:
            by_key = np.zeros(hi - lo, bool)                             # no dead bucket: the cut-off stays "all ones"
        assert np.array_equal(by_position, by_key), f"tile {t}"
        n_dead += int(by_position.sum()); n_live += int((~by_position).sum())
    return n_live, n_dead, n_ties


@pytest.mark.parametrize("kind,P,seed", [("random", 6000, 0), ("lidar", 60000, 1)])
def test_dead_instances_are_those_behind_the_tiles_cutoff_key(oracle32, kind, P, seed):
    W, H = 160, 96
    raw, sc, camd, cam = make_scene(kind, P, W, H, 1, seed)
    if kind == "lidar":   # (the LiDAR-seeded surface is faint: make it opaque enough for pixels to saturate in front of the ends of their lists)
        sc = dict(sc); sc["opac"] = np.full_like(sc["opac"], 0.9)
    fwd = oracle32.forward(sc, camd)
    n_live, n_dead, _ = _check(fwd)
    assert n_live > 0 and n_dead > 0, "the scene must have live and dead buckets for the test to say anything"


def test_cutoff_key_with_equal_depths(oracle32):
    """Several Gaussians at exactly the same depth (LiDAR returns on a plane facing the camera): the id half of the key decides."""
    W, H = 96, 64
    raw, sc, camd, cam = make_scene("random", 4000, W, H, 0, 5)
    view = np.asarray(camd["view"], np.float64).reshape(4, 4)   # column-major as the reference's glm matrices: p_view = p . view (row vector)
    m = sc["means"].astype(np.float64)
    pv = np.concatenate([m, np.ones((m.shape[0], 1))], 1) @ view
    # quantise the view-space depth to a few planes and move the points there along the view's z axis (rows of the rotation part)
    zq = np.round(pv[:, 2] * 2.0) / 2.0
    shift = (zq - pv[:, 2])[:, None] * np.linalg.inv(view[:3, :3])[2][None, :]
    sc = dict(sc); sc["means"] = (m + shift).astype(np.float32)
    fwd = oracle32.forward(sc, camd)
    n_live, n_dead, n_ties = _check(fwd)
    assert n_ties > 0, "no equal-depth neighbours: the construction did not produce ties"
    assert n_live > 0
