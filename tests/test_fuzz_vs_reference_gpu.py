"""-m gpu: a seeded subset of tests/fuzz_vs_reference.py inside the collected suite — 48 random small scenes (256 .. 76 800 Gaussians, image
sizes that are and are not multiples of the 16-pixel tile down to 33x17, SH degree 0..3, both scene kinds), HIP path against the REFERENCE's
own kernels (oracle/_ref/libref_hip.so: forward.cu / backward.cu / rasterizer_impl.cu compiled for gfx950) on the same inputs.

Bars, default (strict) arithmetic — no allowance anywhere:
  radii, tiles_touched, per-tile lists, ranges                bit-exact   (rasterizer_impl.cu:59-231, forward.cu:232-319)
  means2D / depth / conic / opacity / SH colour               bit-exact
  image, final_T, n_contrib                                   bit-exact   (forward.cu:424-445)
  the nine gradients                                          zero elements beyond 1e-4 of the tensor's max-abs (backward.cu:379-597)
The fast arithmetic (opt-in) is run on the same scenes: integer stages exact, the image within threshold flips (counts are printed).
P is a multiple of 256 in every case: with a partial last block the reference's duplicateWithKeys lets out-of-range threads write pad keys
over the last Gaussian's slots (rasterizer_impl.cu:73-131) — other P are held to the C oracle (test_parity_gpu.py).
The script form (python tests/fuzz_vs_reference.py 400) runs the long sequence; its log is profiles/r03_fuzz_vs_reference.txt."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_CASES = 48


def _cases():
    rng = np.random.default_rng(20260926)
    out = []
    # the first eight pin the shapes the review asked for by name; the rest are drawn like the script draws them
    pinned = [(33, 17), (333, 97), (100, 90), (640, 360), (160, 48), (33, 180), (320, 17), (64, 97)]
    for i in range(N_CASES):
        kind = "random" if rng.random() < 0.7 else "lidar"
        P = 256 * int(rng.choice([1, 2, 7, 40, 100, 300]))
        W = int(rng.choice([33, 64, 100, 160, 320, 333, 640]))
        H = int(rng.choice([17, 48, 90, 97, 180, 360]))
        if i < len(pinned):
            W, H = pinned[i]
        deg = int(rng.integers(0, 4))
        seed = int(rng.integers(0, 10 ** 6))
        out.append((kind, P, W, H, deg, seed))
    return out


CASES = _cases()


def test_case_list_covers_the_awkward_shapes():
    """(the generator itself: a changed numpy stream must not silently drop the odd sizes)"""
    assert any(W == 33 and H == 17 for _, _, W, H, _, _ in CASES)
    assert sum(1 for _, _, W, H, _, _ in CASES if W % 16 or H % 16) >= 24
    assert {d for *_, d, _ in CASES} == {0, 1, 2, 3} and {k for k, *_ in CASES} == {"random", "lidar"}


@pytest.mark.parametrize("case", range(N_CASES))
def test_fuzz_scene_is_bit_identical_to_the_reference_kernels(case):
    from oracle.ref_build import refkernels
    if not refkernels.available():
        pytest.skip("oracle/_ref/libref_hip.so not built")
    from refcompare import GRADS, compare, summarize
    from refcompare import assert_path
    kind, P, W, H, deg, seed = CASES[case]
    # round 6: the grouping of the instances is FORCED per case and verified (never `auto`) — a third of the scenes on the stable radix sort with the
    # rows as generated, a third on the block-aggregated atomics with the rows as generated, a third on the atomics with the rows permuted into Morton
    # order + tie_rank (the configuration bench.py times)
    binning, morton = (("radix", False), ("atomic", False), ("atomic", True))[case % 3]
    res = compare(kind, P, W, H, deg, seed, binning=binning, morton=morton)
    print("\n" + summarize(res))
    assert_path(res)
    for mode in ("strict", "fast"):
        st = res[mode]
        assert st["radii_mismatch"] == 0 and st["tiles_touched_mismatch"] == 0, (mode, st["radii_mismatch"], st["tiles_touched_mismatch"])
        assert st["R"] == res["ref"]["R"]
        assert st["point_list_equal"] and st["ranges_equal"], mode
        assert st["means2D_bit_equal"] and st["depths_bit_equal"] and st["conic_opacity_bit_equal"] and st["rgb_bit_equal"], mode
    st = res["strict"]
    assert st["color"]["bit_equal"] and st["final_T"]["bit_equal"] and st["n_contrib_mismatch"] == 0, (st["color"], st["final_T"], st["n_contrib_mismatch"])
    for k in GRADS:
        # (zero elements over 1e-4; an element only counts where fp32 can resolve it and when it is over the bar against every one of up to nine runs of the reference's atomics:
        # refcompare.conditioning_probe — none has occurred in these scenes, the rule is the posed suite's)
        assert st[k]["over"] == st[k].get("over_excused", 0), (k, st[k])
    st = res["fast"]   # opt-in arithmetic: threshold flips only — a handful of elements (printed above), the image never off by more than a contribution
    assert st["color"]["over"] <= max(8, 1e-4 * st["color"]["n"]) and st["color"]["max_rel"] < 5e-2, st["color"]
    assert st["n_contrib_mismatch"] <= max(8, 1e-4 * st["pixels"])


def test_rectangle_tie_scene_matches_the_reference_kernels():
    """Fuzz scene 845806 (case 385 of `tests/fuzz_vs_reference.py 600 8001`, round 6): one Gaussian whose mean, 215.99998474, plus radius 25 plus BLOCK_X
    16 is a rounding tie — the reference's getRect (`p.x + max_radius + BLOCK_X - 1`, auxiliary.h:52-53) reaches tile column 15, a restatement that adds
    15 at once does not, and the Gaussian lost a tile it blends into (tiles_touched 9 against the reference kernels' 10).  The whole scene, both forced
    paths, against the reference's own kernels."""
    from oracle.ref_build import refkernels
    if not refkernels.available():
        pytest.skip("oracle/_ref/libref_hip.so not built")
    from refcompare import GRADS, assert_path, compare, summarize
    for binning, morton in (("radix", False), ("atomic", True)):
        res = compare("random", 25600, 320, 180, 3, 845806, binning=binning, morton=morton)
        print("\n" + summarize(res))
        assert_path(res)
        for mode in ("strict", "fast"):
            st = res[mode]
            assert st["radii_mismatch"] == 0 and st["tiles_touched_mismatch"] == 0, (mode, st["radii_mismatch"], st["tiles_touched_mismatch"])
            assert st["R"] == res["ref"]["R"] and st["point_list_equal"] and st["ranges_equal"], mode
        st = res["strict"]
        assert st["color"]["bit_equal"] and st["final_T"]["bit_equal"] and st["n_contrib_mismatch"] == 0
        for k in GRADS:
            assert st[k]["over"] == st[k].get("over_excused", 0), (k, st[k])
