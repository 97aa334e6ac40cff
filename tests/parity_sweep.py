"""Randomised HIP-vs-oracle parity sweep (run on the MI355X: `python tests/parity_sweep.py [n_cases] [seed]`).  Draws scene kind, size,
image shape (ragged on purpose), SH degree and seeds; checks the integer stages bit-exactly and images / gradients with the bars of
tests/test_parity_gpu.py.  A one-off confidence run, not part of the test suite (the oracle needs seconds per case)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # this file lives in tests/: the oracle is test infrastructure


def main(n_cases, seed):
    from conftest import assert_close_flips, make_scene
    from gpu_helpers import hip_backward, hip_forward, npy
    from gaussian_lic_amd.synthetic import pixel_grad
    from oracle.oracle import Oracle, build
    build()
    orc = Oracle(np.float32)
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(n_cases):
        kind = "random" if rng.random() < 0.7 else "lidar"
        P = int(rng.choice([1, 63, 64, 65, 257, 1000, 4097, 20000, 60000]))
        W = int(rng.integers(17, 400)); H = int(rng.integers(17, 300))
        deg = int(rng.integers(0, 4)); s = int(rng.integers(0, 10000))
        raw, sc, camd, cam = make_scene(kind, P, W, H, deg, s)
        ref = orc.forward(sc, camd)
        got = hip_forward(raw, cam, export=("tiles_touched", "sorted_keys", "point_list", "ranges"))
        tag = f"case {c}: {kind} P={P} {W}x{H} deg={deg} seed={s} R={got['R']}"
        try:
            d = got["dbg"]
            assert got["R"] == ref["num_rendered"], "R"
            np.testing.assert_array_equal(npy(got["radii"]), ref["pre"]["radii"])
            np.testing.assert_array_equal(npy(d["tiles_touched"]).astype(np.uint32), ref["pre"]["tiles_touched"].astype(np.uint32))
            np.testing.assert_array_equal(npy(d["point_list"])[:got["R"]].astype(np.uint32), ref["bins"]["point_list"][:got["R"]].astype(np.uint32))
            np.testing.assert_array_equal(npy(d["ranges"]).reshape(-1, 2).astype(np.uint32), ref["bins"]["ranges"].astype(np.uint32))
            assert_close_flips(npy(got["color"]), ref["color"], 1e-4, "color")
            dL = pixel_grad(H, W, seed=1)
            g = hip_backward(got, dL)
            rg = orc.backward(sc, camd, ref, dL.numpy())
            for k in ("dL_dmean3D", "dL_dopacity", "dL_ddc", "dL_dsh", "dL_dscale"):
                if rg[k].size:
                    assert_close_flips(g[k].reshape(rg[k].shape), rg[k], 1e-4, k, flip_bound=2e-2)
            print("ok  ", tag, flush=True)
        except AssertionError as ex:
            bad += 1
            print("FAIL", tag, str(ex)[:200], flush=True)
    print("sweep done:", n_cases - bad, "ok,", bad, "failed")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
