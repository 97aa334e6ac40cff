#!/usr/bin/env python
"""Randomised shapes for the operator-level comparisons against the reference's OWN kernels (oracle/_ref/libref_hip.so) — the fixed-shape tests of
tests/test_vs_reference_kernels_gpu.py run on many more shapes: fused-SSIM forward / backward / fused loss gradient (bit for bit), simple-knn
(1e-5).  Outside the driver's suite (minutes of GPU); the result of a run is committed under profiles/.
    python tests/fuzz_ops_vs_reference.py [n_ssim = 300] [n_knn = 60] [seed = 1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n_ssim = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n_knn = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    import test_vs_reference_kernels_gpu as T
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(n_ssim):
        B, CH = int(rng.choice([1, 1, 2, 3])), int(rng.choice([1, 3, 3, 4]))
        H = int(rng.choice([1, 2, 5, 11, 31, 32, 33, 63, 64, 65, 97, 180, 270, int(rng.integers(1, 400))]))
        W = int(rng.choice([1, 3, 10, 11, 31, 32, 33, 42, 64, 95, 96, 129, 320, int(rng.integers(1, 500))]))
        try:
            T.test_hip_ssim_matches_reference_kernels(B, CH, H, W)
            ok = True
        except AssertionError as e:
            ok = False
            print(str(e)[:300])
        print(f"ssim {i:3d} B={B} CH={CH} {H}x{W}: {'OK' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    for H, W in [(int(rng.integers(1, 300)), int(rng.integers(1, 400))) for _ in range(max(n_ssim // 10, 1))]:
        try:
            T.test_fused_loss_gradient_is_the_reference_chain_bit_for_bit(H, W)
            ok = True
        except AssertionError as e:
            ok = False
            print(str(e)[:300])
        print(f"fused loss {H}x{W}: {'OK' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    for i in range(n_knn):
        P = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1000, int(rng.integers(1, 5000)), int(rng.integers(5000, 300000))]))
        try:
            if P < 3:   # fewer than three neighbours: both sides return the same non-finite mean (FLT_MAX sums) — compared as bits
                import torch
                from gaussian_lic_amd import knn
                from oracle.ref_build.make_golden import knn_points
                pts = knn_points(P, int(rng.integers(0, 10 ** 6)))
                ref = T._ref().knn(pts)
                got = knn.distCUDA2(torch.from_numpy(pts).to("cuda:0")).cpu().numpy()
                assert np.array_equal(np.asarray(got, np.float32).view(np.uint32), np.asarray(ref, np.float32).view(np.uint32)), (got, ref)
            else:
                T.test_hip_knn_matches_reference_kernels(P, int(rng.integers(0, 10 ** 6)))
            ok = True
        except AssertionError as e:
            ok = False
            print(str(e)[:300])
        except Exception as e:   # (the reference's own kernel may reject degenerate sizes)
            ok = None
            print(f"knn P={P}: {type(e).__name__}: {str(e)[:200]}")
        print(f"knn {i:3d} P={P}: {'OK' if ok else ('MISMATCH' if ok is False else 'SKIPPED')}", flush=True)
        bad += 1 if ok is False else 0
    # Adam (adam.cu:9-38 through the checker's ref_adam) on random row counts / widths / visibility patterns, values with zeros, tiny and huge
    # magnitudes: parameters and both moments BIT FOR BIT, for the one-group entry and the grouped one (all rows of a group in one launch)
    import torch
    from gaussian_lic_amd import optim
    rk = T._ref()
    n_adam = max(n_ssim // 3, 1)
    for i in range(n_adam):
        N = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 256, 1000, int(rng.integers(1, 200000))]))
        M = int(rng.choice([1, 3, 4, 45, int(rng.integers(1, 60))]))
        scale = float(rng.choice([1e-12, 1e-4, 1.0, 1e4]))
        p_ = (scale * rng.standard_normal((N, M))).astype(np.float32); g_ = (scale * rng.standard_normal((N, M))).astype(np.float32)
        g_[rng.random((N, M)) < 0.1] = 0.0
        m_ = (0.1 * scale * rng.standard_normal((N, M))).astype(np.float32); v_ = (0.01 * scale * scale * rng.random((N, M))).astype(np.float32)
        vis = rng.random(N) < float(rng.choice([0.0, 0.3, 0.7, 1.0]))
        lr = float(rng.choice([1.6e-4, 5e-2, 1e-3]))
        rp, rm, rv = p_.copy(), m_.copy(), v_.copy()
        rk.adam(rp, g_, rm, rv, vis, lr)
        tp, tg, tm, tv = (torch.from_numpy(x.copy()).to("cuda:0") for x in (p_, g_, m_, v_))
        optim.adam_update(tp, tg, tm, tv, torch.from_numpy(vis).to("cuda:0"), lr, 0.9, 0.999, 1e-15, N, M)
        ok = all(np.array_equal(a.cpu().numpy().view(np.uint32), b.view(np.uint32)) for a, b in ((tp, rp), (tm, rm), (tv, rv)))
        print(f"adam {i:3d} N={N} M={M} scale={scale:g} visible={int(vis.sum())}: {'OK' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    print(f"{n_ssim} SSIM shapes, {max(n_ssim // 10, 1)} fused-loss shapes, {n_knn} kNN sizes, {n_adam} Adam shapes: {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
