"""Generates tests/golden/*.npz by running the REFERENCE's own kernels (oracle/_ref/libref_hip.so, built by
build_ref.py from /root/reference) on the MI355X over seeded synthetic scenes.  Run on the GPU box:

    python oracle/ref_build/make_golden.py gpurun_out/golden        # then copy the .npz into tests/golden/

The exact fp32 inputs (activated Gaussians, camera matrices, dL/dimage) are stored next to the outputs: the synthetic
generator uses vectorised exp/log/sigmoid whose last ulp differs between CPUs, so seeds alone do not reproduce them.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [  # name, kind, P, W, H, deg, seed, lambda_erank
    # P is a multiple of 256 on purpose: with a partial last block the reference's duplicateWithKeys lets the
    # out-of-range threads (clamped to idx = P-1, rasterizer_impl.cu:73-78) race pad keys over the last Gaussian's slots.
    ("random_1536_160x120_d3", "random", 1536, 160, 120, 3, 11, 0.0),
    ("lidar_1536_160x120_d3", "lidar", 1536, 160, 120, 3, 12, 0.0),
    ("random_1536_70x50_d1", "random", 1536, 70, 50, 1, 13, 0.0),
    ("random_1024_96x64_d0", "random", 1024, 96, 64, 0, 14, 0.0),
    ("random_1024_128x96_d3_erank", "random", 1024, 128, 96, 3, 15, 0.01),
]

# Round 5: general SE(3) camera poses (camera.SE3_POSES: pitch and roll of 20-40 deg, translation on all axes), a scale_modifier != 1 and a
# scene whose visible Gaussians are clamp-masked by the frustum limits (forward.cu:91-94).  name, kind, P, W, H, deg, seed, lambda_erank, extras
POSED_CASES = [
    ("random_1536_160x120_d3_se3a", "random", 1536, 160, 120, 3, 41, 0.0, dict(view="se3_a")),
    ("lidar_1536_160x120_d3_se3c_mod07", "lidar", 1536, 160, 120, 3, 42, 0.0, dict(view="se3_c", scale_modifier=0.7)),
    ("random_1024_128x96_d2_se3d_clamp", "random", 1024, 128, 96, 2, 43, 0.01, dict(view="se3_d", sigma_scale=2.5)),
]


def input_digest(sc, cam):
    h = hashlib.sha256()
    for k in ("means", "scales", "rots", "opac", "dc", "shs"):
        h.update(np.ascontiguousarray(sc[k], np.float32).tobytes())
    for k in ("view", "proj", "campos"):
        h.update(np.ascontiguousarray(cam[k], np.float32).tobytes())
    return h.hexdigest()


def rasterizer_golden(rk, outdir, cases):
    from conftest import clamp_masked_visible, make_scene
    from gaussian_lic_amd.synthetic import pixel_grad
    for case in cases:
        name, kind, P, W, H, deg, seed, le = case[:8]
        extra = case[8] if len(case) > 8 else {}
        view, sigma_scale, mod = extra.get("view"), extra.get("sigma_scale", 1.0), extra.get("scale_modifier", 1.0)
        raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed, view=view, sigma_scale=sigma_scale)
        dL = pixel_grad(H, W, seed=1).numpy()
        out = rk.run(sc, camd, dL, lambda_erank=le, scale_modifier=mod)
        nc = rk.run(sc, camd, None, no_color=True, scale_modifier=mod)
        keep = {k: v for k, v in out.items() if isinstance(v, np.ndarray)}
        keep["final_T_no_color"] = nc["final_T"]
        for k in ("means", "scales", "rots", "opac", "dc", "shs"):
            keep["in_" + k] = np.ascontiguousarray(sc[k], np.float32)
        for k in ("view", "proj", "campos"):
            keep["cam_" + k] = np.ascontiguousarray(camd[k], np.float32)
        keep["cam_scalars"] = np.array([camd[k] for k in ("tanfovx", "tanfovy", "limx_neg", "limx_pos", "limy_neg", "limy_pos")], np.float64)
        keep["in_dL_dpix"] = dL
        meta = dict(kind=kind, P=P, W=W, H=H, deg=deg, seed=seed, lambda_erank=le, R=out["R"], B32=out["B"],
                    R_no_color=nc["R"], B_no_color=nc["B"], digest=input_digest(sc, camd))
        if extra:
            meta.update(view=view, sigma_scale=sigma_scale, scale_modifier=mod, clamp_masked_visible=clamp_masked_visible(sc, camd, out["radii"]))
        np.savez_compressed(os.path.join(outdir, name + ".npz"), meta=np.array(repr(meta)), **keep)
        print(name, "R", out["R"], "B32", out["B"], "visible", int((out["radii"] > 0).sum()),
              "clamp-masked visible", meta.get("clamp_masked_visible"), flush=True)


def main(outdir):
    from oracle.ref_build.refkernels import RefKernels
    os.makedirs(outdir, exist_ok=True)
    rk = RefKernels()
    rasterizer_golden(rk, outdir, CASES + POSED_CASES)
    # Adam golden (adam.cu through ADAM::adamUpdate)
    rng = np.random.default_rng(0)
    N, M = 500, 45
    p, g = rng.standard_normal((N, M)).astype(np.float32), rng.standard_normal((N, M)).astype(np.float32)
    m, v = (0.1 * rng.standard_normal((N, M))).astype(np.float32), (0.01 * rng.random((N, M))).astype(np.float32)
    vis = rng.random(N) < 0.6
    p1, m1, v1 = p.copy(), m.copy(), v.copy()
    rk.adam(p1, g, m1, v1, vis, 2.5e-3)
    np.savez_compressed(os.path.join(outdir, "adam_500x45.npz"), p=p, g=g, m=m, v=v, vis=vis, p1=p1, m1=m1, v1=v1, lr=np.float32(2.5e-3))
    print("adam ok")
    ssim_golden(rk, outdir)
    knn_golden(rk, outdir)


def knn_points(P, seed):
    """LiDAR-like cloud (clustered depths, the shape GaussianModel::initialize feeds distCUDA2, gaussian.cpp:261) as exact fp32."""
    rng = np.random.default_rng(seed)
    d = rng.uniform(2.0, 40.0, P)
    slope = rng.uniform(-0.7, 0.7, (P, 2))          # only IEEE multiplications below: bit-reproducible on every platform
    pts = np.stack([d * slope[:, 0], d * slope[:, 1], d], 1)
    return np.ascontiguousarray(pts, np.float32)


KNN_CASES = [("knn_100096", 100096, 31), ("knn_1500", 1500, 32), ("knn_3", 3, 33)]   # the initialisation size; one partial box; < 4 points


def knn_golden(rk, outdir):
    """simple-knn golden vectors from the reference's own kernels (simple_knn.cu through oracle/ref_build/wrap_knn.hip).  The points
    are regenerated from the seed by knn_points() (numpy Generator streams are platform-independent; fp32 after the cast)."""
    for name, P, seed in KNN_CASES:
        pts = knn_points(P, seed)
        out = rk.knn(pts)
        np.savez_compressed(os.path.join(outdir, name + ".npz"), P=P, seed=seed, points_sha=np.array(hashlib.sha256(pts.tobytes()).hexdigest()),
                            mean_dist2=out)
        print(name, "mean", float(out[np.isfinite(out)].mean()) if np.isfinite(out).any() else None, flush=True)


SSIM_CASES = [("ssim_1x3x70x50", 1, 3, 50, 70, 21),      # ragged: neither side a multiple of the 32x32 block
              ("ssim_2x3x96x64", 2, 3, 64, 96, 22)]      # batch of two, block-aligned


def ssim_golden(rk, outdir):
    """fused-SSIM golden vectors from the reference's own kernels (ssim.cu through oracle/ref_build/wrap_ssim.hip)."""
    for name, B, CH, H, W, seed in SSIM_CASES:
        rng = np.random.default_rng(seed)
        img1 = rng.random((B, CH, H, W)).astype(np.float32)
        img2 = np.clip(img1 + 0.1 * rng.standard_normal((B, CH, H, W)), 0.0, 1.0).astype(np.float32)
        dL = rng.standard_normal((B, CH, H, W)).astype(np.float32)
        m, d1, d2, d3 = rk.ssim_forward(img1, img2)
        g = rk.ssim_backward(img1, img2, dL, d1, d2, d3)
        np.savez_compressed(os.path.join(outdir, name + ".npz"), img1=img1, img2=img2, dL_dmap=dL, ssim_map=m, dm_dmu1=d1, dm_dsigma1_sq=d2,
                            dm_dsigma12=d3, dL_dimg1=g)
        print(name, "mean ssim", float(m.mean()), flush=True)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden")
    if len(sys.argv) > 2 and sys.argv[2] in ("ssim", "knn", "posed"):   # only the fused-SSIM / simple-knn / posed-camera vectors
        from oracle.ref_build.refkernels import RefKernels
        os.makedirs(out, exist_ok=True)
        if sys.argv[2] == "posed":
            rasterizer_golden(RefKernels(), out, POSED_CASES)
        else:
            (ssim_golden if sys.argv[2] == "ssim" else knn_golden)(RefKernels(), out)
    else:
        main(out)
