"""ctypes front-end of oracle/_ref/libref_hip.so (the reference's own kernels built by build_ref.py).
Test infrastructure only (tests/ and the golden generator); needs a GPU."""
import ctypes
import os

import numpy as np

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_ref", "libref_hip.so")
LIB_FMA = os.path.join(os.path.dirname(LIB), "libref_hip_fma.so")   # the same sources under -ffp-contract=fast (build_ref.py): the yardstick build


def available(fma=False):
    return os.path.exists(LIB_FMA if fma else LIB)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class RefKernels:
    def __init__(self, fma=False):
        """fma=True loads the contracted build (hipcc's default -ffp-contract=fast, the analogue of nvcc's default --fmad=true)."""
        self.lib = ctypes.CDLL(LIB_FMA if fma else LIB)
        self.lib.ref_create.restype = ctypes.c_void_p
        self.lib.ref_destroy.argtypes = [ctypes.c_void_p]

    def run(self, sc, cam, dL_dpix=None, no_color=False, lambda_erank=0.0, scale_modifier=1.0):
        """sc: activated numpy scene (synthetic.to_numpy(activate(raw))); cam: Camera.as_dict().  Returns dict of
        every stage boundary the reference's forward/backward produce."""
        L = self.lib
        f32 = lambda x: np.ascontiguousarray(x, np.float32)
        P = sc["means"].shape[0]
        M = 0 if sc["shs"] is None or sc["shs"].size == 0 else sc["shs"].shape[1]
        W, H = cam["W"], cam["H"]
        T = ((W + 15) // 16) * ((H + 15) // 16)
        means, dc, opac, scales, rots = f32(sc["means"]), f32(sc["dc"]), f32(sc["opac"]).reshape(-1), f32(sc["scales"]), f32(sc["rots"])
        shs = f32(sc["shs"]) if M > 0 else None
        view, proj, campos = f32(cam["view"]), f32(cam["proj"]), f32(cam["campos"])
        out = dict(color=np.zeros((3, H, W), np.float32), final_T=np.zeros((H, W), np.float32), radii=np.zeros(P, np.int32))
        R, B = ctypes.c_int(0), ctypes.c_int(0)
        ctx = ctypes.c_void_p(L.ref_create())
        cf = ctypes.c_float
        try:
            if scale_modifier != 1.0:
                L.ref_set_scale_modifier.argtypes = [ctypes.c_void_p, ctypes.c_float]
                L.ref_set_scale_modifier(ctx, cf(scale_modifier))
            rc = L.ref_forward(ctx, P, int(sc["D"]), M, W, H, _p(means), _p(dc), _p(shs), _p(opac), _p(scales), _p(rots), _p(view),
                               _p(proj), _p(campos), cf(cam["tanfovx"]), cf(cam["tanfovy"]), cf(cam["limx_neg"]), cf(cam["limx_pos"]),
                               cf(cam["limy_neg"]), cf(cam["limy_pos"]), int(no_color), _p(out["color"]), _p(out["final_T"]),
                               _p(out["radii"]), ctypes.byref(R), ctypes.byref(B))
            assert rc == 0
            out["R"], out["B"] = R.value, B.value
            ex = dict(tiles_touched=np.zeros(P, np.uint32), means2D=np.zeros((P, 2), np.float32), depths=np.zeros(P, np.float32),
                      conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32), clamped=np.zeros((P, 3), np.uint8),
                      keys=np.zeros(max(R.value, 1), np.uint64), point_list=np.zeros(max(R.value, 1), np.uint32),
                      ranges=np.zeros((T, 2), np.uint32), n_contrib=np.zeros((H, W), np.uint32), max_contrib=np.zeros(T, np.uint32))
            rc = L.ref_export(ctx, *[_p(ex[k]) for k in ("tiles_touched", "means2D", "depths", "conic_opacity", "rgb", "clamped", "keys",
                                                         "point_list", "ranges", "n_contrib", "max_contrib")])
            assert rc == 0
            ex["keys"], ex["point_list"] = ex["keys"][:R.value], ex["point_list"][:R.value]
            out.update(ex)
            if dL_dpix is not None and not no_color:
                z = lambda *s: np.zeros(s, np.float32)
                g = dict(dL_dmean2D=z(P, 3), dL_dconic=z(P, 4), dL_dopacity=z(P, 1), dL_dcolor=z(P, 3), dL_dmean3D=z(P, 3),
                         dL_dcov3D=z(P, 6), dL_ddc=z(P, 1, 3), dL_dsh=z(P, max(M, 1), 3), dL_dscale=z(P, 3), dL_drot=z(P, 4))
                rc = L.ref_backward(ctx, _p(f32(dL_dpix)), cf(lambda_erank), *[_p(g[k]) for k in (
                    "dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_dmean3D", "dL_dcov3D", "dL_ddc", "dL_dsh", "dL_dscale", "dL_drot")])
                assert rc == 0
                if M == 0:
                    g["dL_dsh"] = z(P, 0, 3)
                out.update(g)
        finally:
            L.ref_destroy(ctx)
        return out

    def adam(self, param, grad, m, v, visible, lr, b1=0.9, b2=0.999, eps=1e-15):
        N = param.shape[0]
        M = param.size // max(N, 1)
        cf = ctypes.c_float
        vis = np.ascontiguousarray(visible.astype(np.uint8))
        rc = self.lib.ref_adam(_p(param), _p(np.ascontiguousarray(grad, np.float32)), _p(m), _p(v), _p(vis), cf(lr), cf(b1), cf(b2), cf(eps),
                               ctypes.c_uint(N), ctypes.c_uint(M))
        assert rc == 0

    def ssim_forward(self, img1, img2, C1=0.01 ** 2, C2=0.03 ** 2):
        """The reference's fusedssimCUDA (ssim.cu:186-285) with train = true.  img [B,CH,H,W] -> (map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)."""
        a, b = np.ascontiguousarray(img1, np.float32), np.ascontiguousarray(img2, np.float32)
        B, CH, H, W = a.shape
        outs = [np.zeros_like(a) for _ in range(4)]
        cf = ctypes.c_float
        rc = self.lib.ref_ssim_forward(B, CH, H, W, cf(C1), cf(C2), _p(a), _p(b), *[_p(o) for o in outs])
        assert rc == 0
        return tuple(outs)

    def ssim_backward(self, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, C1=0.01 ** 2, C2=0.03 ** 2):
        """The reference's fusedssim_backwardCUDA (ssim.cu:287-365) -> dL_dimg1."""
        arrs = [np.ascontiguousarray(x, np.float32) for x in (img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)]
        B, CH, H, W = arrs[0].shape
        out = np.zeros_like(arrs[0])
        cf = ctypes.c_float
        rc = self.lib.ref_ssim_backward(B, CH, H, W, cf(C1), cf(C2), *[_p(x) for x in arrs], _p(out))
        assert rc == 0
        return out

    def knn(self, points):
        """The reference's SimpleKNN::knn (simple_knn.cu:185-221, compiled through wrap_knn.hip): points [P,3] -> mean of the three
        smallest squared distances to the other points, [P]."""
        pts = np.ascontiguousarray(points, np.float32)
        out = np.zeros(pts.shape[0], np.float32)
        rc = self.lib.ref_knn(int(pts.shape[0]), _p(pts), _p(out))
        assert rc == 0
        return out
