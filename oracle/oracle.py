"""ctypes front-end of the CPU oracle (oracle/gs_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing under
gaussian-lic_amd/ (the product) does.  It never reads /root/reference.

Arrays are numpy, fp32 (liboracle_f32.so) or fp64 (liboracle_f64.so, finite-difference checks only).
Layouts are the reference's: view/proj matrices as float[16] with element (r,c) at [4c+r].
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile both oracle libraries with oracle/Makefile (gcc, a few seconds)."""
    libs = [os.path.join(_HERE, n) for n in ("liboracle_f32.so", "liboracle_f64.so")]
    src = os.path.join(_HERE, "gs_oracle.c")
    stale = force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "all"], check=True, capture_output=not force)
    return libs


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Oracle:
    def __init__(self, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        name = "liboracle_f32.so" if self.dtype == np.float32 else "liboracle_f64.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        self.lib = ctypes.CDLL(path)
        self.real = ctypes.c_float if self.dtype == np.float32 else ctypes.c_double
        assert self.lib.orc_real_bytes() == self.dtype.itemsize
        self.lib.orc_binning.restype = ctypes.c_int64
        self.lib.orc_logf.restype = ctypes.c_float
        self.lib.orc_logf.argtypes = [ctypes.c_float]
        self.lib.orc_higher_msb.restype = ctypes.c_uint32
        self.lib.orc_max_threads.restype = ctypes.c_int

    # ------------------------------------------------------------------ helpers
    def a(self, x, shape=None):
        x = np.ascontiguousarray(np.asarray(x, dtype=self.dtype))
        if shape is not None:
            x = x.reshape(shape)
        return x

    def set_threads(self, n):
        self.lib.orc_set_threads(int(n))

    def max_threads(self):
        return int(self.lib.orc_max_threads())

    def logf(self, x):
        return float(self.lib.orc_logf(ctypes.c_float(x)))

    def get_rect(self, px, py, radius, gx, gy):
        """getRect of auxiliary.h:46-56 for one mean (pixels) and radius: (x0, y0, x1, y1) in tiles."""
        out = (ctypes.c_int * 4)()
        real = self.real
        self.lib.orc_get_rect.argtypes = [real, real, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        self.lib.orc_get_rect.restype = None
        self.lib.orc_get_rect(real(px), real(py), int(radius), int(gx), int(gy), out)
        return tuple(out)

    # ------------------------------------------------------------------ stages
    def preprocess(self, sc, cam, no_color=False):
        """sc: dict(means,scales,rots,opac,dc,shs,D) activated parameters; cam: dict(W,H,view,proj,campos,tanfovx,...)"""
        P = sc["means"].shape[0]
        M = 0 if sc["shs"] is None or sc["shs"].size == 0 else sc["shs"].shape[1]
        r = self.real
        out = dict(
            radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), self.dtype), depths=np.zeros(P, self.dtype),
            cov3D=np.zeros((P, 6), self.dtype), conic_opacity=np.zeros((P, 4), self.dtype),
            rgb=np.zeros((P, 3), self.dtype), clamped=np.zeros((P, 3), np.uint8), tiles_touched=np.zeros(P, np.uint32))
        means, scales, rots = self.a(sc["means"]), self.a(sc["scales"]), self.a(sc["rots"])
        opac, dc = self.a(sc["opac"]).reshape(-1), self.a(sc["dc"])
        shs = self.a(sc["shs"]) if M > 0 else None
        view, proj, campos = self.a(cam["view"]), self.a(cam["proj"]), self.a(cam["campos"])
        self.lib.orc_preprocess(
            ctypes.c_int(P), ctypes.c_int(int(sc["D"])), ctypes.c_int(M), _ptr(means), _ptr(scales),
            r(sc.get("scale_modifier", 1.0)), _ptr(rots), _ptr(opac), _ptr(dc), _ptr(shs), _ptr(view), _ptr(proj),
            _ptr(campos), ctypes.c_int(cam["W"]), ctypes.c_int(cam["H"]), r(cam["tanfovx"]), r(cam["tanfovy"]),
            r(cam["limx_neg"]), r(cam["limx_pos"]), r(cam["limy_neg"]), r(cam["limy_pos"]), ctypes.c_int(int(no_color)),
            _ptr(out["radii"]), _ptr(out["means2D"]), _ptr(out["depths"]), _ptr(out["cov3D"]),
            _ptr(out["conic_opacity"]), _ptr(out["rgb"]), _ptr(out["clamped"]), _ptr(out["tiles_touched"]))
        return out

    def binning(self, pre, W, H):
        P = pre["radii"].shape[0]
        T = ((W + 15) // 16) * ((H + 15) // 16)
        Rmax = int(pre["tiles_touched"].astype(np.int64).sum())
        keys = np.zeros(max(Rmax, 1), np.uint64)
        plist = np.zeros(max(Rmax, 1), np.uint32)
        ranges = np.zeros((T, 2), np.uint32)
        R = self.lib.orc_binning(
            ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _ptr(pre["radii"]), _ptr(pre["means2D"]),
            _ptr(pre["depths"]), _ptr(pre["conic_opacity"]), _ptr(pre["tiles_touched"]), _ptr(keys), _ptr(plist),
            _ptr(ranges))
        assert R == Rmax
        return dict(R=int(R), keys=keys[:R], point_list=plist[:R], ranges=ranges)

    def render_forward(self, pre, bins, W, H, no_color=False):
        T = bins["ranges"].shape[0]
        color = np.zeros((3, H, W), self.dtype)
        final_T = np.zeros((H, W), self.dtype)
        n_contrib = np.zeros((H, W), np.uint32)
        max_contrib = np.zeros(T, np.uint32)
        ev = ctypes.c_int64(0)
        self.lib.orc_render_forward(
            ctypes.c_int(W), ctypes.c_int(H), _ptr(bins["ranges"]), _ptr(bins["point_list"]), _ptr(pre["means2D"]),
            _ptr(pre["conic_opacity"]), _ptr(pre["rgb"]), ctypes.c_int(int(no_color)), _ptr(color), _ptr(final_T),
            _ptr(n_contrib), _ptr(max_contrib), ctypes.byref(ev))
        return dict(color=color, final_T=final_T, n_contrib=n_contrib, max_contrib=max_contrib, evals=ev.value)

    def forward(self, sc, cam, no_color=False):
        """Full reference forward (rasterizer_impl.cu:312-474): returns every stage boundary."""
        W, H = cam["W"], cam["H"]
        pre = self.preprocess(sc, cam, no_color)
        bins = self.binning(pre, W, H)
        img = self.render_forward(pre, bins, W, H, no_color)
        out = dict(pre=pre, bins=bins, **img)
        out["num_rendered"] = bins["R"]
        r = bins["ranges"].astype(np.int64)
        out["num_buckets_ref32"] = 0 if no_color else int(((r[:, 1] - r[:, 0] + 31) // 32).sum())
        return out

    def backward(self, sc, cam, fwd, dL_dpix, lambda_erank=0.0, camera_grads=False):
        """Full reference backward (rasterizer_impl.cu:476-581); returns all ten gradient tensors.  camera_grads=True adds
        dL_dviewmatrix [16], dL_dprojmatrix [16], dL_dcampos [3] (float64; no reference counterpart, see gs_oracle.c)."""
        W, H = cam["W"], cam["H"]
        P = sc["means"].shape[0]
        M = 0 if sc["shs"] is None or sc["shs"].size == 0 else sc["shs"].shape[1]
        pre, bins = fwd["pre"], fwd["bins"]
        z = lambda *s: np.zeros(s, self.dtype)
        g = dict(dL_dmean2D=z(P, 3), dL_dconic=z(P, 4), dL_dopacity=z(P, 1), dL_dcolor=z(P, 3), dL_dmean3D=z(P, 3),
                 dL_dcov3D=z(P, 6), dL_ddc=z(P, 1, 3), dL_dsh=z(P, M, 3), dL_dscale=z(P, 3), dL_drot=z(P, 4))
        dL = self.a(dL_dpix, (3, H, W))
        self.lib.orc_render_backward(
            ctypes.c_int(W), ctypes.c_int(H), ctypes.c_int(P), _ptr(bins["ranges"]), _ptr(bins["point_list"]),
            _ptr(pre["means2D"]), _ptr(pre["conic_opacity"]), _ptr(pre["rgb"]), _ptr(fwd["color"]),
            _ptr(fwd["n_contrib"]), _ptr(dL), _ptr(g["dL_dmean2D"]), _ptr(g["dL_dconic"]), _ptr(g["dL_dopacity"]),
            _ptr(g["dL_dcolor"]))
        r = self.real
        means, scales, rots, dc = self.a(sc["means"]), self.a(sc["scales"]), self.a(sc["rots"]), self.a(sc["dc"])
        shs = self.a(sc["shs"]) if M > 0 else None
        view, proj, campos = self.a(cam["view"]), self.a(cam["proj"]), self.a(cam["campos"])
        args = (ctypes.c_int(P), ctypes.c_int(int(sc["D"])), ctypes.c_int(M), _ptr(means), _ptr(pre["radii"]), _ptr(dc),
                _ptr(shs), _ptr(pre["clamped"]), _ptr(scales), _ptr(rots), r(sc.get("scale_modifier", 1.0)),
                _ptr(pre["cov3D"]), _ptr(view), _ptr(proj), ctypes.c_int(W), ctypes.c_int(H), r(cam["tanfovx"]),
                r(cam["tanfovy"]), r(cam["limx_neg"]), r(cam["limx_pos"]), r(cam["limy_neg"]), r(cam["limy_pos"]),
                _ptr(campos), _ptr(g["dL_dmean2D"]), _ptr(g["dL_dconic"]), _ptr(g["dL_dcolor"]), _ptr(g["dL_dmean3D"]),
                _ptr(g["dL_dcov3D"]), _ptr(g["dL_ddc"]), _ptr(g["dL_dsh"]) if M > 0 else None, _ptr(g["dL_dscale"]),
                _ptr(g["dL_drot"]), r(lambda_erank))
        if camera_grads:
            camg = np.zeros(35, np.float64)
            self.lib.orc_preprocess_backward_cam(*args, _ptr(camg))
            g["dL_dviewmatrix"], g["dL_dprojmatrix"], g["dL_dcampos"] = camg[:16].copy(), camg[16:32].copy(), camg[32:].copy()
        else:
            self.lib.orc_preprocess_backward(*args)
        g["dL_dconic"] = g["dL_dconic"].reshape(P, 2, 2)
        return g

    def adam(self, param, grad, m, v, visible, lr, b1=0.9, b2=0.999, eps=1e-15):
        """In place on param, m, v (numpy arrays of this oracle's dtype, [N,M...])."""
        N = param.shape[0]
        M = param.size // max(N, 1)
        r = self.real
        vis = np.ascontiguousarray(visible.astype(np.uint8))
        grad = self.a(grad)
        self.lib.orc_adam(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), _ptr(vis), r(lr), r(b1), r(b2), r(eps),
                          ctypes.c_uint32(N), ctypes.c_uint32(M))

    def ssim_forward(self, img1, img2, C1=0.01 ** 2, C2=0.03 ** 2, train=True):
        img1, img2 = self.a(img1), self.a(img2)
        B, CH, H, W = img1.shape
        r = self.real
        m = np.zeros_like(img1)
        d = [np.zeros_like(img1) for _ in range(3)] if train else [None] * 3
        self.lib.orc_ssim_forward(ctypes.c_int(B), ctypes.c_int(CH), ctypes.c_int(H), ctypes.c_int(W), r(C1), r(C2),
                                  _ptr(img1), _ptr(img2), _ptr(m), _ptr(d[0]), _ptr(d[1]), _ptr(d[2]))
        return m, d[0], d[1], d[2]

    def ssim_backward(self, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
        img1, img2, dL = self.a(img1), self.a(img2), self.a(dL_dmap)
        B, CH, H, W = img1.shape
        out = np.zeros_like(img1)
        self.lib.orc_ssim_backward(ctypes.c_int(B), ctypes.c_int(CH), ctypes.c_int(H), ctypes.c_int(W), _ptr(img1),
                                   _ptr(img2), _ptr(dL), _ptr(self.a(dm_dmu1)), _ptr(self.a(dm_dsigma1_sq)),
                                   _ptr(self.a(dm_dsigma12)), _ptr(out))
        return out

    def knn(self, pts):
        pts = self.a(pts, (-1, 3))
        out = np.zeros(pts.shape[0], self.dtype)
        self.lib.orc_knn(ctypes.c_int(pts.shape[0]), _ptr(pts), _ptr(out))
        return out

    def extend_select(self, points, depths_rsp, R_cw, t_cw, fx, fy, cx, cy, W, H, final_T):
        pts, d = self.a(points, (-1, 3)), self.a(depths_rsp).reshape(-1)
        n = pts.shape[0]
        keep = np.zeros(n, np.uint8)
        r = self.real
        self.lib.orc_extend_select(ctypes.c_int(n), _ptr(pts), _ptr(d), _ptr(self.a(R_cw).reshape(-1)), _ptr(self.a(t_cw).reshape(-1)),
                                   r(fx), r(fy), r(cx), r(cy), ctypes.c_int(W), ctypes.c_int(H), _ptr(self.a(final_T).reshape(-1)), _ptr(keep))
        return keep.astype(bool)

    def extend_emit(self, keep, points, colors, depths_rsp, scaling_scale, focal, M):
        pts, col, d = self.a(points, (-1, 3)), self.a(colors, (-1, 3)), self.a(depths_rsp).reshape(-1)
        k = int(keep.sum())
        z = lambda *s: np.zeros(s, self.dtype)
        out = dict(xyz=z(k, 3), dc=z(k, 1, 3), rest=z(k, M, 3), opacity=z(k, 1), scaling=z(k, 3), rotation=z(k, 4))
        r = self.real
        kk = self.lib.orc_extend_emit(ctypes.c_int(pts.shape[0]), _ptr(np.ascontiguousarray(keep.astype(np.uint8))), _ptr(pts), _ptr(col),
                                      _ptr(d), r(scaling_scale), r(focal), ctypes.c_int(M), _ptr(out["xyz"]), _ptr(out["dc"]),
                                      _ptr(out["rest"]) if M > 0 else None, _ptr(out["opacity"]), _ptr(out["scaling"]), _ptr(out["rotation"]))
        assert kk == k
        return out
