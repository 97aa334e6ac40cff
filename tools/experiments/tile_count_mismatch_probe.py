#!/usr/bin/env python
"""One Gaussian of fuzz scene (random, P = 25600, 320x180, degree 3, seed 845806: case 385 of `tests/fuzz_vs_reference.py 600 8001`) gets a different
tiles_touched from the HIP path than from the reference's own kernels.  Who is right?  Prints the Gaussian's record from the three sides (HIP, the
reference's kernels on this GPU, the C oracle in fp32 and fp64) and the per-tile test values around the tile that differs."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_scene
from gpu_helpers import hip_forward, npy
from gaussian_lic_amd import _lib
from oracle.ref_build import refkernels
from oracle.oracle import Oracle, build
kind, P, W, H, deg, seed = "random", 25600, 320, 180, 3, 845806
raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed)
rk = refkernels.RefKernels()
ref = rk.run(sc, camd, None)
_lib.set_binning_mode("radix"); _lib.set_math_mode(True)
got = hip_forward(raw, cam, export=("tiles_touched", "means2D", "depths", "conic_opacity"))
tt = npy(got["dbg"]["tiles_touched"]).astype(np.uint32)
bad = np.nonzero(tt != ref["tiles_touched"])[0]
print("mismatching Gaussians:", bad, "HIP", tt[bad], "reference", ref["tiles_touched"][bad], "radii HIP", npy(got["radii"])[bad], "ref", ref["radii"][bad])
build()
for dt in (np.float32, np.float64):
    o = Oracle(dt)
    f = o.preprocess(sc, camd)
    print("oracle", dt.__name__, "tiles_touched", np.asarray(f["tiles_touched"])[bad], "radii", np.asarray(f["radii"])[bad], "means2D", f["means2D"][bad], "conic_opacity", f["conic_opacity"][bad])
g = int(bad[0])
print("means2D HIP", npy(got["dbg"]["means2D"])[g], "ref", ref["means2D"][g])
print("conic_opacity HIP", npy(got["dbg"]["conic_opacity"])[g], "ref", ref["conic_opacity"][g])
print("bits equal:", np.array_equal(npy(got["dbg"]["means2D"])[g], ref["means2D"][g]), np.array_equal(npy(got["dbg"]["conic_opacity"])[g], ref["conic_opacity"][g]))

# the exact tile test of forward.cu:151-230 evaluated in fp64 for this Gaussian over its rectangle (which tile is the marginal one)
mx, my = (float(v) for v in ref["means2D"][g]); cA, cB, cC, op = (float(v) for v in ref["conic_opacity"][g]); r = int(ref["radii"][g])
import math
gx, gy = (W + 15) // 16, (H + 15) // 16
x0, x1 = min(gx, max(0, int((mx - r) / 16))), min(gx, max(0, int((mx + r + 15) / 16)))
y0, y1 = min(gy, max(0, int((my - r) / 16))), min(gy, max(0, int((my + r + 15) / 16)))
print("rect", x0, x1, y0, y1, "opacity", op, "threshold power ln(1/(255 op)) =", math.log(1.0 / (255.0 * op)))

# the sixteen tiles of the rectangle: the reference's formula (forward.h:39-78) in numpy fp32 (every operation rounded on its own) and in fp64
def power(dt, tx, ty):
    f = dt
    A, B, C = f(ref["conic_opacity"][g][0]), f(ref["conic_opacity"][g][1]), f(ref["conic_opacity"][g][2])
    mxx, myy = f(ref["means2D"][g][0]), f(ref["means2D"][g][1])
    lo_x, lo_y, hi_x, hi_y = f(tx * 16), f(ty * 16), f(tx * 16 + 15), f(ty * 16 + 15)
    gap_x = lo_x - mxx; left = f(1.0) if gap_x > 0 else f(0.0); out_x = left + (f(1.0) if mxx > hi_x else f(0.0))
    gap_y = lo_y - myy; above = f(1.0) if gap_y > 0 else f(0.0); out_y = above + (f(1.0) if myy > hi_y else f(0.0))
    if not (out_y + out_x > 0):
        return f(0.0)
    sx, sy = hi_x - lo_x, hi_y - lo_y
    px = left * lo_x + (f(1.0) - left) * hi_x; py = above * lo_y + (f(1.0) - above) * hi_y
    dx = f(np.copysign(sx, gap_x)); dy = f(np.copysign(sy, gap_y))
    ox, oy = mxx - px, myy - py
    rx, ry = f(1.0) / (sx * sx * A), f(1.0) / (sy * sy * C)
    sat = lambda v: v if 0 < v < 1 else (f(1.0) if v >= 1 else f(0.0))
    u = out_y * sat((dx * A * ox + dx * B * oy) * rx); v = out_x * sat((dy * B * ox + dy * C * oy) * ry)
    qx, qy = px + u * dx, py + v * dy
    ex, ey = mxx - qx, myy - qy
    return f(0.5) * (A * ex * ex + C * ey * ey) + B * ex * ey
thr32 = np.float32(math.log(float(np.float32(op) / (np.float32(1.0) / np.float32(255.0)))))
print("threshold (glibc log of the fp32 quotient, rounded to fp32):", repr(thr32))
for ty in range(y0, y1):
    print("  row", ty, " ".join(f"{float(power(np.float32, tx, ty)):.7f}{'*' if power(np.float32, tx, ty) <= thr32 else ' '}/{float(power(np.float64, tx, ty)):.7f}" for tx in range(x0, x1)))

# in which tiles' lists does the Gaussian appear on either side?
def tiles_of(point_list, ranges, gid):
    pos = np.nonzero(np.asarray(point_list) == gid)[0]
    r = np.asarray(ranges).reshape(-1, 2)
    return sorted(int(np.nonzero((r[:, 0] <= p) & (p < r[:, 1]))[0][0]) for p in pos)
got2 = hip_forward(raw, cam, export=("point_list", "ranges"))
t_ref = tiles_of(ref["point_list"], ref["ranges"], g); t_hip = tiles_of(npy(got2["dbg"]["point_list"]), npy(got2["dbg"]["ranges"]), g)
fmt = lambda ts: [(t % gx, t // gx) for t in ts]
print("tiles (x, y) listing the Gaussian — reference:", fmt(t_ref)); print("                                   HIP:      ", fmt(t_hip))
print("only in the reference's lists:", fmt(sorted(set(t_ref) - set(t_hip))), " only in HIP's:", fmt(sorted(set(t_hip) - set(t_ref))))
ref2 = rk.run(sc, camd, None)
print("a second run of the reference's kernels: tiles_touched", ref2["tiles_touched"][g])
