import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from conftest import make_scene, rel_err
from gpu_helpers import hip_backward, hip_forward, npy
from gaussian_lic_amd.synthetic import pixel_grad
from oracle.oracle import Oracle
for (kind, P, W, H, deg, seed) in [("random", 3000, 70, 50, 2, 5), ("random", 30000, 320, 192, 3, 1)]:
    raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed)
    got = hip_forward(raw, cam)
    dL = pixel_grad(H, W)
    g = hip_backward(got, dL)
    orc = Oracle(np.float32)
    ref = orc.forward(sc, camd)
    gref = orc.backward(sc, camd, ref, dL.numpy())
    print(kind, P, W, H, {k: float("%.2e" % rel_err(g[k].reshape(-1), gref[k].reshape(-1))) for k in g})
