#!/usr/bin/env python
"""Digests of the HIP path's nine gradients on posed fuzz case 20 (tests/test_pose_reference_gpu.py), strict arithmetic, atomic binning, Morton rows:
which library build changes which bits.  GSLIC_HIP_LIB=<lib> python tools/experiments/case20_digest.py"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_scene
from gpu_helpers import hip_backward, hip_forward
from gaussian_lic_amd import _lib, trainer
from gaussian_lic_amd.synthetic import pixel_grad, activate
from test_pose_reference_gpu import FUZZ
kind, P, W, H, deg, seed, view, sigma_scale, scale_modifier = FUZZ[20]
raw, sc, camd, cam = make_scene(kind, P, W, H, deg, seed, view=view, sigma_scale=sigma_scale)
dL = pixel_grad(H, W, seed=1)
perm = trainer.morton_order(raw["xyz"]); tie = perm.to(torch.int32).to("cuda:0")
act = {k: (v[perm].contiguous() if torch.is_tensor(v) else v) for k, v in activate(raw).items()}
_lib.set_binning_mode("atomic"); _lib.set_math_mode(True)
got = hip_forward(raw, cam, export=("n_contrib",), scale_modifier=scale_modifier, tie_rank=tie, act=act)
g = hip_backward(got, dL)
for k in sorted(g):
    print(k, hashlib.sha256(np.ascontiguousarray(g[k]).tobytes()).hexdigest()[:16])
