#!/usr/bin/env python
"""Timing probe: the joint map + camera-pose step (gslic_rasterize_backward_camera: preprocess_bwd_kernel<.., CAM = true>) and the N > 1 style
un-fused backward (no Adam inside: gradients written), per step and per kernel.  python tools/pose_step_probe.py [P W H n]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_lic_amd
from gaussian_lic_amd import trainer, _lib
from gaussian_lic_amd.camera import synthetic_camera
from gaussian_lic_amd.synthetic import random_scene, gt_image
from gaussian_lic_amd.trainer import DEFAULT_LRS
P, W, H, N = (int(v) for v in (sys.argv[1:5] + ["2000000", "1920", "1080", "100"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev, order="morton")
model.training_setup({k: v * 0.01 for k, v in DEFAULT_LRS.items()})
cam = synthetic_camera(W, H, 3).to_device(dev); gt = gt_image(H, W, seed=2).to(dev); bg = torch.zeros(3, device=dev)
def clock(fn, n=N):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    _lib.profile_reset(); _lib.profile_enable(True)
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t) / n
    k = _lib.profile_collect(); _lib.profile_enable(False)
    return round(dt, 4), {n_: round(v[0] / max(v[1], 1), 4) for n_, v in k.items() if n_ in ("preprocess_bwd", "adam", "render_bwd")}
for _ in range(25): trainer.training_step_fused(model, cam, gt, bg)
print("pose_step", clock(lambda: trainer.training_step_with_pose(model, cam, gt, bg, pose_lr=1e-6)), flush=True)
print("fused_step", clock(lambda: trainer.training_step_fused(model, cam, gt, bg)), flush=True)
