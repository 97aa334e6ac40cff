#!/usr/bin/env python
"""Timing probe: the fused step through (a) the allocator-callback forward (two host round trips per forward), (b) the capacity-mode forward as
eager launches (GraphedStep(use_graph=False)), (c) the capacity-mode step as one hipGraph replay.  python tools/experiments/capacity_eager_probe.py [P W H n]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gaussian_lic_amd
from gaussian_lic_amd import trainer
from gaussian_lic_amd.camera import synthetic_camera
from gaussian_lic_amd.synthetic import random_scene, gt_image
from gaussian_lic_amd.trainer import DEFAULT_LRS
P, W, H, N = (int(v) for v in (sys.argv[1:5] + ["2000000", "1920", "1080", "300"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev, order=os.environ.get("PROBE_ORDER", "morton"))
model.training_setup({k: v * 0.01 for k, v in DEFAULT_LRS.items()})
cam = synthetic_camera(W, H).to_device(dev); gt = gt_image(H, W, seed=2).to(dev); bg = torch.zeros(3, device=dev)
dbg = os.environ.get("PROBE_DEBUG") == "1"
def mark(s):
    if dbg:
        torch.cuda.synchronize(); print("ok:", s, flush=True)
def clock(fn, n=N):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / n
for _ in range(int(os.environ.get("PROBE_WARM", "25"))): trainer.training_step_fused(model, cam, gt, bg)
mark("warm")
which = os.environ.get("PROBE_WHICH", "ceg")
res = {}
if "e" in which:
    ge = trainer.GraphedStep(model, cam, gt, bg, check_every=int(os.environ.get("PROBE_CHECK", "16")), use_graph=False, headroom=float(os.environ.get("PROBE_HEADROOM", "1.25"))); mark("ge built")
if "g" in which:
    gg = trainer.GraphedStep(model, cam, gt, bg, check_every=int(os.environ.get("PROBE_CHECK", "16")), use_graph=True); mark("gg built")
for r in range(2):
    if "c" in which: res.setdefault("callbacks", []).append(round(clock(lambda: trainer.training_step_fused(model, cam, gt, bg)), 4)); mark("callbacks")
    if "e" in which: res.setdefault("capacity_eager", []).append(round(clock(ge.step), 4)); mark("eager")
    if "g" in which: res.setdefault("graph", []).append(round(clock(gg.step), 4)); mark("graph")
print(res, flush=True)
