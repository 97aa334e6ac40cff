"""Workload for the PMC passes: a calibration copy of known size (1 GiB read + 1 GiB write through torch's vectorised
copy kernel), then a few default-config bench steps.  Run under `rocprofv3 --pmc FETCH_SIZE ...` / `--pmc WRITE_SIZE ...`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gaussian_lic_amd
from gaussian_lic_amd import trainer
from gaussian_lic_amd.camera import synthetic_camera
from gaussian_lic_amd.synthetic import random_scene, gt_image

dev = torch.device("cuda:0")
a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)          # calibration: 2^30 bytes read, 2^30 bytes written per launch
torch.cuda.synchronize()
del a, b
W, H, P = 1920, 1080, 2_000_000
model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev); model.training_setup({k: v * 0.01 for k, v in trainer.DEFAULT_LRS.items()})  # stationary scene, as bench.py
cam = synthetic_camera(W, H).to_device(dev)
gt = gt_image(H, W).to(dev); bg = torch.zeros(3, device=dev)
fused = os.environ.get("PMC_HOST", "fused") == "fused"   # the default bench path; PMC_HOST=dropin for the per-op path
for _ in range(4):
    (trainer.training_step_fused if fused else trainer.training_step)(model, cam, gt, bg)
torch.cuda.synchronize()
print("pmc workload done")
