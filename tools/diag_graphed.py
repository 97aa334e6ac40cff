"""GraphedStep on a small scene with a watchdog that prints the Python stack of a stall (development aid)."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(45, repeat=False, file=sys.stderr)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gaussian_lic_amd
from gaussian_lic_amd import trainer
from gaussian_lic_amd.camera import synthetic_camera
from gaussian_lic_amd.synthetic import random_scene, gt_image
dev = torch.device("cuda:0")
W, H, P = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (640, 360, 200000)))
model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev); model.training_setup({k: v * 0.01 for k, v in trainer.DEFAULT_LRS.items()})
cam = synthetic_camera(W, H).to_device(dev); gt = gt_image(H, W).to(dev); bg = torch.zeros(3, device=dev)
for _ in range(5):
    trainer.training_step_fused(model, cam, gt, bg)
torch.cuda.synchronize(); print("eager ok", flush=True)
gs = trainer.GraphedStep(model, cam, gt, bg, check_every=0)
print("captured", flush=True)
for i in range(23):
    gs.step()
torch.cuda.synchronize(); print("replayed", flush=True)
print("check ->", gs.check(), flush=True)
