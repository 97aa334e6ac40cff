"""Ad-hoc: split a training step into phases (wall + device time) and watch the caching allocator."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gaussian_lic_amd
from gaussian_lic_amd import trainer, loss as loss_utils
from gaussian_lic_amd.rasterizer import render
from gaussian_lic_amd.camera import synthetic_camera
from gaussian_lic_amd.synthetic import random_scene, gt_image

dev = torch.device("cuda:0")
W, H, P = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
raw = random_scene(P, W, H, 3, 0)
model = trainer.GaussianModel(raw, dev); model.training_setup()
cam = synthetic_camera(W, H).to_device(dev)
gt = gt_image(H, W).to(dev); bg = torch.zeros(3, device=dev)

def phase_step():
    t = [time.perf_counter()]
    def mark():
        torch.cuda.synchronize(); t.append(time.perf_counter())
    image, _, _, visible, _ = render(cam, model, bg); mark()
    Ll1 = loss_utils.l1_loss(image, gt)
    ssim = loss_utils.fused_ssim(image.unsqueeze(0), gt.unsqueeze(0))
    loss = 0.8 * Ll1 + 0.2 * (1.0 - ssim); mark()
    loss.backward(); mark()
    grads = [p.grad for p in model.parameters()]
    model.optimizer.set_visibility_and_N(visible, P)
    model.optimizer.step(grads); model.optimizer.zero_grad(True); mark()
    return [1e3 * (b - a) for a, b in zip(t[:-1], t[1:])]

for i in range(8):
    s0 = torch.cuda.memory_stats()
    ph = phase_step()
    s1 = torch.cuda.memory_stats()
    print(f"step {i}: fwd {ph[0]:.2f} loss {ph[1]:.2f} bwd {ph[2]:.2f} adam {ph[3]:.2f} ms | device allocs +{s1['num_device_alloc']-s0['num_device_alloc']}"
          f" frees +{s1['num_device_free']-s0['num_device_free']} reserved {s1['reserved_bytes.all.current']/2**30:.2f} GiB", flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    trainer.training_step(model, cam, gt, bg)
torch.cuda.synchronize()
print("unsplit ms/step", 1e2 * (time.perf_counter() - t0))
