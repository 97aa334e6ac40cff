"""Per-step wall time of the first steps of the N > 1 path in a ONE-rank RCCL group (collectives are copies): where the warm-up of that path goes.
    GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 python tools/diag_dist_warmup.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
if os.environ.get("GSLIC_FORCE_DIST") == "1":
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import gaussian_lic_amd
from gaussian_lic_amd import trainer
from gaussian_lic_amd.camera import synthetic_camera
from gaussian_lic_amd.synthetic import random_scene, gt_image
W, H, P = 1920, 1080, 2_000_000
model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev); model.training_setup({k: v * 0.01 for k, v in trainer.DEFAULT_LRS.items()})
cam = synthetic_camera(W, H).to_device(dev); gt = gt_image(H, W).to(dev); bg = torch.zeros(3, device=dev)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 160):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.training_step_fused(model, cam, gt, bg)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("ms per step (synchronised), steps 0..:", " ".join(f"{t:.2f}" for t in ts[:24]))
for a in range(24, len(ts), 16):
    print(f"steps {a}..{a + 15}: mean {sum(ts[a:a + 16]) / len(ts[a:a + 16]):.3f}")
print("steps slower than 1.5x the median:", [(i, round(t, 2)) for i, t in enumerate(ts) if t > 1.5 * sorted(ts)[len(ts) // 2]])
print("memory allocated / reserved MB:", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
