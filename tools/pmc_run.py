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
MAP_ORDER = os.environ.get("PMC_MAP_ORDER", "morton")   # bench.py's default row order
model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev, order=MAP_ORDER); model.training_setup({k: v * 0.01 for k, v in trainer.DEFAULT_LRS.items()})  # stationary scene, as bench.py
cam = synthetic_camera(W, H).to_device(dev)
gt = gt_image(H, W).to(dev); bg = torch.zeros(3, device=dev)
fused = os.environ.get("PMC_HOST", "fused") == "fused"   # the default bench path; PMC_HOST=dropin for the per-op path
# the driver's command is `bench.py --steps 20 --warmup 5`: 25 steps from the seeded start, so that the counters describe the workload of the
# line they are replayed into (bench.py replays them only when its own unit counts are within 2 % of the ones stored here)
for _ in range(int(os.environ.get("PMC_STEPS", "25"))):
    (trainer.training_step_fused if fused else trainer.training_step)(model, cam, gt, bg)
torch.cuda.synchronize()
if os.environ.get("PMC_UNITS"):
    import json
    from gaussian_lic_amd import _lib, rasterizer as rz
    from gaussian_lic_amd.rasterizer import render
    with torch.no_grad():
        vis = render(cam, model, bg)[3]
        e = torch.empty(0, device=dev)
        rs = rz.GaussianRasterizationSettings(H, W, float(cam.tanfovx), float(cam.tanfovy), float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg),
                                              float(cam.limy_pos), bg, 1.0, cam.d_world_view_transform, cam.d_full_proj_transform, 3, cam.d_camera_center)
        fwd = rz.rasterize_gaussians(bg, model.get_xyz(), e, model.get_opacity(), model.get_scaling(), model.get_rotation(), 1.0, e, rs.viewmatrix, rs.projmatrix,
                                     rs.tanfovx, rs.tanfovy, H, W, rs.limx_neg, rs.limx_pos, rs.limy_neg, rs.limy_pos, model.get_features_dc(),
                                     model.get_features_rest(), 3, rs.campos, False, False, False)
        dbg = rz.debug_export(rs, P, 15, fwd[0], fwd[1], fwd[5], fwd[6], fwd[7], fwd[8], what=("ranges", "max_contrib"))
    n_t = (dbg["ranges"][:, 1] - dbg["ranges"][:, 0]).long()
    live_b = (dbg["max_contrib"].long() + 63) // 64
    strict = bool(_lib.set_math_mode(True)); _lib.set_math_mode(strict)
    json.dump({"P": P, "V": int(vis.sum().item()), "R": int(fwd[0]), "B": int(fwd[1]), "B_live": int(live_b.sum().item()),
               "R_live": int(torch.minimum(n_t, 64 * live_b).sum().item()), "steps_before": int(os.environ.get("PMC_STEPS", "25")), "strict": strict, "map_order": MAP_ORDER},
              open(os.environ["PMC_UNITS"], "w"))
print("pmc workload done")
