#!/usr/bin/env python
"""TIMING PROBE (results of the overlapped mode are numerically WRONG on purpose): how much of the optimiser's SH tail can hide behind the next
step's geometry / sort / binning stages?  Three schedules of the 2M / 1080p training step in one process:
  fused      the default step (Adam inside the per-Gaussian backward), one stream
  split      per-Gaussian backward writes the small gradients + dRGB, then ONE launch rebuilds the SH rows and runs Adam of all groups — same stream
  overlap    the same two kernels, the second one on a SIDE stream; the next forward only waits for it right before the blend forward
             (gslic_raster_params.sh_ready_event).  The next step's preprocess reads parameters that are being updated: timing only.
    python tools/experiments/overlap_probe.py [--steps 200] [--map-order morton]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--map-order", default="morton")
    args = ap.parse_args()
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import loss as loss_utils
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd import trainer
    from gaussian_lic_amd.camera import synthetic_camera
    from gaussian_lic_amd.synthetic import gt_image, random_scene
    from gaussian_lic_amd.trainer import DEFAULT_LRS, GradSlab
    dev = torch.device("cuda:0")
    W, H, P = 1920, 1080, 2_000_000
    model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev, order=args.map_order)
    model.training_setup({k: v * 0.01 for k, v in DEFAULT_LRS.items()})
    cam = synthetic_camera(W, H).to_device(dev)
    gt, bg = gt_image(H, W, seed=2).to(dev), torch.zeros(3, device=dev)
    fl = loss_utils.FusedLoss(trainer.LAMBDA_DSSIM)
    e = torch.empty(0, device=dev)
    slab = model._grad_slab = GradSlab(model)
    slab.vis_or = torch.zeros(P, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    main_s = torch.cuda.current_stream(dev)
    state = dict(ev=None)

    def split_step(overlap):
        xyz, dc, rest = model.xyz.detach(), model.features_dc.detach(), model.features_rest.detach()
        op, sc, rot = model.opacity.detach(), model.scaling.detach(), model.rotation.detach()
        scal = (float(cam.tanfovx), float(cam.tanfovy))
        lims = (float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos))
        (R, B, image, _T, radii, geom, binning, img, sample) = rz.rasterize_gaussians(
            bg, xyz, e, op, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, *scal, H, W, *lims, dc, rest, 3, cam.d_camera_center,
            False, False, False, raw_params=True, tie_rank=model.tie_rank, sh_ready_event=state["ev"] if overlap else None)
        dL, _terms = fl.forward_backward(image, gt)
        rz.rasterize_gaussians_backward(bg, xyz, radii, e, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, *scal, *lims, dL, dc, rest, 3,
                                        cam.d_camera_center, geom, R, binning, img, B, sample, 0.0, False, raw_params=True, out=slab.views, rgb_out=slab.rgb,
                                        payload=(slab.pay_vis, slab.pay_campos))
        v = slab.views

        def tail():
            model.optimizer.step_all_from_exchange(xyz, slab.pay_campos, slab.rgb.view(-1), 3, 1, 0, slab.pay_vis, slab.pay_bytes, slab.vis_or,
                                                   (v["xyz"], v["opacity"], v["scaling"], v["rotation"]))
        if overlap:
            done = torch.cuda.Event()
            done.record(main_s)
            with torch.cuda.stream(side):
                side.wait_event(done)
                tail()
                ev = torch.cuda.Event()
                ev.record(side)
            state["ev"] = ev
        else:
            tail()

    def clock(fn, n):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / n

    for _ in range(25):
        trainer.training_step_fused(model, cam, gt, bg)
    res = {}
    for rep in range(2):
        res.setdefault("fused", []).append(round(clock(lambda: trainer.training_step_fused(model, cam, gt, bg), args.steps), 4))
        res.setdefault("split", []).append(round(clock(lambda: split_step(False), args.steps), 4))
        res.setdefault("overlap", []).append(round(clock(lambda: split_step(True), args.steps), 4))
    print("ms per step:", res, flush=True)


if __name__ == "__main__":
    main()
