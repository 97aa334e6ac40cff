#!/usr/bin/env python
"""Diagnostic: where a live wave of render_bwd_scan_kernel spends its lifetime.  Needs a library built with -DGS_SCAN_TIMELINE=1
(tools/ab/build_one.sh render_bwd_scan.hip "-DGS_SCAN_TIMELINE=1" tools/ab/libgslic_hip_tl.so) passed as GSLIC_HIP_LIB: every live wave adds the
shader-clock cycles between its phase stamps (each stamp waits for all of the wave's outstanding memory operations) to six device counters.
python tools/bwd_timeline_probe.py [P W H n]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_lic_amd
from gaussian_lic_amd import trainer, _lib
from gaussian_lic_amd.camera import synthetic_camera
from gaussian_lic_amd.synthetic import random_scene, gt_image
from gaussian_lic_amd.trainer import DEFAULT_LRS
P, W, H, N = (int(v) for v in (sys.argv[1:5] + ["2000000", "1920", "1080", "20"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev, order="morton")
model.training_setup({k: v * 0.01 for k, v in DEFAULT_LRS.items()})
cam = synthetic_camera(W, H).to_device(dev); gt = gt_image(H, W, seed=2).to(dev); bg = torch.zeros(3, device=dev)
for _ in range(25): trainer.training_step_fused(model, cam, gt, bg)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()
assert L.gslic_debug_scan_timeline(buf, 1) == 0
for _ in range(N): trainer.training_step_fused(model, cam, gt, bg)
torch.cuda.synchronize()
assert L.gslic_debug_scan_timeline(buf, 0) == 0
n = buf[8]
names = ["header chain (bucket -> tile -> range -> slot)", "record gather + entry staging", "per-pixel loads of the quadrants", "quadrant set-up (masks, compaction, LDS records)",
         "block passes", "rows + flags stored and acknowledged"]
tot = buf[9]
print(f"{n} live waves over {N} launches = {n / N:.0f} per launch; mean lifetime {tot / n:.0f} cycles of the 100 MHz-class shader counter")
for k, nm in enumerate(names):
    print(f"  {nm:55s} {buf[k] / n:10.1f} cycles  {100.0 * buf[k] / tot:5.1f} %")
print(f"  chunk x group iterations per live wave {buf[6] / n:.1f} (16 entries x 4 pixels each), block passes per live wave {buf[7] / n:.2f}")
