"""-m gpu: saveMap from the DEVICE side (src/gaussian.cpp:306-397).  A map that has been trained on the HIP kernels and grown by
an extend() append that had to double its capacity is exported with io_ply.save_map (skybox rows dropped, :310-316), compared BYTE FOR BYTE
with the file the reference's own vendored tinyply writes for the same tensors (oracle/_ref/ply_writer: src/tinyply.cpp compiled in place +
ply_driver.cpp replaying saveMap's call sequence), and loaded back into the benchmark through `bench.py --ply`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import make_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLY_WRITER = os.path.join(ROOT, "oracle", "_ref", "ply_writer")


def test_trained_and_grown_device_map_exports_like_the_reference(tmp_path):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import io_ply, trainer
    from gaussian_lic_amd.synthetic import gt_image, lidar_scene
    P, W, H, SKY = 20000, 320, 192, 500
    raw, sc, camd, cam = make_scene("random", P, W, H, 3, 71)
    dev = torch.device("cuda:0")
    cam.to_device(dev)
    model = trainer.GaussianModel(raw, dev, capacity=P)          # no headroom: the append below has to reallocate
    model.training_setup()
    gt, bg = gt_image(H, W).to(dev), torch.zeros(3, device=dev)
    for _ in range(3):
        trainer.training_step_fused(model, cam, gt, bg)
    cap0 = model.capacity
    frame = lidar_scene(4000, W, H, sh_degree=3, seed=72)
    pts = frame["xyz"].to(dev)
    col = (frame["features_dc"].reshape(-1, 3) * 0.28209479177387814 + 0.5).to(dev)
    Rcw = torch.from_numpy(cam.world_view_transform[:3, :3].T.copy())
    tcw = torch.from_numpy(cam.world_view_transform[3, :3].copy())
    k = model.extend(cam, pts, col, frame["xyz"][:, 2].contiguous().to(dev), Rcw, tcw, (float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy)))
    assert k > 0 and model.P == P + k and model.capacity >= 2 * cap0, (k, model.P, cap0, model.capacity)   # capacity doubling (trainer._reserve)
    for _ in range(2):
        trainer.training_step_fused(model, cam, gt, bg)           # the grown map trains on: new rows have moments and move
    torch.cuda.synchronize()

    # ---- export from the device tensors
    path = str(tmp_path / "point_cloud.ply")
    n = io_ply.save_map(model, path, skybox_points_num=SKY)
    assert n == model.P - SKY
    # ---- the reference's writer on the same tensors, prepared as saveMap prepares them (gaussian.cpp:309-316)
    if not os.path.exists(PLY_WRITER):
        pytest.skip("oracle/_ref/ply_writer not built (needs /root/reference at build time)")
    s = slice(SKY, None)
    g = lambda t: t.detach()[s].float().cpu()
    arrs = dict(xyz=g(model.xyz), f_dc=g(model.features_dc).transpose(1, 2).flatten(1), f_rest=g(model.features_rest).transpose(1, 2).flatten(1),
                opacity=g(model.opacity), scale=g(model.scaling), rotation=g(model.rotation))
    files = []
    for key in ("xyz", "f_dc", "f_rest", "opacity", "scale", "rotation"):
        f = str(tmp_path / (key + ".f32"))
        np.ascontiguousarray(arrs[key].contiguous().numpy(), "<f4").tofile(f)
        files.append(f)
    want_path = str(tmp_path / "reference.ply")
    subprocess.run([PLY_WRITER, want_path, str(n), str(model.features_rest.shape[1])] + files, check=True)
    got, want = open(path, "rb").read(), open(want_path, "rb").read()
    assert got == want, f"{len(got)} vs {len(want)} bytes, first difference at {next((i for i, (x, y) in enumerate(zip(got, want)) if x != y), None)}"

    # ---- and back in: the loader returns the trained rows bit for bit, and the benchmark trains on the file
    back = io_ply.load_map(path)
    for name in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        assert torch.equal(back[name], getattr(model, name).detach()[s].cpu()), name
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ply", path, "--steps", "3", "--warmup", "2", "--width", str(W), "--height", str(H),
                        "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["value"] > 0 and str(n) in line["config"]["workload"] and "saveMap PLY" in line["config"]["workload"]
