"""CPU: PLY wire format of saveMap and the evaluation metrics of loss_utils.h (row 4)."""
import os

import pytest
import numpy as np
import torch

from conftest import rel_err


class _M:
    pass


def test_ply_roundtrip_and_layout(tmp_path):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import io_ply
    from gaussian_lic_amd.synthetic import random_scene
    raw = random_scene(257, 64, 48, 3, 7)
    m = _M()
    for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        setattr(m, k, raw[k])
    path = str(tmp_path / "point_cloud.ply")
    n = io_ply.save_map(m, path, skybox_points_num=7)
    assert n == 250
    blob = open(path, "rb").read()
    head, body = blob.split(b"end_header\n", 1)
    lines = head.decode().strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 250"]
    props = [l.split()[2] for l in lines[3:]]
    assert props == ["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + ["opacity", "scale_0", "scale_1",
                                                                                                          "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(body) == 250 * 59 * 4
    row0 = np.frombuffer(body[:59 * 4], "<f4")
    np.testing.assert_array_equal(row0[:3], raw["xyz"][7].numpy())
    # f_rest is channel-major: f_rest_0..14 = all 15 coefficients of channel 0 (gaussian.cpp:312-313)
    np.testing.assert_array_equal(row0[6:6 + 15], raw["features_rest"][7, :, 0].numpy())
    np.testing.assert_array_equal(row0[6 + 15:6 + 30], raw["features_rest"][7, :, 1].numpy())
    back = io_ply.load_map(path)
    for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        assert torch.equal(back[k], raw[k][7:]), k
    assert back["sh_degree"] == 3


def test_eval_ssim_and_psnr_match_oracle(oracle32):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import loss
    rng = np.random.default_rng(0)
    a, b = rng.random((1, 3, 40, 56)).astype(np.float32), rng.random((1, 3, 40, 56)).astype(np.float32)
    s = loss.ssim(torch.from_numpy(a), torch.from_numpy(b))
    m = oracle32.ssim_forward(a, b, train=False)[0]
    assert abs(float(s) - float(m.mean())) < 2e-5
    p = loss.psnr(torch.from_numpy(a), torch.from_numpy(b))
    assert abs(float(p) - 10 * np.log10(1.0 / ((a - b) ** 2).mean())) < 1e-4


def test_ply_equals_reference_tinyply_golden(tmp_path):
    """save_map byte-for-byte against files written by the REFERENCE's own tinyply with saveMap's call sequence
    (tests/golden/savemap_*.ply from oracle/ref_build/make_ply_golden.py: src/tinyply.cpp compiled in place + ply_driver.cpp)."""
    import os
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import io_ply
    from oracle.ref_build.make_ply_golden import CASES, model_for
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, P, deg, sky, seed in CASES:
        raw = model_for(P, deg, seed)
        m = _M()
        for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
            setattr(m, k, raw[k])
        path = str(tmp_path / (name + ".ply"))
        io_ply.save_map(m, path, skybox_points_num=sky)
        want = open(os.path.join(gdir, name + ".ply"), "rb").read()
        got = open(path, "rb").read()
        assert got == want, f"{name}: {len(got)} vs {len(want)} bytes, first difference at {next((i for i, (x, y) in enumerate(zip(got, want)) if x != y), None)}"
        back = io_ply.load_map(os.path.join(gdir, name + ".ply"))   # and the loader reads the reference-written file
        for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
            assert torch.equal(back[k], raw[k][sky:]), (name, k)


def test_save_map_of_a_permuted_model_writes_the_original_order(tmp_path):
    """A model that stores its rows permuted (trainer.GaussianModel(order="morton")) exposes original_order(); saveMap then writes the rows in the
    map's ORIGINAL order, byte for byte the file of the unpermuted model (CPU stand-in models; the device model: tests/test_morton_order_gpu.py)."""
    import types
    import torch
    from gaussian_lic_amd import io_ply, trainer
    from gaussian_lic_amd.synthetic import random_scene
    raw = random_scene(500, 64, 48, sh_degree=3, seed=3)
    perm = trainer.morton_order(raw["xyz"])
    assert sorted(perm.tolist()) == list(range(500))
    a = types.SimpleNamespace(**{k: v for k, v in raw.items() if torch.is_tensor(v)})
    b = types.SimpleNamespace(**{k: v[perm] for k, v in raw.items() if torch.is_tensor(v)})
    b.original_order = lambda: torch.argsort(perm)
    pa, pb = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    io_ply.save_map(a, pa, skybox_points_num=7); io_ply.save_map(b, pb, skybox_points_num=7)
    assert open(pa, "rb").read() == open(pb, "rb").read()


@pytest.mark.parametrize("name", ["eval_3x70x50", "eval_3x96x64", "eval_1x33x17"])
def test_eval_metrics_match_the_reference_header_compiled_on_cpu(oracle32, name):
    """Evaluation PSNR / conv-SSIM pinned to REFERENCE CODE (round-4 review, missing 6): tests/golden/eval_*.npz are the numbers of the reference's
    own loss_utils.h:30-128 (l1_loss, psnr, ssim through gaussian() / create_window() / _ssim()), compiled unmodified against CPU LibTorch by
    oracle/ref_build/make_eval_golden.py.  Held to them here: the host-side mirror gaussian_lic_amd.loss (same LibTorch ops) and the C oracle's
    SSIM map, which tests/test_eval_gpu.py holds the device path (fused-SSIM kernel in inference mode) to."""
    import torch
    from gaussian_lic_amd import loss
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    a, b = torch.from_numpy(z["img1"]), torch.from_numpy(z["img2"])
    np.testing.assert_array_equal(loss._gaussian_window(11, 1.5, a.size(0), a).numpy(), z["window"])      # gaussian() / create_window(), bit for bit
    assert abs(float(loss.l1_loss(a, b)) - float(z["l1"])) <= 1e-7
    assert abs(float(loss.psnr(a, b)) - float(z["psnr"])) <= 2e-6 * float(z["psnr"])
    assert abs(float(loss.ssim(a, b)) - float(z["ssim"])) <= 2e-7
    assert abs(float(loss.ssim(a[None], b[None], size_average=False)) - float(z["ssim_per_image"])) <= 2e-7
    # the C oracle's SSIM map (separable 11-tap window, zero padding: ssim.cu:8-18,261-282) averages to the reference's conv2d formulation
    m = oracle32.ssim_forward(z["img1"][None], z["img2"][None], train=False)[0]
    assert abs(float(m.astype(np.float64).mean()) - float(z["ssim"])) <= 2e-6
