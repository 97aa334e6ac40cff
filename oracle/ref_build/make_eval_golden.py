"""Generates tests/golden/eval_*.npz with the REFERENCE's own evaluation metrics (src/loss_utils.h:30-128 — l1_loss, psnr,
psnr_gaussian_splatting, ssim via conv2d with the 11x11 window of gaussian() / create_window()) compiled unmodified against CPU LibTorch
(eval_driver.cpp).  Runs wherever /root/reference is mounted (no GPU):    python oracle/ref_build/make_eval_golden.py

Each file holds the two images and the reference's five numbers + its window.  tests/test_io_eval.py holds gaussian_lic_amd.loss (the host-side
mirror) and the C oracle's SSIM map to them on the CPU; tests/test_eval_gpu.py holds the device path to the oracle."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "_ref")
BIN = os.path.join(OUT, "eval_driver")

CASES = [("eval_3x70x50", 3, 50, 70, 71), ("eval_3x96x64", 3, 64, 96, 72), ("eval_1x33x17", 1, 17, 33, 73)]   # name, C, H, W, seed


def build():
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT, exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O1", "-std=c++17", "-w", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    for inc in cpp_extension.include_paths():
        cmd += ["-isystem", inc]
    # <fused-ssim/ssim.h> of the reference pulls in torch/extension.h (Python.h): the shim's header declares the same two functions without it
    cmd += ["-I", os.path.join(ROOT, "gaussian-lic_amd", "shim", "include"), "-I", "/root/reference/src", os.path.join(HERE, "eval_driver.cpp"), "-o", BIN, "-L", tlib, "-ltorch", "-ltorch_cpu", "-lc10", f"-Wl,-rpath,{tlib}"]
    subprocess.run(cmd, check=True)
    return BIN


def images(C, H, W, seed):
    rng = np.random.default_rng(seed)
    a = rng.random((C, H, W)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal((C, H, W)), 0.0, 1.0).astype(np.float32)
    return a, b


def main():
    exe = build()
    gdir = os.path.join(ROOT, "tests", "golden")
    for name, C, H, W, seed in CASES:
        a, b = images(C, H, W, seed)
        with tempfile.TemporaryDirectory() as d:
            np.array([C, H, W], np.float64).tofile(os.path.join(d, "meta.f64"))
            a.tofile(os.path.join(d, "img1.f32")); b.tofile(os.path.join(d, "img2.f32"))
            r = subprocess.run([exe, d], capture_output=True, text=True, check=True)
            vals = {l.split()[0]: np.float32(l.split()[1]) for l in r.stdout.splitlines() if len(l.split()) == 2}
            window = np.fromfile(os.path.join(d, "window.f32"), np.float32).reshape(C, 1, 11, 11)
        np.savez_compressed(os.path.join(gdir, name + ".npz"), img1=a, img2=b, window=window, **vals)
        print(name, {k: float(v) for k, v in vals.items()}, flush=True)


if __name__ == "__main__":
    main()
