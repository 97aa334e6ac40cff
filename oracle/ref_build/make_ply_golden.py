"""Generates tests/golden/savemap_*.ply with the REFERENCE's own PLY writer: src/tinyply.{h,cpp} compiled in place (g++, no GPU)
together with ply_driver.cpp, which replays saveMap's call sequence (src/gaussian.cpp:306-397).  Runs wherever /root/reference is
mounted:    python oracle/ref_build/make_ply_golden.py
The binary goes to oracle/_ref/ (git-ignored); the few-KB .ply files are committed and gaussian-lic_amd/io_ply.save_map must
reproduce them byte for byte (tests/test_io_eval.py)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SRC = "/root/reference/src"
OUT = os.path.join(os.path.dirname(HERE), "_ref")
BIN = os.path.join(OUT, "ply_writer")
CASES = [("savemap_64_d3", 64, 3, 0, 7), ("savemap_40_d0", 40, 0, 0, 8), ("savemap_50_d3_skybox10", 50, 3, 10, 9)]  # name, P, degree, skybox rows, seed


def build():
    os.makedirs(OUT, exist_ok=True)
    cmd = ["g++", "-O1", "-std=c++17", "-w", "-I", REF_SRC, os.path.join(HERE, "ply_driver.cpp"), os.path.join(REF_SRC, "tinyply.cpp"), "-o", BIN]
    subprocess.run(cmd, check=True)
    return BIN


def model_for(P, deg, seed):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd.synthetic import random_scene
    return random_scene(P, 160, 120, sh_degree=deg, seed=seed)


def main():
    if not os.path.isdir(REF_SRC):
        print("reference not mounted; nothing generated")
        return 0
    build()
    gdir = os.path.join(ROOT, "tests", "golden")
    for name, P, deg, sky, seed in CASES:
        raw = model_for(P, deg, seed)
        s = slice(sky, None)
        # what saveMap hands to tinyply (gaussian.cpp:309-316): rows from skybox_points_num on, features transposed to [P,3,K] and flattened
        arrs = dict(xyz=raw["xyz"][s], f_dc=raw["features_dc"][s].transpose(1, 2).flatten(1), f_rest=raw["features_rest"][s].transpose(1, 2).flatten(1),
                    opacity=raw["opacity"][s], scale=raw["scaling"][s], rotation=raw["rotation"][s])
        n = arrs["xyz"].shape[0]
        M = raw["features_rest"].shape[1]
        with tempfile.TemporaryDirectory() as d:
            paths = []
            for k in ("xyz", "f_dc", "f_rest", "opacity", "scale", "rotation"):
                pth = os.path.join(d, k + ".f32")
                np.ascontiguousarray(arrs[k].contiguous().numpy(), "<f4").tofile(pth)
                paths.append(pth)
            out = os.path.join(gdir, name + ".ply")
            subprocess.run([BIN, out, str(n), str(M)] + paths, check=True)
        print(name, os.path.getsize(out), "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
