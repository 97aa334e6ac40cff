"""Compiles the REFERENCE's own rasterizer kernels (unmodified sources, where they lie under /root/reference)
for gfx950 into oracle/_ref/libref_hip.so — a checker used to generate tests/golden/ on the MI355X and to
validate the CPU oracle.  Never shipped; oracle/_ref/ is git-ignored (but travels to the GPU box).

The reference needs glm, cub, cooperative_groups and CUDA warp intrinsics; oracle/ref_build/shim/ provides
stand-ins written for this repo (glm subset, cub -> hipcub alias, *_sync warp intrinsics mapped to the 32-lane
half-wave).  -ffp-contract=off keeps the arithmetic in source order (the canonical order of the oracle).
No reference source is copied: the compiler reads the files in place.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/rasterizer/cuda_rasterizer"
OUT = os.path.join(os.path.dirname(HERE), "_ref")
SRCS = ["backward.cu", "adam.cu"]            # compiled directly from /root/reference
WRAPS = ["wrap_forward.hip", "wrap_rasterizer_impl.hip",  # #include the reference .cu after re-defining WARP_SIZE
         "wrap_ssim.hip"]  # fused-ssim kernels.  (simple_knn.cu cannot be compiled in place: its `<< <grid, block >> >` launches only parse with nvcc)


def main():
    if not os.path.isdir(REF):
        print("reference not mounted; skipping oracle/_ref build")
        return 0
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libref_hip.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-fno-fast-math", "-DHIP_DISABLE_WARP_SYNC_BUILTINS", "-Wno-everything", "-I", os.path.join(HERE, "shim"), "-I", REF,
           "-include", os.path.join(HERE, "shim", "ref_prelude.h")]
    for s in SRCS:
        cmd += ["-x", "hip", os.path.join(REF, s)]
    for s in WRAPS:
        cmd += ["-x", "hip", os.path.join(HERE, s)]
    cmd += ["-x", "hip", os.path.join(HERE, "ref_driver.hip"), "-o", lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-4000:] + r.stderr[-8000:])
        print("oracle/_ref build FAILED")
        return 1
    print(lib)
    return 0


if __name__ == "__main__":
    sys.exit(main())
