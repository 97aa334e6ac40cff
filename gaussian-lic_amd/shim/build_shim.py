"""Builds libgslic_torch_shim.so: plain g++ against LibTorch headers, linked to libgslic_hip.so by C-ABI only
(no hipcc, no hipify, no torch.utils.cpp_extension JIT: the .so stays in-tree and travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libgslic_torch_shim.so")
# gslic_stream.h: hand LibTorch's CURRENT HIP stream to the C-ABI (c10::hip::getCurrentHIPStream: needs the HIP headers and c10_hip)
STREAM_FLAGS = ["-DGSLIC_SHIM_CURRENT_STREAM", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM", "-isystem", "/opt/rocm/include"]


def build(force=False):
    import torch
    from torch.utils import cpp_extension
    src = os.path.join(HERE, "gslic_torch_shim.cpp")
    deps = [src, os.path.join(PKG, "..", "include", "gslic_hip.h")] + [os.path.join(HERE, "include", p) for p in
                                                                       ("rasterizer/rasterize_points.h", "fused-ssim/ssim.h", "simple-knn/spatial.h", "gslic_stream.h")]
    deps.append(os.path.join(PKG, "libgslic_hip.so"))
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(d) for d in deps):
        return OUT
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + STREAM_FLAGS
    for inc in cpp_extension.include_paths():
        cmd += ["-isystem", inc]
    cmd += ["-I", os.path.join(HERE, "include"), src, "-o", OUT, "-L", PKG, "-lgslic_hip", f"-Wl,-rpath,$ORIGIN", "-L", tlib,
            "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", f"-Wl,-rpath,{tlib}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-3000:] + r.stderr[-6000:])
        raise RuntimeError("shim build failed")
    return OUT


REF_SRC = "/root/reference/src"
CHECK = os.path.join(PKG, "dropin_check")
CHECK_GROUPS = os.path.join(PKG, "dropin_check_groups")
CHECK_DIST = os.path.join(PKG, "dropin_check_dist")
CHECK_RENDER_REF = os.path.join(PKG, "dropin_check_render_ref")   # render() of the REFERENCE's renderer.cpp (on the stand-in Camera / GaussianModel)
CHECK_RENDER = os.path.join(PKG, "dropin_check_render")           # render() of shim/renderer.cpp, the drop-in replacement
CHECK_RENDER_LOSS = os.path.join(PKG, "dropin_check_render_loss") # ... + the optional one-node loss (loss_utils_fused.h)
CHECK_RENDER_REFHOST = os.path.join(PKG, "dropin_check_render_refhost")   # EVERY host line the reference's: its renderer.cpp AND its optim_utils.h (six clones + six launches)


def build_dropin_check(force=False, groups=False):
    """Compiles the REFERENCE's own host code (rasterizer/rasterizer.cpp + headers, read in place) together with
    dropin_check.cpp and links it against the shim: the drop-in claim, exercised.  Needs /root/reference; the binary stays
    in-tree (git-ignored) and travels to the GPU box.  groups=True builds the same program with shim/include ahead of the
    reference's src/ on the include path, i.e. with this repo's optim_utils.h (one Adam launch per step) instead of the reference's."""
    if not os.path.isdir(REF_SRC):
        return None
    if groups == "dist":   # + the N > 1 exchange step over c10d / RCCL (shim/include/gslic_dist.h) between loss.backward() and step()
        return _build_check(CHECK_DIST, ["-DGSLIC_DIST", "-DUSE_C10D_NCCL", "-DUSE_ROCM", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(HERE, "include"),
                                         "-I", os.path.join(HERE, "include", "nccl_fwd"), "-isystem", "/opt/rocm/include"], force)
    if groups == "render_refhost":
        # nothing of this repository's on the include path but the stand-in Camera / GaussianModel types: the reference's renderer.cpp, rasterizer.cpp,
        # loss_utils.h and optim_utils.h (SparseGaussianAdam::custom_step: grad.clone() + adamUpdate per group, :102-137) as they are — what bench.py
        # reports as the drop-in host's rate
        return _build_check(CHECK_RENDER_REFHOST, ["-DGSLIC_CHECK_RENDER", "-I", os.path.join(HERE, "standin")], force,
                            extra_src=[os.path.join(REF_SRC, "rasterizer", "renderer.cpp")])
    if groups in ("render_ref", "render", "render_loss"):
        # this repo's optim_utils.h (one Adam launch) in all three, so that the renderer / the loss is the only thing that differs
        inc = ["-DGSLIC_CHECK_RENDER", "-I", os.path.join(HERE, "standin"), "-I", os.path.join(HERE, "include")]
        if groups == "render_ref":
            return _build_check(CHECK_RENDER_REF, inc, force, extra_src=[os.path.join(REF_SRC, "rasterizer", "renderer.cpp")])
        if groups == "render_loss":
            inc = ["-DGSLIC_ONE_NODE_LOSS"] + inc
        return _build_check(CHECK_RENDER_LOSS if groups == "render_loss" else CHECK_RENDER, inc, force, extra_src=[os.path.join(HERE, "renderer.cpp")])
    if groups:
        return _build_check(CHECK_GROUPS, ["-I", os.path.join(HERE, "include")], force)
    return _build_check(CHECK, [], force)


def _build_check(CHECK, first_includes, force, extra_src=()):
    import sysconfig
    import torch
    from torch.utils import cpp_extension
    src = os.path.join(HERE, "dropin_check.cpp")
    newest = max([os.path.getmtime(src), os.path.getmtime(OUT), os.path.getmtime(os.path.join(HERE, "include", "optim_utils.h")),
                  os.path.getmtime(os.path.join(HERE, "include", "gslic_dist.h")), os.path.getmtime(os.path.join(HERE, "include", "loss_utils_fused.h"))] +
                 [os.path.getmtime(x) for x in extra_src])
    if not force and os.path.exists(CHECK) and os.path.getmtime(CHECK) > newest:
        return CHECK
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O1", "-std=c++17", "-w", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + STREAM_FLAGS
    for inc in cpp_extension.include_paths():
        cmd += ["-isystem", inc]
    cmd += first_includes
    cmd += ["-isystem", sysconfig.get_paths()["include"], "-I", REF_SRC, "-I", os.path.join(REF_SRC, "rasterizer"), src, os.path.join(REF_SRC, "rasterizer", "rasterizer.cpp"), *extra_src,
            "-o", CHECK, "-L", PKG, "-lgslic_torch_shim", "-lgslic_hip", "-Wl,-rpath,$ORIGIN", "-L", tlib, "-ltorch", "-ltorch_cpu",
            "-ltorch_hip", "-lc10", "-lc10_hip", f"-Wl,-rpath,{tlib}", "-Wl,--no-as-needed", "-ltorch_hip", "-Wl,--as-needed",
            "-L", os.path.join(sysconfig.get_config_var("LIBDIR") or "/usr/lib"), f"-lpython{sysconfig.get_python_version()}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-3000:] + r.stderr[-6000:])
        raise RuntimeError("dropin_check build failed")
    return CHECK


CHECK_FUSED = os.path.join(PKG, "fused_check")


def build_fused_check(force=False):
    """fused_check: the fused training step from C++ (shim/include/gslic_fused.h + fused_check.cpp) — LibTorch and libgslic_hip.so only,
    no reference source, so it builds anywhere this repository does."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension
    src = os.path.join(HERE, "fused_check.cpp")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "include", "gslic_fused.h")),
                 os.path.getmtime(os.path.join(PKG, "..", "include", "gslic_hip.h")), os.path.getmtime(os.path.join(PKG, "libgslic_hip.so")))
    if not force and os.path.exists(CHECK_FUSED) and os.path.getmtime(CHECK_FUSED) > newest:
        return CHECK_FUSED
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O1", "-std=c++17", "-w", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] + STREAM_FLAGS
    for inc in cpp_extension.include_paths():
        cmd += ["-isystem", inc]
    cmd += ["-isystem", sysconfig.get_paths()["include"], "-I", os.path.join(HERE, "include"), src, "-o", CHECK_FUSED, "-L", PKG, "-lgslic_hip",
            "-Wl,-rpath,$ORIGIN", "-L", tlib, "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", f"-Wl,-rpath,{tlib}",
            "-Wl,--no-as-needed", "-ltorch_hip", "-Wl,--as-needed",
            "-L", os.path.join(sysconfig.get_config_var("LIBDIR") or "/usr/lib"), f"-lpython{sysconfig.get_python_version()}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-3000:] + r.stderr[-6000:])
        raise RuntimeError("fused_check build failed")
    return CHECK_FUSED


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_dropin_check(force="--force" in sys.argv))
    print(build_dropin_check(force="--force" in sys.argv, groups=True))
    print(build_dropin_check(force="--force" in sys.argv, groups="dist"))
    for g in ("render_ref", "render", "render_loss", "render_refhost"):
        print(build_dropin_check(force="--force" in sys.argv, groups=g))
    print(build_fused_check(force="--force" in sys.argv))
