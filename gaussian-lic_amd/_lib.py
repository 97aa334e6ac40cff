"""ctypes binding of libgslic_hip.so (the C-ABI of include/gslic_hip.h).

There is NO fallback: if the library is missing or fails to load, every operator of this package raises.
torch is imported first so that the process has exactly one HIP runtime (torch's bundled libamdhip64.so.7;
libgslic_hip.so's DT_NEEDED entry resolves to the already-loaded object).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSLIC_HIP_LIB") or os.path.join(_HERE, "libgslic_hip.so")   # (override: A/B kernel experiments, tools/ab/)

ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


class RasterParams(ctypes.Structure):
    _fields_ = [("P", ctypes.c_int32), ("D", ctypes.c_int32), ("M", ctypes.c_int32), ("width", ctypes.c_int32),
                ("height", ctypes.c_int32), ("tan_fovx", ctypes.c_float), ("tan_fovy", ctypes.c_float),
                ("limx_neg", ctypes.c_float), ("limx_pos", ctypes.c_float), ("limy_neg", ctypes.c_float),
                ("limy_pos", ctypes.c_float), ("scale_modifier", ctypes.c_float), ("prefiltered", ctypes.c_int32),
                ("debug", ctypes.c_int32), ("no_color", ctypes.c_int32), ("raw_params", ctypes.c_int32), ("tie_rank", ctypes.c_void_p)]


class AdamGroup(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("lr", ctypes.c_float), ("M", ctypes.c_uint32)]


class AdamFused(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p * 6), ("exp_avg", ctypes.c_void_p * 6), ("exp_avg_sq", ctypes.c_void_p * 6),
                ("lr", ctypes.c_float * 6), ("b1", ctypes.c_float), ("b2", ctypes.c_float), ("eps", ctypes.c_float), ("visible_out", ctypes.c_void_p)]


EXPORTS = [
    "gslic_rasterize_forward", "gslic_rasterize_backward", "gslic_rasterize_backward_adam", "gslic_rasterize_backward_camera", "gslic_adam_update", "gslic_adam_update_groups",
    "gslic_fusedssim_forward", "gslic_fusedssim_backward", "gslic_knn_mean_dist2", "gslic_abi_version",
    "gslic_last_error", "gslic_geom_bytes", "gslic_img_bytes", "gslic_binning_bytes", "gslic_sample_bytes",
    "gslic_profile_enable", "gslic_profile_reset", "gslic_profile_collect", "gslic_profile_num_kernels",
    "gslic_profile_kernel_name", "gslic_profile_get", "gslic_debug_export", "gslic_extend_select", "gslic_extend_emit", "gslic_loss_partials_count", "gslic_l1_ssim_loss_forward", "gslic_l1_ssim_loss_forward_backward",
    "gslic_l1_ssim_loss_backward", "gslic_set_math_mode", "gslic_set_binning_mode", "gslic_rasterize_forward_capacity", "gslic_rasterize_backward_rgb",
    "gslic_rasterize_backward_rgb_rows", "gslic_sh_grad_from_rgb", "gslic_sh_grad_from_rgb_adam", "gslic_rasterize_backward_rgb_payload",
    "gslic_sh_grad_from_rgb_adam_all", "gslic_get_binning_path", "gslic_scratch_round_up",
]

_lib = None


class GslicError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GslicError(f"{LIB_PATH} is missing: run `python gaussian-lic_amd/build.py` (hipcc, gfx950). "
                         "There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    vp, i32, f32, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_uint32
    L.gslic_last_error.restype = ctypes.c_char_p
    L.gslic_profile_kernel_name.restype = ctypes.c_char_p
    L.gslic_profile_kernel_name.argtypes = [i32]
    L.gslic_profile_get.argtypes = [i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
    for n in ("gslic_geom_bytes", "gslic_img_bytes", "gslic_binning_bytes", "gslic_sample_bytes"):
        getattr(L, n).restype = ctypes.c_size_t
    L.gslic_geom_bytes.argtypes = [i32]
    L.gslic_img_bytes.argtypes = [i32, i32]
    L.gslic_binning_bytes.argtypes = [i32, i32]
    L.gslic_sample_bytes.argtypes = [i32]
    L.gslic_rasterize_forward.argtypes = (
        [ctypes.POINTER(RasterParams)] + [ALLOC_FN, vp] * 4 + [vp] * 12 + [vp, vp, vp] +
        [ctypes.POINTER(i32), ctypes.POINTER(i32), vp])
    L.gslic_rasterize_forward_capacity.argtypes = (
        [ctypes.POINTER(RasterParams)] + [vp, ctypes.c_size_t] * 4 + [vp] * 12 + [vp, vp, vp] +
        [ctypes.POINTER(i32), ctypes.POINTER(i32), vp, vp])
    L.gslic_rasterize_backward.argtypes = (
        [ctypes.POINTER(RasterParams), i32, i32] + [vp] * 12 + [vp] * 4 + [vp] + [vp] * 10 + [f32, vp])
    L.gslic_rasterize_backward_adam.argtypes = (
        [ctypes.POINTER(RasterParams), i32, i32] + [vp] * 12 + [vp] * 4 + [vp] + [vp] * 6 + [f32, ctypes.POINTER(AdamFused), vp])
    L.gslic_rasterize_backward_camera.argtypes = (
        [ctypes.POINTER(RasterParams), i32, i32] + [vp] * 12 + [vp] * 4 + [vp] + [vp] * 10 + [f32, vp, vp, vp, vp])
    L.gslic_rasterize_backward_rgb.argtypes = (
        [ctypes.POINTER(RasterParams), i32, i32] + [vp] * 12 + [vp] * 4 + [vp] + [vp] * 5 + [f32, vp])
    L.gslic_rasterize_backward_rgb_rows.argtypes = (
        [ctypes.POINTER(RasterParams), i32, i32] + [vp] * 12 + [vp] * 4 + [vp] + [vp] * 5 + [f32, i32, i32, i32, vp])
    L.gslic_rasterize_backward_rgb_payload.argtypes = (
        [ctypes.POINTER(RasterParams), i32, i32] + [vp] * 12 + [vp] * 4 + [vp] + [vp] * 5 + [f32, vp, vp, vp])
    L.gslic_sh_grad_from_rgb_adam_all.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, vp, ctypes.c_int64, vp, ctypes.POINTER(AdamFused), vp, vp, vp, vp,
                                                  ctypes.c_int64, vp]
    L.gslic_sh_grad_from_rgb.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, vp, vp, ctypes.c_int64, vp]
    L.gslic_sh_grad_from_rgb_adam.argtypes = [i32, i32, i32, i32, vp, vp, vp, i32, vp, ctypes.POINTER(AdamFused), vp, vp, ctypes.c_int64, vp]
    L.gslic_adam_update.argtypes = [vp, vp, vp, vp, vp, f32, f32, f32, f32, u32, u32, vp]
    L.gslic_adam_update_groups.argtypes = [ctypes.POINTER(AdamGroup), i32, vp, f32, f32, f32, u32, vp]
    L.gslic_fusedssim_forward.argtypes = [i32, i32, i32, i32, f32, f32] + [vp] * 6 + [vp]
    L.gslic_fusedssim_backward.argtypes = [i32, i32, i32, i32, f32, f32] + [vp] * 7 + [vp]
    L.gslic_knn_mean_dist2.argtypes = [i32, vp, vp, ALLOC_FN, vp, vp]
    L.gslic_extend_select.argtypes = [i32, vp, vp, vp, vp, f32, f32, f32, f32, i32, i32, vp, ALLOC_FN, vp,
                                      ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(i32), vp]
    L.gslic_extend_emit.argtypes = [i32, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.gslic_debug_export.argtypes = [ctypes.POINTER(RasterParams), i32, i32] + [vp] * 4 + [vp] * 10 + [vp]
    L.gslic_loss_partials_count.restype = ctypes.c_int64
    L.gslic_loss_partials_count.argtypes = [i32, i32, i32, i32]
    L.gslic_l1_ssim_loss_forward.argtypes = [i32, i32, i32, i32, f32, f32] + [vp] * 7 + [vp]
    L.gslic_l1_ssim_loss_backward.argtypes = [i32, i32, i32, i32, f32] + [vp] * 6 + [vp]
    L.gslic_l1_ssim_loss_forward_backward.argtypes = [i32, i32, i32, i32, f32, f32, f32] + [vp] * 8 + [vp]
    L.gslic_set_math_mode.argtypes = [i32]
    L.gslic_set_binning_mode.argtypes = [i32]
    L.gslic_get_binning_path.argtypes = [ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.gslic_scratch_round_up.restype = ctypes.c_size_t
    L.gslic_scratch_round_up.argtypes = [ctypes.c_size_t]
    if L.gslic_abi_version() != 8:
        raise GslicError("libgslic_hip.so ABI version mismatch")
    _lib = L
    return L


def set_math_mode(strict):
    """gslic_set_math_mode: True (default) = the blend kernels in the reference's arithmetic (bit-identical image), False = fast (opt-in; GSLIC_FAST_MATH=1).
    Returns the previous mode."""
    return bool(lib().gslic_set_math_mode(int(bool(strict))))


def set_binning_mode(mode):
    """gslic_set_binning_mode: "auto" (default), "radix" or "atomic" — how the forward groups the instances by tile (same lists bit for bit).
    Returns the previous mode's name."""
    names = ("auto", "radix", "atomic")
    return names[lib().gslic_set_binning_mode(names.index(mode))]


def binning_path():
    """gslic_get_binning_path: ("none" | "radix" | "atomic", sampled global atomics, sampled instances) of the calling thread's last forward
    that had instances — which grouping ran, and what `auto` decides on (atomics <= 0.4 * instances keeps the atomic path)."""
    a, n = ctypes.c_uint32(0), ctypes.c_uint32(0)
    path = lib().gslic_get_binning_path(ctypes.byref(a), ctypes.byref(n))
    return ("none", "radix", "atomic")[path], a.value, n.value


def check(rc):
    if rc != 0:
        raise GslicError(f"libgslic_hip error {rc}: {lib().gslic_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None / empty tensor -> NULL, like .data<float>() of an empty tensor)."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def current_stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _AllocBox:
    """What the allocator callback closes over: the device and the tensor it handed out.  Deliberately NOT the TensorAllocator itself —
    a ctypes thunk that referenced a bound method of its owner would form a reference cycle, and the ~1 GB of scratch of a 2M-Gaussian
    step would stay pinned until the cyclic collector happened to run."""
    __slots__ = ("device", "tensor", "__weakref__")

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)


class TensorAllocator:
    """Allocator callback backed by torch's caching allocator — the role of resizeFunctional
    (rasterize_points.cu:40-48): one growable uint8 tensor per scratch buffer.  Freed by reference counting alone."""

    GRANULE = 32 << 20  # large requests are rounded up (gslic_scratch_round_up: the rule lives in the library, the C++ shim's callbacks call it too) so
                        # that step-to-step drift of R / B (the Gaussians move) keeps hitting the same cached block instead of forcing a fresh
                        # hipMalloc; above 64 MB the granule is min(half the largest power of two in the request, 256 MB) (366 MB -> 384, 457 -> 512,
                        # 2.1 GB -> 2.25): a map that GROWS (extend() every few iterations) crosses a granule once per ~1.3x of growth instead of
                        # at every append — a hipMalloc is milliseconds, a step is 1.8 — and a large map does not over-allocate by half

    def __init__(self, device):
        box = self._box = _AllocBox(device)
        round_up = lib().gslic_scratch_round_up

        def _alloc(_ctx, nbytes):
            box.tensor = torch.empty(int(round_up(int(nbytes))), dtype=torch.uint8, device=box.device)
            return box.tensor.data_ptr()

        self.cb = ALLOC_FN(_alloc)

    @property
    def device(self):
        return self._box.device

    @property
    def tensor(self):
        return self._box.tensor


# ------------------------------------------------------------------------------------------- profiling
def profile_enable(on=True, only=None):
    """on=False: off.  only=None: time every kernel; only=[names]: time just those kernels."""
    L = lib()
    if not on:
        check(L.gslic_profile_enable(0))
        return
    if only is None:
        check(L.gslic_profile_enable(1))
        return
    names = [L.gslic_profile_kernel_name(i).decode() for i in range(L.gslic_profile_num_kernels())]
    mask = 0
    for n in only:
        mask |= 1 << names.index(n)
    check(L.gslic_profile_enable(mask if mask not in (0, 1) else (mask | (1 << 30))))


def profile_reset():
    check(lib().gslic_profile_reset())


def profile_collect():
    """Synchronise and return {kernel_name: (total_ms, launches)} accumulated since the last reset."""
    L = lib()
    check(L.gslic_profile_collect())
    out = {}
    for i in range(L.gslic_profile_num_kernels()):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        check(L.gslic_profile_get(i, ctypes.byref(ms), ctypes.byref(n)))
        if n.value:
            out[L.gslic_profile_kernel_name(i).decode()] = (ms.value, n.value)
    return out
