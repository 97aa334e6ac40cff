"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/gslic_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gslic_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gslic_[a-z0-9_]+)\s*\(", src)) - {"gslic_alloc_fn"})


def test_library_exports_every_declared_symbol():
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    from gaussian_lic_amd.build import build
    build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in gslic_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names
    assert _lib.lib().gslic_abi_version() == 8


def test_scratch_sizes_and_errors_without_gpu():
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    L = _lib.lib()
    # sizes grow monotonically and cover the documented per-element footprints
    assert L.gslic_geom_bytes(1000) >= 1000 * 56
    assert L.gslic_geom_bytes(2000) > L.gslic_geom_bytes(1000)
    assert L.gslic_img_bytes(1920, 1080) >= 8160 * (8 + 4 + 4 + 4096)
    assert L.gslic_binning_bytes(1000, 1) < L.gslic_binning_bytes(1000, 0)
    assert L.gslic_sample_bytes(10) >= 10 * 4096
    # argument validation happens before any device work
    prm = _lib.RasterParams(10, 5, 15, 64, 48, 1.0, 1.0, -1, 1, -1, 1, 1.0, 0, 0, 0, 0)
    R, B = ctypes.c_int32(7), ctypes.c_int32(7)
    rc = L.gslic_rasterize_forward(ctypes.byref(prm), *([_lib.ALLOC_FN(lambda c, n: 0), None] * 4), *([None] * 15),
                                   ctypes.byref(R), ctypes.byref(B), None)
    assert rc == -1 and b"degree" in L.gslic_last_error()
    prm.D = 3
    prm.P = 0
    rc = L.gslic_rasterize_forward(ctypes.byref(prm), *([_lib.ALLOC_FN(lambda c, n: 0), None] * 4), *([None] * 15),
                                   ctypes.byref(R), ctypes.byref(B), None)
    assert rc == 0 and R.value == 0 and B.value == 0


def test_exchange_entry_points_validate_without_gpu():
    """gslic_sh_grad_from_rgb(_adam) / gslic_rasterize_backward_rgb: argument errors are reported before any device work; P = 0 is a no-op."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    L = _lib.lib()
    assert L.gslic_sh_grad_from_rgb(0, 3, 15, 2, None, None, None, 0, None, None, 0, None) == 0
    assert L.gslic_sh_grad_from_rgb(10, 4, 15, 2, None, None, None, 0, None, None, 0, None) == -1        # SH degree > 3
    assert L.gslic_sh_grad_from_rgb(10, 3, 15, 0, None, None, None, 0, None, None, 0, None) == -1        # no views
    assert L.gslic_sh_grad_from_rgb(10, 3, 15, 2, None, None, None, 0, None, None, 0, None) == -1 and b"NULL" in L.gslic_last_error()
    ad = _lib.AdamFused()
    assert L.gslic_sh_grad_from_rgb_adam(0, 3, 15, 2, None, None, None, 0, None, ctypes.byref(ad), None, None, 0, None) == 0
    assert L.gslic_sh_grad_from_rgb_adam(10, 3, 15, 2, None, None, None, 0, None, None, None, None, 0, None) == -1
    prm = _lib.RasterParams(10, 3, 15, 64, 48, 1.0, 1.0, -1, 1, -1, 1, 1.0, 0, 0, 0, 1)
    rc = L.gslic_rasterize_backward_rgb(ctypes.byref(prm), 0, 0, *([None] * 12), *([None] * 4), None, *([None] * 5), 0.0, None)
    assert rc == -1 and b"dL_drgb" in L.gslic_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgslic_hip.so")
    with pytest.raises(_lib.GslicError):
        _lib.lib()


def test_tensor_allocator_is_freed_by_refcount_alone():
    """The scratch of a forward (~1 GB at 2M Gaussians) must die with the last reference, not whenever the cyclic GC runs:
    the allocator callback may not close a reference cycle over its owner."""
    import gc
    import weakref
    import torch
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    gc.disable()
    try:
        a = _lib.TensorAllocator(torch.device("cpu"))
        ptr = a.cb(None, 1000)
        assert ptr == a.tensor.data_ptr() and a.tensor.numel() == 1000
        big = a.cb(None, (1 << 20) + 1)
        assert big == a.tensor.data_ptr() and a.tensor.numel() == _lib.TensorAllocator.GRANULE
        wr_box, wr_alloc = weakref.ref(a._box), weakref.ref(a)
        del a
        assert wr_alloc() is None and wr_box() is None
    finally:
        gc.enable()


def test_large_scratch_requests_round_up_geometrically():
    """A growing map must not pay a hipMalloc at every append: above 64 MB the granule is min(half the largest power of two in the request,
    256 MB) — gslic_scratch_round_up, ONE rule in the library that _lib.TensorAllocator and the C++ shim's resize callbacks all call.
    Sizes only: nothing this large is allocated here."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    G = _lib.TensorAllocator.GRANULE
    seen = []

    import torch
    real_empty = torch.empty
    try:
        torch.empty = lambda n, **kw: (seen.append(int(n)), real_empty(0, dtype=torch.uint8))[1]
        a = _lib.TensorAllocator(torch.device("cpu"))
        for n in (1000, (1 << 20) + 1, 50 << 20, (64 << 20) + 1, 366 << 20, 457 << 20, 597 << 20, (1 << 30) + 5, (2 << 30) + 5):
            a.cb(None, n)
    finally:
        torch.empty = real_empty
    assert seen[-9:] == [1000, G, 64 << 20, 96 << 20, 384 << 20, 512 << 20, 768 << 20, 5 << 28, 9 << 28]
    # 1.5M -> 2.0M Gaussians: the binning buffer (61 B per instance, 4.5M -> 6M instances) takes ONE size on the way, with 32 MB granules three
    seen.clear()
    try:
        torch.empty = lambda n, **kw: (seen.append(int(n)), real_empty(0, dtype=torch.uint8))[1]
        for r in range(4_500_000, 6_000_001, 50_000):
            a.cb(None, 61 * r)
    finally:
        torch.empty = real_empty
    assert len(set(seen)) == 1 and len({(61 * r + G - 1) // G for r in range(4_500_000, 6_000_001, 50_000)}) == 3


def test_scratch_round_up_worst_case_overhead():
    """the rounding must not turn a map that fits into an out-of-memory.  Over-allocation is below 50 % up to 512 MB (where the
    granule is half a power of two) and below 256 MB in absolute terms beyond; never below the request, idempotent, monotonic."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    up = _lib.lib().gslic_scratch_round_up
    prev = 0
    sizes = [1, 1 << 20, (1 << 20) + 1] + [int(1.07 ** k * (1 << 20)) for k in range(1, 160)] + [(k << 28) + 1 for k in range(1, 200, 7)]
    for n in sorted(sizes):
        r = up(n)
        assert r >= n and up(r) == r and r >= prev, (n, r)
        prev = r
        if n > (512 << 20):
            assert r - n < (256 << 20), (n, r)
        elif n > (1 << 20):
            assert r < 1.5 * n + (32 << 20), (n, r)
    assert up((2 << 30) + 5) == (9 << 28) and up((100 << 30) + 1) == (100 << 30) + (256 << 20)


def test_binning_mode_switch_without_gpu():
    """gslic_set_binning_mode touches host state only: returns the previous mode, ignores values outside 0..2; gslic_get_binning_path reports
    "none" for a thread that has not run a forward with instances"""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    old = _lib.set_binning_mode("radix")
    try:
        assert _lib.set_binning_mode("atomic") == "radix"
        assert _lib.lib().gslic_set_binning_mode(7) == 2 and _lib.lib().gslic_set_binning_mode(-1) == 2   # (only reads the mode)
        assert _lib.set_binning_mode("auto") == "atomic"
    finally:
        _lib.set_binning_mode(old)
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(_lib.binning_path()))   # (a fresh host thread: the state is per thread)
    t.start(); t.join()
    assert seen == [("none", 0, 0)]


def test_binning_mode_is_safe_to_set_while_other_threads_read_it():
    """The mode is an atomic and setting it bumps an epoch every thread's auto state follows (a plain global + the calling
    thread's state only).  Host-side only: hammer set / read from four threads, the value is always one of the three modes."""
    import threading
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import _lib
    L = _lib.lib()
    old = _lib.set_binning_mode("auto")
    bad = []

    def work(k):
        for i in range(2000):
            v = L.gslic_set_binning_mode((i + k) % 3)
            if v not in (0, 1, 2) or L.gslic_set_binning_mode(-1) not in (0, 1, 2):
                bad.append(v)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    _lib.set_binning_mode(old)
    assert not bad
