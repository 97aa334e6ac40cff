"""gaussian-lic_amd — MI355X-native (gfx950) differentiable 3D-Gaussian-splatting hot path of Gaussian-LIC.

The product is csrc/ (hand-written HIP kernels behind the C-ABI of include/gslic_hip.h, built into
libgslic_hip.so) plus a thin host-side mirror of the reference's operator interface.  There is no CPU
fallback: every operator raises if the HIP library is missing.
"""
from . import camera, synthetic  # noqa: F401  (CPU-only helpers: input contract + seeded scenes)

__all__ = ["camera", "synthetic"]
