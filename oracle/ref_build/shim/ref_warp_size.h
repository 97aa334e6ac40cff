// The reference computes `0xFFFFFFFFU >> (WARP_SIZE - lane_idx)` (rasterizer_impl.cu:116), which for lane 0 is a shift
// by 32: undefined in C++, 0 on NVIDIA hardware (shift amounts clamp), but v >> 0 on AMD hardware (shift amounts wrap
// mod 32).  To run the reference's own source with the semantics it was written against, WARP_SIZE is re-defined — after
// forward.h defined it — as a value that still converts to the integer 32 everywhere but turns `WARP_SIZE - lane` into a
// tagged shift amount whose `>>` clamps like CUDA.  Only oracle/ref_build uses this.
#pragma once
#include <stdint.h>
struct RefShiftAmt { unsigned s; };
struct RefWarpSize {
    __host__ __device__ constexpr operator int() const { return 32; }
};
__host__ __device__ inline RefShiftAmt operator-(RefWarpSize, unsigned lane) { return RefShiftAmt{32u - lane}; }
__host__ __device__ inline unsigned operator>>(unsigned v, RefShiftAmt a) { return a.s >= 32u ? 0u : (v >> a.s); }
