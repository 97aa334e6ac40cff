// CUDA -> HIP name map for compiling the reference's .cu files UNMODIFIED for gfx950 as a test oracle
// (oracle/ref_build; never shipped).  Warp-level primitives are mapped to the 32-lane half of the wave64 the
// calling lane belongs to, which preserves the reference's 32-wide semantics (WARP_SIZE 32, 32-bit masks).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include <stdio.h>

#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaGetErrorString hipGetErrorString
#define cudaSuccess hipSuccess
#define cudaError_t hipError_t

#if defined(__HIPCC__)
__device__ inline unsigned int __ref_half_shift() { return (__lane_id() & 32u); }
__device__ inline unsigned int __ballot_sync(unsigned int /*mask*/, int pred)
{
    const unsigned long long b = __ballot(pred);
    return (unsigned int)(b >> __ref_half_shift());
}
template <typename T> __device__ inline T __shfl_sync(unsigned int /*mask*/, T v, int src) { return __shfl(v, src, 32); }
// __frcp_rn is provided by clang's HIP math header as the IEEE 1.0f/x.  HIP's __saturatef returns NaN for NaN;
// CUDA's returns 0 — keep the CUDA behaviour the reference was written against.
__device__ inline float __ref_saturatef(float x) { return (x > 0.0f) ? ((x < 1.0f) ? x : 1.0f) : 0.0f; }
#define __saturatef __ref_saturatef
#ifndef __trap
#define __trap() __builtin_trap()
#endif
#endif
