// extend.hip — selection of the LiDAR points extend() turns into new Gaussians.
// Replaces, on the device, the host-side part of extend() (gaussian.cpp:536-603): projection with LibTorch ops, a
// device->host copy of every point, a CPU unordered_map<std::string, pair<int,float>> keyed by "x_y" strings for the
// nearest-point-per-pixel dedupe (:557-572), index_select round trips, and boolean-mask filtering.  Here:
//   extend_zbuffer_kernel  one 64-bit atomicMin per point on a per-pixel key {ordered bits of camera z, point index}:
//                          smaller z wins, the lower index wins ties — exactly the map's "replace only if strictly smaller".
//   extend_flag_kernel     winner of its pixel AND sensor range > 0 AND rendered alpha = 1 - final_T < 0.99 (:585-603)
//   (u32 scan)             exclusive positions of the survivors, ascending point index
//   extend_emit_kernel     writes the new parameter rows of :605-626 straight into the model's tensors at row P
#include "gslic_common.h"

namespace gslic {

__device__ __forceinline__ uint32_t ordered_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone map of float order onto unsigned order
}

__global__ __launch_bounds__(256) void extend_zbuffer_kernel(int n, const float* __restrict__ pts, const float* __restrict__ Rcw,
                                                             const float* __restrict__ tcw, float fx, float fy, float cx, float cy, int W,
                                                             int H, unsigned long long* __restrict__ zbuf, int32_t* __restrict__ pix)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float p0 = pts[3 * i], p1 = pts[3 * i + 1], p2 = pts[3 * i + 2];
    const float c0 = p0 * Rcw[0] + p1 * Rcw[1] + p2 * Rcw[2] + tcw[0];
    const float c1 = p0 * Rcw[3] + p1 * Rcw[4] + p2 * Rcw[5] + tcw[1];
    const float c2 = p0 * Rcw[6] + p1 * Rcw[7] + p2 * Rcw[8] + tcw[2];
    const float xf = floorf((c0 * fx) / c2 + cx), yf = floorf((c1 * fy) / c2 + cy);
    int id = -1;
    if (xf >= 0.f && xf < (float)W && yf >= 0.f && yf < (float)H) {
        id = (int)yf * W + (int)xf;
        atomicMin(&zbuf[id], ((unsigned long long)ordered_bits(c2) << 32) | (unsigned int)i);
    }
    pix[i] = id;
}

__global__ __launch_bounds__(256) void extend_flag_kernel(int n, const int32_t* __restrict__ pix, const unsigned long long* __restrict__ zbuf,
                                                          const float* __restrict__ depths_rsp, const float* __restrict__ final_T,
                                                          uint32_t* __restrict__ flags)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int id = pix[i];
    bool keep = false;
    if (id >= 0) keep = ((uint32_t)zbuf[id] == (uint32_t)i) && (depths_rsp[i] > 0.f) && ((1.0f - final_T[id]) < 0.99f);
    flags[i] = keep ? 1u : 0u;
}

__global__ __launch_bounds__(256) void extend_emit_kernel(int n, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos,
                                                          const float* __restrict__ pts, const float* __restrict__ colors,
                                                          const float* __restrict__ depths_rsp, float scaling_scale, float focal, int M,
                                                          float* xyz, float* dc, float* rest, float* opacity, float* scaling, float* rotation)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const size_t k = pos[i];
    const float sc = logf(scaling_scale * depths_rsp[i] / focal);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        xyz[3 * k + j] = pts[3 * i + j];
        dc[3 * k + j] = (colors[3 * i + j] - 0.5f) / 0.28209479177387814f;
        scaling[3 * k + j] = sc;
    }
    for (int j = 0; j < 3 * M; j++) rest[3 * (size_t)M * k + j] = 0.f;
    opacity[k] = logf(0.1f / (1.0f - 0.1f));  // inverse_sigmoid(0.1), general_utils.h
    reinterpret_cast<float4*>(rotation)[k] = make_float4(1.f, 0.f, 0.f, 0.f);
}

int extend_select(int n, const float* points, const float* depths_rsp, const float* Rcw, const float* tcw, float fx, float fy, float cx,
                  float cy, int W, int H, const float* final_T, gslic_alloc_fn alloc, void* ctx, uint32_t** flags_out,
                  uint32_t** pos_out, int32_t* count, hipStream_t s)
{
    *count = 0;
    if (n <= 0) return GSLIC_OK;
    const size_t N = (size_t)W * H;
    size_t bytes;
    {
        Carver c(nullptr);
        c.take<unsigned long long>(N); c.take<int32_t>(n); c.take<uint32_t>(n); c.take<uint32_t>(n); c.take<uint32_t>(scan_temp_elems(n));
        bytes = c.used(nullptr) + 256;
    }
    char* base = alloc(ctx, bytes);
    if (!base) return set_error(GSLIC_ERR_ALLOC, "extend scratch allocator returned NULL for %zu bytes", bytes);
    Carver c(base);
    unsigned long long* zbuf = c.take<unsigned long long>(N);
    int32_t* pix = c.take<int32_t>(n);
    uint32_t* flags = c.take<uint32_t>(n);
    uint32_t* pos = c.take<uint32_t>(n);
    uint32_t* stemp = c.take<uint32_t>(scan_temp_elems(n));
    GS_HIP(hipMemsetAsync(zbuf, 0xff, N * sizeof(unsigned long long), s));
    GS_LAUNCH(K_EXTEND, extend_zbuffer_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, points, Rcw, tcw, fx, fy, cx, cy, W, H, zbuf, pix);
    GS_LAUNCH(K_EXTEND, extend_flag_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, (const int32_t*)pix,
              (const unsigned long long*)zbuf, depths_rsp, final_T, flags);
    GS_TRY(scan_u32(flags, pos, (size_t)n, true, stemp, s));
    uint32_t last[2] = {0, 0};
    GS_HIP(hipMemcpyAsync(&last[0], pos + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(hipMemcpyAsync(&last[1], flags + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(hipStreamSynchronize(s));  // the host sizes the append from the count (the reference syncs many times here)
    *count = (int32_t)(last[0] + last[1]);
    *flags_out = flags;
    *pos_out = pos;
    return GSLIC_OK;
}

int extend_emit(int n, const uint32_t* flags, const uint32_t* pos, const float* points, const float* colors, const float* depths_rsp,
                float scaling_scale, float focal, int M, float* xyz, float* dc, float* rest, float* opacity, float* scaling,
                float* rotation, hipStream_t s)
{
    if (n <= 0) return GSLIC_OK;
    GS_LAUNCH(K_EXTEND, extend_emit_kernel, dim3(div_up(n, 256)), dim3(256), 0, s, n, flags, pos, points, colors, depths_rsp,
              scaling_scale, focal, M, xyz, dc, rest, opacity, scaling, rotation);
    return GSLIC_OK;
}

}  // namespace gslic
