// knn.hip — mean squared distance to the 3 nearest neighbours (replaces SimpleKNN::knn and its kernels,
// src/simple-knn/simple_knn.cu:45-221).  Same plan as the reference — Morton order, 1024-point boxes, exact
// search with box pruning — but with no host round trips (the bounding box stays on the device), no
// cudaMalloc inside (scratch from the caller), the library's own radix sort, and the points pre-gathered in
// Morton order so that the inner loops read contiguous memory that a wave broadcasts.
#include "gslic_common.h"
#include <float.h>

namespace gslic {

static constexpr int BOX = 1024;

// min/max over all points with the reference's (0,0,0) initial value (simple_knn.cu:191-199)
__global__ __launch_bounds__(256) void knn_minmax_kernel(int P, const float* __restrict__ pts, float* __restrict__ mm /*[6] pre-zeroed*/)
{
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = pts[3 * (size_t)i + k];
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
        }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], d, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // all values are <= 0 (min) / >= 0 (max): signed-int ordering of the float bits is monotone for each
            atomicMax(reinterpret_cast<unsigned int*>(mm + k), __float_as_uint(mn[k]));      // more negative = larger bits
            atomicMax(reinterpret_cast<unsigned int*>(mm + 3 + k), __float_as_uint(mx[k]));  // non-negative floats
        }
    }
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ mm,
                                                         uint32_t* __restrict__ codes, uint32_t* __restrict__ ids)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mn = mm[k], mx = mm[3 + k];
        const float f = ((pts[3 * (size_t)i + k] - mn) / (mx - mn)) * (float)((1 << 10) - 1);
        // float -> uint32 conversion: NaN/negative -> 0 (degenerate axis), matching CUDA's saturating cast
        c[k] = prep_morton((f > 0.f) ? (uint32_t)fminf(f, 4294967040.f) : 0u);
    }
    codes[i] = (uint32_t)(c[0] | (c[1] << 1) | (c[2] << 2));
    ids[i] = (uint32_t)i;
}

// gather points in Morton order + per-box AABB (boxMinMax, simple_knn.cu:78-117)
__global__ __launch_bounds__(BOX) void knn_boxes_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                        float* __restrict__ sorted, float* __restrict__ boxes /*[nb][6]*/)
{
    __shared__ float red[6][BOX / 64];
    const int i = blockIdx.x * BOX + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < P) {
        const uint32_t src = order[i];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float v = pts[3 * (size_t)src + k];
            sorted[3 * (size_t)i + k] = v;
            mn[k] = mx[k] = v;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], d, 64));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[k][threadIdx.x >> 6] = mn[k]; red[3 + k][threadIdx.x >> 6] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < BOX / 64; w++) v = (threadIdx.x < 3) ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        boxes[6 * (size_t)blockIdx.x + threadIdx.x] = v;
    }
}

__device__ __forceinline__ void update3(float (&best)[3], float dist)
{
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}

// boxMeanDist (simple_knn.cu:147-183)
__global__ __launch_bounds__(256) void knn_search_kernel(int P, const float* __restrict__ sorted, const uint32_t* __restrict__ order,
                                                         const float* __restrict__ boxes, float* __restrict__ dists)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float x = sorted[3 * (size_t)idx], y = sorted[3 * (size_t)idx + 1], z = sorted[3 * (size_t)idx + 2];
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    const int lo = idx - 3 < 0 ? 0 : idx - 3, hi = idx + 3 > P - 1 ? P - 1 : idx + 3;
    for (int i = lo; i <= hi; i++) {
        if (i == idx) continue;
        const float dx = sorted[3 * (size_t)i] - x, dy = sorted[3 * (size_t)i + 1] - y, dz = sorted[3 * (size_t)i + 2] - z;
        update3(best, dx * dx + dy * dy + dz * dz);
    }
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    const int nb = (P + BOX - 1) / BOX;
    for (int b = 0; b < nb; b++) {
        const float* bx = boxes + 6 * (size_t)b;
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
        if (x < bx[0] || x > bx[3]) ddx = fminf(fabsf(x - bx[0]), fabsf(x - bx[3]));
        if (y < bx[1] || y > bx[4]) ddy = fminf(fabsf(y - bx[1]), fabsf(y - bx[4]));
        if (z < bx[2] || z > bx[5]) ddz = fminf(fabsf(z - bx[2]), fabsf(z - bx[5]));
        const float dist = ddx * ddx + ddy * ddy + ddz * ddz;
        if (dist > reject || dist > best[2]) continue;
        const int e = (b + 1) * BOX < P ? (b + 1) * BOX : P;
        for (int i = b * BOX; i < e; i++) {
            if (i == idx) continue;
            const float dx = sorted[3 * (size_t)i] - x, dy = sorted[3 * (size_t)i + 1] - y, dz = sorted[3 * (size_t)i + 2] - z;
            update3(best, dx * dx + dy * dy + dz * dz);
        }
    }
    dists[order[idx]] = (best[0] + best[1] + best[2]) / 3.0f;
}

int knn_mean_dist2(int P, const float* points, float* mean_dists, gslic_alloc_fn alloc, void* ctx, hipStream_t s)
{
    if (P <= 0) return GSLIC_OK;
    const SortPlan plan = sort_plan((size_t)P, 30);
    const int nb = (P + BOX - 1) / BOX;
    size_t bytes = 0;
    {
        Carver c(nullptr);
        c.take<uint32_t>(P); c.take<uint32_t>(P); c.take<uint32_t>(P); c.take<uint32_t>(P);
        c.take<char>(sort_scratch_bytes(plan));
        c.take<float>(3 * (size_t)P); c.take<float>(6 * (size_t)nb); c.take<float>(8);
        bytes = c.used(nullptr) + 256;
    }
    char* base = alloc(ctx, bytes);
    if (!base) return set_error(GSLIC_ERR_ALLOC, "knn scratch allocator returned NULL for %zu bytes", bytes);
    Carver c(base);
    uint32_t* keys[2] = {c.take<uint32_t>(P), c.take<uint32_t>(P)};
    uint32_t* vals[2] = {c.take<uint32_t>(P), c.take<uint32_t>(P)};
    void* sort_scratch = c.take<char>(sort_scratch_bytes(plan));
    float* sorted = c.take<float>(3 * (size_t)P);
    float* boxes = c.take<float>(6 * (size_t)nb);
    float* mm = c.take<float>(8);
    GS_HIP(hipMemsetAsync(mm, 0, 8 * sizeof(float), s));
    int mblocks = div_up(P, 256);
    if (mblocks > 1024) mblocks = 1024;
    GS_LAUNCH(K_KNN_MINMAX, knn_minmax_kernel, dim3(mblocks), dim3(256), 0, s, P, points, mm);
    GS_LAUNCH(K_KNN_MORTON, knn_morton_kernel, dim3(div_up(P, 256)), dim3(256), 0, s, P, points, (const float*)mm, keys[0], vals[0]);
    SortBuffers sb;
    sb.keys[0] = keys[0]; sb.keys[1] = keys[1]; sb.v0[0] = vals[0]; sb.v0[1] = vals[1]; sb.v1[0] = sb.v1[1] = nullptr; sb.v2[0] = sb.v2[1] = nullptr; sb.v0_identity = false;
    GS_TRY(radix_sort_u32(sb, plan, sort_scratch, false, K_SORT_HIST, K_SORT_SCATTER, s));
    const uint32_t* order = vals[plan.passes & 1];
    GS_LAUNCH(K_KNN_BOXES, knn_boxes_kernel, dim3(nb), dim3(BOX), 0, s, P, points, order, sorted, boxes);
    GS_LAUNCH(K_KNN_SEARCH, knn_search_kernel, dim3(div_up(P, 256)), dim3(256), 0, s, P, (const float*)sorted, order,
              (const float*)boxes, mean_dists);
    return GSLIC_OK;
}

}  // namespace gslic
