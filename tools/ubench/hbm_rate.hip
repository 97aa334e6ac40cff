// Micro-benchmark: practical HBM rates on MI355X for the access shapes of the hot path (float4 streams).
//   copy   : 1 read + 1 write stream
//   rmw3   : three read-modify-write streams (Adam's param / exp_avg / exp_avg_sq) + one read stream (gradient)
//   read   : 1 read stream (sum)
// hipcc --offload-arch=gfx950 -O3 -o hbm_rate hbm_rate.hip && ./hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_rmw3(float4* __restrict__ p, float4* __restrict__ m, float4* __restrict__ v, const float4* __restrict__ g, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 P = p[i], M = m[i], V = v[i]; const float4 G = g[i];
        M.x = 0.9f * M.x + 0.1f * G.x; M.y = 0.9f * M.y + 0.1f * G.y; M.z = 0.9f * M.z + 0.1f * G.z; M.w = 0.9f * M.w + 0.1f * G.w;
        V.x = 0.999f * V.x + 0.001f * G.x * G.x; V.y = 0.999f * V.y + 0.001f * G.y * G.y; V.z = 0.999f * V.z + 0.001f * G.z * G.z; V.w = 0.999f * V.w + 0.001f * G.w * G.w;
        P.x -= 1e-3f * M.x / (sqrtf(V.x) + 1e-15f); P.y -= 1e-3f * M.y / (sqrtf(V.y) + 1e-15f); P.z -= 1e-3f * M.z / (sqrtf(V.z) + 1e-15f); P.w -= 1e-3f * M.w / (sqrtf(V.w) + 1e-15f);
        p[i] = P; m[i] = M; v[i] = V;
    }
}
// the same two kernels with non-temporal stores (and loads): does bypassing the L2's retention help a pure stream on this part?
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_nt(const v4* __restrict__ a, v4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
__global__ __launch_bounds__(256) void k_rmw3_nt(v4* __restrict__ p, v4* __restrict__ m, v4* __restrict__ v, const v4* __restrict__ g, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v4 P = __builtin_nontemporal_load(p + i), M = __builtin_nontemporal_load(m + i), V = __builtin_nontemporal_load(v + i);
        const v4 G = __builtin_nontemporal_load(g + i);
        M = 0.9f * M + 0.1f * G;
        V = 0.999f * V + 0.001f * G * G;
        P.x -= 1e-3f * M.x / (sqrtf(V.x) + 1e-15f); P.y -= 1e-3f * M.y / (sqrtf(V.y) + 1e-15f); P.z -= 1e-3f * M.z / (sqrtf(V.z) + 1e-15f); P.w -= 1e-3f * M.w / (sqrtf(V.w) + 1e-15f);
        __builtin_nontemporal_store(P, p + i); __builtin_nontemporal_store(M, m + i); __builtin_nontemporal_store(V, v + i);
    }
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* out, size_t n)
{
    float s = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 x = a[i]; s += x.x + x.y + x.z + x.w; }
    if (s == 12345.678f) out[0] = s;
}
int main()
{
    const size_t n = (size_t)1 << 26;  // 64M float4 = 1 GiB per stream
    float4 *a, *b, *c, *d; float* o;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&d, n * 16)); CK(hipMalloc(&o, 16));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(c, 0, n * 16)); CK(hipMemset(d, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[3] = {256 * 8, 256 * 32, 256 * 128};
    for (int gi = 0; gi < 3; gi++) {
        const int grid = grids[gi];
        float ms;
        for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("grid %6d  copy  %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 16 / ms * 1e-9);
        for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_rmw3, dim3(grid), dim3(256), 0, 0, a, b, c, d, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("grid %6d  rmw3  %.3f ms  %.2f TB/s\n", grid, ms, 7.0 * n * 16 / ms * 1e-9);
        for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy_nt, dim3(grid), dim3(256), 0, 0, (const v4*)a, (v4*)b, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("grid %6d  copy, non-temporal  %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 16 / ms * 1e-9);
        for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_rmw3_nt, dim3(grid), dim3(256), 0, 0, (v4*)a, (v4*)b, (v4*)c, (const v4*)d, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("grid %6d  rmw3, non-temporal  %.3f ms  %.2f TB/s\n", grid, ms, 7.0 * n * 16 / ms * 1e-9);
        for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, o, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("grid %6d  read  %.3f ms  %.2f TB/s\n", grid, ms, 1.0 * n * 16 / ms * 1e-9);
    }
    return 0;
}
