// Micro-benchmark: what does grouping R instances by tile cost WITHOUT a sort — position = start[tile] + atomicAdd(cursor[tile], 1), three scattered
// 4-byte payload writes per instance — for a random tile pattern (the synthetic scene's rows as generated) and for a coherent one (rows in Morton
// order: neighbouring Gaussians touch neighbouring tiles)?  Per-instance atomics, and wave-aggregated ones (match-any on the tile id, one atomic per
// distinct tile of a 64-lane item).  Compare with the stable two-pass radix sort of the tile ids with three payloads: 0.146 ms at R = 6M.
// hipcc --offload-arch=gfx950 -O3 -o atomic_scatter atomic_scatter.hip && ./atomic_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_plain(const uint32_t* __restrict__ tile, const uint32_t* __restrict__ start, uint32_t* cursor, uint32_t* o0,
                                               uint32_t* o1, uint32_t* o2, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = tile[i];
    const uint32_t p = start[t] + atomicAdd(&cursor[t], 1u);
    o0[p] = i; o1[p] = i * 3u; o2[p] = i ^ 0x5555u;
}
__device__ __forceinline__ uint32_t popc_below(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__global__ __launch_bounds__(256) void k_agg(const uint32_t* __restrict__ tile, const uint32_t* __restrict__ start, uint32_t* cursor, uint32_t* o0,
                                             uint32_t* o1, uint32_t* o2, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool valid = i < n;
    const uint32_t t = valid ? tile[i] : 0u;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 13; b++) {
        const bool bit = (t >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    const uint32_t lower = popc_below(peers);
    uint32_t base = 0;
    if (valid && lower == 0) base = atomicAdd(&cursor[t], (uint32_t)__popcll(peers));
    base = (uint32_t)__shfl((int)base, __ffsll((unsigned long long)peers) - 1, 64);
    if (!valid) return;
    const uint32_t p = start[t] + base + lower;
    o0[p] = i; o1[p] = i * 3u; o2[p] = i ^ 0x5555u;
}

int main()
{
    const uint32_t R = 6000000, T = 8160, GX = 120;
    std::vector<uint32_t> h_rand(R), h_coh(R), cnt(T), start(T);
    srand(1);
    for (uint32_t i = 0; i < R; i++) h_rand[i] = (uint32_t)(((uint64_t)rand() * 32768ull + rand()) % T);
    // coherent: Gaussian g = i / 4 has a home tile that advances slowly with g (Morton-like: 2M Gaussians over 8160 tiles, with a depth spread that
    // makes neighbours in memory revisit a 4 x 4 tile neighbourhood), its four instances are the home tile and three neighbours
    for (uint32_t i = 0; i < R; i++) {
        const uint32_t g = i / 4, j = i % 4;
        const uint32_t home = (uint32_t)(((uint64_t)g * T) / (R / 4));
        int hx = (int)(home % GX) + (rand() % 4) - 2, hy = (int)(home / GX) + (rand() % 4) - 2;
        hx += (int)(j & 1); hy += (int)(j >> 1);
        hx = hx < 0 ? 0 : (hx >= (int)GX ? (int)GX - 1 : hx); hy = hy < 0 ? 0 : (hy >= 68 ? 67 : hy);
        h_coh[i] = (uint32_t)hy * GX + (uint32_t)hx;
    }
    uint32_t *d_tile, *d_start, *d_cur, *o0, *o1, *o2;
    CK(hipMalloc(&d_tile, R * 4)); CK(hipMalloc(&d_start, T * 4)); CK(hipMalloc(&d_cur, T * 4));
    CK(hipMalloc(&o0, R * 4)); CK(hipMalloc(&o1, R * 4)); CK(hipMalloc(&o2, R * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pat = 0; pat < 2; pat++) {
        const std::vector<uint32_t>& h = pat ? h_coh : h_rand;
        for (auto& c : cnt) c = 0;
        for (uint32_t i = 0; i < R; i++) cnt[h[i]]++;
        uint32_t s = 0; for (uint32_t t = 0; t < T; t++) { start[t] = s; s += cnt[t]; }
        CK(hipMemcpy(d_tile, h.data(), R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_start, start.data(), T * 4, hipMemcpyHostToDevice));
        for (int k = 0; k < 2; k++) {
            float ms_sum = 0;
            for (int it = 0; it < 12; it++) {
                CK(hipMemset(d_cur, 0, T * 4));
                CK(hipEventRecord(e0));
                if (k == 0) hipLaunchKernelGGL(k_plain, dim3((R + 255) / 256), dim3(256), 0, 0, d_tile, d_start, d_cur, o0, o1, o2, R);
                else hipLaunchKernelGGL(k_agg, dim3((R + 255) / 256), dim3(256), 0, 0, d_tile, d_start, d_cur, o0, o1, o2, R);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) ms_sum += ms;
            }
            printf("%s tiles, %s atomics: %.1f us per launch (R = %u, T = %u)\n", pat ? "coherent" : "random  ", k ? "wave-aggregated" : "per-instance   ", 1e3 * ms_sum / 10, R, T);
        }
    }
    return 0;
}
