// Micro-benchmark on the REAL emission-order tile stream of the 2M / 1080p scene (tools/ubench/tiles_{morton,insertion}.bin, written by
// tools/ubench/gen_tile_streams.py from the CPU oracle): grouping the instances by tile with BLOCK-aggregated atomics instead of the stable two-pass
// radix sort (0.146 ms with three payloads).  A block of 4096 instances counts its tiles in an LDS histogram (8192 bins); the lane that drew rank 0
// of a bin reserves the bin's count at the tile's global cursor; everybody stores at cursor + rank.
//   k_hist    : tile histogram (LDS, then one global atomic per distinct tile of the block)
//   k_bin<V>  : V = 0 three 4-byte stores, 1 one 16-byte store, 2 one 4-byte + one 8-byte store
// hipcc --offload-arch=gfx950 -O3 -o tile_binning tile_binning.hip && ./tile_binning tiles_morton.bin tiles_insertion.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static constexpr int TB = 8192, IPT = 16, BLK = 256 * IPT;

__global__ __launch_bounds__(256) void k_hist(const uint32_t* __restrict__ tile, uint32_t* count, uint32_t n)
{
    __shared__ uint32_t h[TB];
    for (int i = threadIdx.x; i < TB; i += 256) h[i] = 0u;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * BLK;
    uint32_t t[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) { const uint32_t i = b0 + j * 256 + threadIdx.x; t[j] = i < n ? tile[i] : 0xffffffffu; }
#pragma unroll
    for (int j = 0; j < IPT; j++) if (t[j] != 0xffffffffu) atomicAdd(&h[t[j]], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < TB; i += 256) { const uint32_t c = h[i]; if (c) atomicAdd(&count[i], c); }
}

template <int V>
__global__ __launch_bounds__(256) void k_bin(const uint32_t* __restrict__ tile, const uint32_t* __restrict__ gid, const uint32_t* __restrict__ depth,
                                             uint32_t* cursor, uint32_t* o0, uint32_t* o1, uint32_t* o2, uint32_t n)
{
    __shared__ uint32_t h[TB];
    for (int i = threadIdx.x; i < TB; i += 256) h[i] = 0u;
    __syncthreads();
    const uint32_t b0 = blockIdx.x * BLK;
    uint32_t t[IPT], r[IPT], g[IPT], d[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        const uint32_t i = b0 + j * 256 + threadIdx.x;
        t[j] = i < n ? tile[i] : 0xffffffffu; g[j] = i < n ? gid[i] : 0u; d[j] = i < n ? depth[i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < IPT; j++) r[j] = t[j] != 0xffffffffu ? atomicAdd(&h[t[j]], 1u) : 1u;
    __syncthreads();
    uint32_t c[IPT];
#pragma unroll
    for (int j = 0; j < IPT; j++) c[j] = (r[j] == 0u && t[j] != 0xffffffffu) ? h[t[j]] : 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; j++) if (c[j]) h[t[j]] = atomicAdd(&cursor[t[j]], c[j]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; j++) {
        if (t[j] == 0xffffffffu) continue;
        const uint32_t i = b0 + j * 256 + threadIdx.x;
        const uint32_t p = h[t[j]] + r[j];
        if (V == 0) { o0[p] = d[j]; o1[p] = g[j]; o2[p] = i; }
        else if (V == 1) reinterpret_cast<uint4*>(o0)[p] = make_uint4(d[j], g[j], i, 0u);
        else { o0[p] = d[j]; reinterpret_cast<uint2*>(o1)[p] = make_uint2(g[j], i); }
    }
}

int main(int argc, char** argv)
{
    const uint32_t T = 8160;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int f = 1; f < argc; f++) {
        FILE* fp = fopen(argv[f], "rb"); if (!fp) { printf("cannot open %s\n", argv[f]); continue; }
        fseek(fp, 0, SEEK_END); const uint32_t R = (uint32_t)(ftell(fp) / 4); fseek(fp, 0, SEEK_SET);
        std::vector<uint32_t> h(R); if (fread(h.data(), 4, R, fp) != R) return 1; fclose(fp);
        std::vector<uint32_t> cnt(T, 0), start(T), hg(R), hd(R);
        for (uint32_t i = 0; i < R; i++) { cnt[h[i]]++; hg[i] = i / 4; hd[i] = 0x40000000u + (i * 2654435761u >> 8); }
        uint32_t s = 0; for (uint32_t t = 0; t < T; t++) { start[t] = s; s += cnt[t]; }
        uint32_t *d_tile, *d_gid, *d_depth, *d_cnt, *d_cur, *o0, *o1, *o2;
        CK(hipMalloc(&d_tile, R * 4)); CK(hipMalloc(&d_gid, R * 4)); CK(hipMalloc(&d_depth, R * 4)); CK(hipMalloc(&d_cnt, TB * 4)); CK(hipMalloc(&d_cur, TB * 4));
        CK(hipMalloc(&o0, (size_t)R * 16)); CK(hipMalloc(&o1, (size_t)R * 8)); CK(hipMalloc(&o2, R * 4));
        CK(hipMemcpy(d_tile, h.data(), R * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_gid, hg.data(), R * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_depth, hd.data(), R * 4, hipMemcpyHostToDevice));
        const unsigned grid = (R + BLK - 1) / BLK;
        for (int k = 0; k < 4; k++) {
            float ms_sum = 0;
            for (int it = 0; it < 12; it++) {
                if (k == 0) CK(hipMemset(d_cnt, 0, TB * 4)); else CK(hipMemcpy(d_cur, start.data(), T * 4, hipMemcpyHostToDevice));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (k == 0) hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, 0, d_tile, d_cnt, R);
                else if (k == 1) hipLaunchKernelGGL(k_bin<0>, dim3(grid), dim3(256), 0, 0, d_tile, d_gid, d_depth, d_cur, o0, o1, o2, R);
                else if (k == 2) hipLaunchKernelGGL(k_bin<1>, dim3(grid), dim3(256), 0, 0, d_tile, d_gid, d_depth, d_cur, o0, o1, o2, R);
                else hipLaunchKernelGGL(k_bin<2>, dim3(grid), dim3(256), 0, 0, d_tile, d_gid, d_depth, d_cur, o0, o1, o2, R);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) ms_sum += ms;
            }
            if (k == 0) {   // check the histogram
                std::vector<uint32_t> c2(T); CK(hipMemcpy(c2.data(), d_cnt, T * 4, hipMemcpyDeviceToHost));
                for (uint32_t t = 0; t < T; t++) if (c2[t] != cnt[t]) { printf("histogram mismatch at %u\n", t); break; }
            }
            if (k == 1) {   // check the grouping: every output position's instance has the position's tile
                std::vector<uint32_t> oi(R); CK(hipMemcpy(oi.data(), o2, R * 4, hipMemcpyDeviceToHost));
                uint32_t bad = 0; for (uint32_t t = 0; t < T; t++) for (uint32_t p = start[t]; p < start[t] + cnt[t]; p++) bad += h[oi[p]] != t;
                printf("  grouping check: %u misplaced\n", bad);
            }
            static const char* names[4] = {"k_hist", "k_bin, 3 x 4-byte stores", "k_bin, one 16-byte store", "k_bin, 4 + 8-byte stores"};
            printf("%s: %s %.1f us per launch (R = %u, %u blocks)\n", argv[f], names[k], 1e3 * ms_sum / 10, R, grid);
        }
        hipFree(d_tile); hipFree(d_gid); hipFree(d_depth); hipFree(d_cnt); hipFree(d_cur); hipFree(o0); hipFree(o1); hipFree(o2);
    }
    return 0;
}
