// tile_bin.hip — the instances grouped by tile WITHOUT a sort (round 5).
//
// The reference sorts (tile << 32 | depth) keys (rasterizer_impl.cu:419-424).  Since the per-tile depth sort (radix_sort.hip) orders a tile's
// instances by (depth, tie key) — a total order — the order in which they ARRIVE in the tile's segment does not matter, and the stable two-pass
// radix sort on the tile id (0.146 ms at 2M / 1080p, three payloads) can be any grouping.  Per-instance atomics on the tiles' cursors lose to it
// (tools/ubench/atomic_scatter.hip: 240 us); BLOCK-aggregated ones do not, when the map's rows are stored in an order that keeps a block's
// instances on few tiles (Morton order of the centres: 4096 consecutive instances touch 330 tiles on the 2M / 1080p scene, 3200 when the rows
// are in random order): a block counts its tiles in an LDS histogram, the thread that drew rank 0 of a tile reserves the tile's count at the
// global cursor, and everybody stores one 16-byte row {depth, tie key, Gaussian id, emission slot} at cursor + rank — runs of ~12 rows per tile
// and block.  tools/ubench/tile_binning.hip on the scene's real tile stream: histogram 14 us + binning 46 us (rows in Morton order; 51 + 153
// in random order, where api.hip keeps the radix sort).
//
//   tile_hist_kernel : count[tile] += the block's count (ranges[tile].y, zero on entry)
//   tile_scan_kernel : ranges[tile] = {start, start} (the cursor), {0, 0} for an empty tile (as the reference's zeroed ranges)
//   tile_bin_kernel  : position = atomicAdd(ranges[tile].y, the block's count) + rank; on exit ranges[tile] = {start, end};
//                      every 16th block: status[GS_FLAG_BIN_ATOMICS] += its distinct tiles, [+ 1] += its instances
#include "kernels.h"
#include <atomic>
#include <cstdlib>

namespace gslic {

// Workgroup: THREADS x TB_ITEMS instances (launch_tile_bin picks THREADS: one histogram of up to 144 KB serves sixteen waves or four).
static constexpr int TB_ITEMS = 16;

__device__ __forceinline__ uint32_t tb_count(const TileBinArgs& a)
{
    return a.n_dev ? (*a.n_dev < a.n_cap ? *a.n_dev : a.n_cap) : a.n_cap;
}

template <int TB_THREADS>
__global__ __launch_bounds__(TB_THREADS) void tile_hist_kernel(const TileBinArgs a)
{
    extern __shared__ uint32_t h[];   // [T]
    if (a.status[2] != 0u) return;
    const uint32_t n = tb_count(a);
    const uint32_t b0 = blockIdx.x * (uint32_t)(TB_THREADS * TB_ITEMS);
    if (b0 >= n) return;
    for (int i = threadIdx.x; i < a.T; i += TB_THREADS) h[i] = 0u;
    __syncthreads();
    uint32_t t[TB_ITEMS];
#pragma unroll
    for (int j = 0; j < TB_ITEMS; j++) {
        const uint32_t i = b0 + (uint32_t)(j * TB_THREADS) + threadIdx.x;
        t[j] = i < n ? a.tile[i] : 0xffffffffu;
    }
#pragma unroll
    for (int j = 0; j < TB_ITEMS; j++)
        if (t[j] < (uint32_t)a.T) atomicAdd(&h[t[j]], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < a.T; i += TB_THREADS) {
        const uint32_t c = h[i];
        if (c) atomicAdd(&a.ranges[i].y, c);
    }
}

// One workgroup: the counts reach the threads through LDS (coalesced loads; a thread scans TS_PER consecutive tiles), the ranges leave the same way.
// The tiles' 64-entry bucket counts are scanned on the way (perTileBucketCount + InclusiveSum, rasterizer_impl.cu:433-441) and max_contrib is
// zeroed: what bucket_scan_kernel does on the radix path, without its launch.
static constexpr int TS_THREADS = 1024;
__global__ __launch_bounds__(TS_THREADS) void tile_scan_kernel(const TileBinArgs a)
{
    extern __shared__ uint32_t c[];   // [per * TS_THREADS (+ padding)]: count of tile t at c[t + t / 32] (a thread's run starts on its own bank)
    __shared__ uint32_t wsum[TS_THREADS / 64], wsum_b[TS_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.status[2] != 0u) {   // capacity mode: the instances did not fit — no lists, no buckets (B = 0 reaches the status words, as on the radix path)
        for (int t = tid; t < a.T; t += TS_THREADS) {
            a.ranges[t] = make_uint2(0u, 0u);
            if (a.bucket_offsets) { a.bucket_offsets[t] = 0u; a.max_contrib[t] = 0u; }
        }
        return;
    }
    const int per = (a.T + TS_THREADS - 1) / TS_THREADS;
    auto at = [&](int t) -> uint32_t& { return c[t + (t >> 5)]; };
    for (int t = tid; t < a.T; t += TS_THREADS) at(t) = a.ranges[t].y;
    __syncthreads();
    const int t0 = tid * per, t1 = (t0 + per) < a.T ? (t0 + per) : a.T;
    uint32_t sum = 0, sum_b = 0;
    for (int t = t0; t < t1; t++) { const uint32_t n = at(t); sum += n; sum_b += (n + (uint32_t)(GS_BUCKET - 1)) / (uint32_t)GS_BUCKET; }
    const uint32_t inc = wave_inclusive_scan(sum), inc_b = wave_inclusive_scan(sum_b);
    if (lane == 63) { wsum[wave] = inc; wsum_b[wave] = inc_b; }
    __syncthreads();
    uint32_t start = inc - sum, run_b = inc_b - sum_b;
    for (int w = 0; w < wave; w++) { start += wsum[w]; run_b += wsum_b[w]; }
    for (int t = t0; t < t1; t++) {
        const uint32_t n = at(t);
        at(t) = n ? start : 0xffffffffu;
        start += n;
        if (a.bucket_offsets) { run_b += (n + (uint32_t)(GS_BUCKET - 1)) / (uint32_t)GS_BUCKET; a.bucket_offsets[t] = run_b; a.max_contrib[t] = 0u; }
    }
    __syncthreads();
    for (int t = tid; t < a.T; t += TS_THREADS) {
        const uint32_t st = at(t);
        a.ranges[t] = st != 0xffffffffu ? make_uint2(st, st) : make_uint2(0u, 0u);
    }
}

// (134 VGPRs = three waves per SIMD; capped at 128 for four the kernel spills 24 and slows down: 0.056 -> 0.069 ms, profiles/r06n_occupancy_others_ab.log)
template <int TB_THREADS>
__global__ __launch_bounds__(TB_THREADS) void tile_bin_kernel(const TileBinArgs a)
{
    extern __shared__ uint32_t h[];   // [T]: the block's count per tile, then the tile's reserved position
    __shared__ uint32_t n_owned;
    if (a.status[2] != 0u) return;
    const uint32_t n = tb_count(a);
    const uint32_t b0 = blockIdx.x * (uint32_t)(TB_THREADS * TB_ITEMS);
    if (b0 >= n) return;
    for (int i = threadIdx.x; i < a.T; i += TB_THREADS) h[i] = 0u;
    if (threadIdx.x == 0) n_owned = 0u;
    __syncthreads();
    uint32_t t[TB_ITEMS], r[TB_ITEMS], g[TB_ITEMS], d[TB_ITEMS];
#pragma unroll
    for (int j = 0; j < TB_ITEMS; j++) {   // (all of a thread's loads in flight before the first use)
        const uint32_t i = b0 + (uint32_t)(j * TB_THREADS) + threadIdx.x;
        const bool in = i < n;
        t[j] = in ? a.tile[i] : 0xffffffffu;
        g[j] = in ? a.gid[i] : 0u;
        d[j] = in ? a.depth[i] : 0u;
        if (in && a.dead) a.dead[i] = 0;
    }
#pragma unroll
    for (int j = 0; j < TB_ITEMS; j++) {
        if (t[j] >= (uint32_t)a.T) t[j] = 0xffffffffu;
        r[j] = t[j] != 0xffffffffu ? atomicAdd(&h[t[j]], 1u) : 1u;   // rank among the block's instances of the tile (any order will do)
    }
    uint32_t tie[TB_ITEMS];
#pragma unroll
    for (int j = 0; j < TB_ITEMS; j++) tie[j] = (a.tie_rank && t[j] != 0xffffffffu) ? a.tie_rank[g[j]] : g[j];
    __syncthreads();
    uint32_t c[TB_ITEMS];
#pragma unroll
    for (int j = 0; j < TB_ITEMS; j++) c[j] = (r[j] == 0u && t[j] != 0xffffffffu) ? h[t[j]] : 0u;
    __syncthreads();
    uint32_t owned = 0;
#pragma unroll
    for (int j = 0; j < TB_ITEMS; j++)
        if (c[j]) { h[t[j]] = atomicAdd(&a.ranges[t[j]].y, c[j]); owned++; }   // (one global atomic per distinct tile of the block)
    // the path's cost on this map, for api.hip's choice: runs of rows it stores = global atomics it needed (one word per wave)
    // (ONE word per workgroup: 5.8k atomics on one address — one per wave — took as long as the rest of the kernel, 56 -> 93 us)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) owned += (uint32_t)__shfl_xor((int)owned, d, 64);
    if ((threadIdx.x & 63) == 0 && owned) atomicAdd(&n_owned, owned);
    __syncthreads();
    if (threadIdx.x == 0 && (blockIdx.x & 15u) == 0u) {   // (a sample of the workgroups: every one of them adding to the same word cost the kernel 8 us of 56)
        atomicAdd(a.status + GS_FLAG_BIN_ATOMICS, n_owned);
        const uint32_t left = n - b0;
        atomicAdd(a.status + GS_FLAG_BIN_ATOMICS + 1, left < (uint32_t)(TB_THREADS * TB_ITEMS) ? left : (uint32_t)(TB_THREADS * TB_ITEMS));
    }
#pragma unroll
    for (int j = 0; j < TB_ITEMS; j++) {
        if (t[j] == 0xffffffffu) continue;
        const uint32_t i = b0 + (uint32_t)(j * TB_THREADS) + threadIdx.x;
        a.binned[h[t[j]] + r[j]] = make_uint4(d[j], tie[j], g[j], i);
    }
}

// More than 64 KB of dynamic LDS per workgroup has to be asked for, once per kernel and device.  ONE query per device covers the four kernels
// (state per device: 0 = not asked, 1 = granted, 2 = refused; atomics — host threads may race to ask, the answer is the same).
static constexpr int TB_BIG_LDS_T = 14336;   // tiles above which the scan's padded counters (and, from 15 000, the histograms) pass 60 000 bytes
static std::atomic<int> g_big_lds[16];
bool tile_bin_lds_ok(int T)
{
    if (T <= TB_BIG_LDS_T) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return false; }
    int st = g_big_lds[dev].load(std::memory_order_acquire);
    if (st == 0) {
        const int want = 160 * 1024 - 256;
        bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(tile_hist_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bin_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bin_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(tile_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        (void)hipGetLastError();
        st = ok ? 1 : 2;
        g_big_lds[dev].store(st, std::memory_order_release);
    }
    return st == 1;
}
template <int TB_THREADS>
static int launch_tile_hist_t(const TileBinArgs& a, hipStream_t s)
{
    const unsigned block = (unsigned)(TB_THREADS * TB_ITEMS);
    GS_LAUNCH(K_TILE_HIST, tile_hist_kernel<TB_THREADS>, dim3((a.n_cap + block - 1u) / block), dim3(TB_THREADS), (size_t)a.T * sizeof(uint32_t), s, a);
    return GSLIC_OK;
}
template <int TB_THREADS>
static int launch_tile_bin_t(const TileBinArgs& a, hipStream_t s)
{
    const unsigned block = (unsigned)(TB_THREADS * TB_ITEMS);
    GS_LAUNCH(K_TILE_BIN, tile_bin_kernel<TB_THREADS>, dim3((a.n_cap + block - 1u) / block), dim3(TB_THREADS), (size_t)a.T * sizeof(uint32_t), s, a);
    return GSLIC_OK;
}

int launch_tile_bin(const TileBinArgs& a, hipStream_t s)
{
    if (a.n_cap == 0 || a.T <= 0) return GSLIC_OK;
    if (a.T > GS_TILE_BIN_MAX_T) return set_error(GSLIC_ERR_INVALID_ARG, "tile binning: more than %d tiles", GS_TILE_BIN_MAX_T);
    // Workgroup sizes as measured at 2M / 1080p (8160 tiles; histogram 14 -> 7 us with 1024 threads, binning 55 -> 61) and at the config-5 shape
    // (32 400 tiles, 130 KB of counters: one workgroup per CU either way — 0.36 -> 0.18 ms for the two with sixteen waves instead of four)
    // (api.hip's binning_choice only comes here when the device grants the LDS these launches ask for; a direct caller is told)
    if (!tile_bin_lds_ok(a.T)) return set_error(GSLIC_ERR_HIP, "tile binning: the dynamic LDS limit could not be raised");
    GS_TRY(launch_tile_hist_t<1024>(a, s));
    {
        const int per = (a.T + TS_THREADS - 1) / TS_THREADS;
        const size_t slds = ((size_t)per * TS_THREADS + (size_t)per * TS_THREADS / 32 + 1) * sizeof(uint32_t);
        GS_LAUNCH(K_TILE_SCAN, tile_scan_kernel, dim3(1), dim3(TS_THREADS), slds, s, a);
    }
    static const int forced = [] { const char* e = getenv("GSLIC_BIN_THREADS"); return e ? atoi(e) : 0; }();   // (A/B runs)
    if (forced == 1024 || (a.T > 8192 && !(forced == 256 && a.T <= 16384))) return launch_tile_bin_t<1024>(a, s);
    return launch_tile_bin_t<256>(a, s);
}

}  // namespace gslic
