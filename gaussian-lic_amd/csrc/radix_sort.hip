// radix_sort.hip — stable LSD radix sort of (u64 key, u32 payload) pairs on a bit range
// (replaces cub::DeviceRadixSort::SortPairs, rasterizer_impl.cu:419-424).
//
// 8-bit digits.  Per pass: (1) per-block digit histogram, (2) exclusive scan of the digit-major
// [256][nblk] counter matrix, (3) scatter.  A block owns GS_SORT_TILE = 4096 consecutive keys; inside a block
// the order is wave-major, then item, then lane.  Stability comes from ranking every key among the equal-digit
// keys that precede it: a 64-lane match-any (8 ballots) gives the rank inside one wave-wide item, per-wave LDS
// digit counters carry the rank across items, and a 256-thread pass folds the four waves.  Keys are then
// staged through LDS in block-sorted order so that the global writes are contiguous runs per digit.
#include "gslic_common.h"

namespace gslic {

static constexpr int RS_THREADS = GS_SORT_BLOCK;
static constexpr int RS_ITEMS = GS_SORT_ITEMS;
static constexpr int RS_TILE = GS_SORT_TILE;
static constexpr int RS_WAVES = RS_THREADS / 64;
static constexpr int RS_WAVE_TILE = RS_TILE / RS_WAVES;  // keys per wave

// Mask of the lanes (among `valid` ones) whose 8-bit digit equals mine.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}
__device__ __forceinline__ uint32_t popc_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__global__ __launch_bounds__(RS_THREADS) void sort_hist_kernel(const uint64_t* __restrict__ keys, size_t n, int shift,
                                                               uint32_t* __restrict__ hist, uint32_t nblk)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * RS_TILE + (size_t)wave * RS_WAVE_TILE;
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = valid ? (uint32_t)((keys[idx] >> shift) & 0xffu) : 0u;
        const uint64_t peers = match_digit(d, valid);
        if (valid && popc_below(peers) == 0) atomicAdd(&h[d], (uint32_t)__popcll(peers));
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) void sort_scatter_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                                  uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, size_t n,
                                                                  int shift, const uint32_t* __restrict__ hist_scanned, uint32_t nblk)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint32_t dstart[256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t red[8];
    __shared__ uint64_t skeys[RS_TILE];
    __shared__ uint32_t svals[RS_TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) cnt[w][tid] = 0;
    __syncthreads();

    const size_t blk_base = (size_t)blockIdx.x * RS_TILE;
    const size_t base = blk_base + (size_t)wave * RS_WAVE_TILE;
    uint64_t key[RS_ITEMS];
    uint32_t val[RS_ITEMS];
    uint32_t rank[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * 64 + lane;
        const bool valid = idx < n;
        key[i] = valid ? kin[idx] : 0ull;
        val[i] = valid ? vin[idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (uint32_t)((key[i] >> shift) & 0xffu);
        const uint64_t peers = match_digit(d, valid);
        const uint32_t lower = popc_below(peers);
        uint32_t old = 0;
        if (valid) old = cnt[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && lower == 0) cnt[wave][d] = old + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[i] = old + lower;
    }
    __syncthreads();

    // thread t owns digit t: fold the four waves, then exclusive scan over digits
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) {
        const uint32_t c = cnt[w][tid];
        cnt[w][tid] = tot;
        tot += c;
    }
    uint32_t block_total;
    const uint32_t excl = block256_exclusive_prefix(tot, block_total, red);
    dstart[tid] = excl;
    gbase[tid] = hist_scanned[(size_t)tid * nblk + blockIdx.x];
    __syncthreads();

#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * 64 + lane;
        if (idx < n) {
            const uint32_t d = (uint32_t)((key[i] >> shift) & 0xffu);
            const uint32_t pos = dstart[d] + cnt[wave][d] + rank[i];
            skeys[pos] = key[i];
            svals[pos] = val[i];
        }
    }
    __syncthreads();

    const size_t remain = n - blk_base;
    const uint32_t nvalid = remain < (size_t)RS_TILE ? (uint32_t)remain : (uint32_t)RS_TILE;
    for (uint32_t j = tid; j < nvalid; j += RS_THREADS) {
        const uint64_t k = skeys[j];
        const uint32_t d = (uint32_t)((k >> shift) & 0xffu);
        const size_t g = (size_t)gbase[d] + (j - dstart[d]);
        kout[g] = k;
        vout[g] = svals[j];
    }
}

SortPlan sort_plan(size_t n, int end_bit)
{
    SortPlan p;
    p.n = n;
    p.passes = (end_bit + 7) / 8;
    p.nblk = div_up_sz(n, RS_TILE);
    if (p.nblk == 0) p.nblk = 1;
    p.hist_elems = 256 * p.nblk;
    return p;
}

int radix_sort_pairs(uint64_t* keys[2], uint32_t* vals[2], const SortPlan& plan, uint32_t* hist, uint32_t* scan_temp,
                     hipStream_t s)
{
    if (plan.n == 0) return GSLIC_OK;
    const unsigned nblk = (unsigned)plan.nblk;
    for (int p = 0; p < plan.passes; p++) {
        const int src = p & 1, dst = src ^ 1;
        GS_LAUNCH(K_SORT_HIST, sort_hist_kernel, dim3(nblk), dim3(RS_THREADS), 0, s, (const uint64_t*)keys[src], plan.n, p * 8,
                  hist, nblk);
        GS_TRY(scan_u32(hist, hist, plan.hist_elems, true, scan_temp, s));
        GS_LAUNCH(K_SORT_SCATTER, sort_scatter_kernel, dim3(nblk), dim3(RS_THREADS), 0, s, (const uint64_t*)keys[src],
                  (const uint32_t*)vals[src], keys[dst], vals[dst], plan.n, p * 8, (const uint32_t*)hist, nblk);
    }
    return GSLIC_OK;
}

}  // namespace gslic
