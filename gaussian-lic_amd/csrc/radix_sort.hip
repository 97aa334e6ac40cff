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

// =====================================================================================================================
// Onesweep variant (opt-in, GSLIC_SORT_ONESWEEP=1): one histogram kernel for ALL digit positions, then ONE kernel per digit that ranks its tile,
// publishes the tile's per-digit counts and resolves the tile's global offsets by decoupled look-back over the preceding
// tiles — 1 + 1 + passes launches and (8 + 24*passes) B/key of traffic instead of 5*passes launches and 32*passes B/key.
//
// Inter-workgroup protocol (MI355X: 8 XCDs with private, mutually non-coherent L2s — guide §6 G16): every shared word is an
// aligned 8-byte granule {flag:2 | value:62} written by ONE relaxed agent-scope atomic store and polled with relaxed
// agent-scope atomic loads (sc1: served past the non-coherent caches); the data IS the flag, so no fences are needed and
// nothing depends on block placement.  Tiles take their index from an atomic ticket, so every tile a block waits for has
// already started: the look-back cannot deadlock.  All words are zeroed by a memset node ahead of the first pass.
// =====================================================================================================================
static constexpr unsigned long long OS_FLAG_LOCAL = 1ull << 62;   // value = this tile's count of the digit
static constexpr unsigned long long OS_FLAG_GLOBAL = 2ull << 62;  // value = inclusive count over tiles 0..this
static constexpr unsigned long long OS_VALUE_MASK = (1ull << 62) - 1;

__global__ __launch_bounds__(RS_THREADS) void sort_ghist_kernel(const uint64_t* __restrict__ keys, size_t n, int passes,
                                                                uint32_t* __restrict__ ghist /*[passes][256], zeroed*/)
{
    __shared__ uint32_t h[8 * 256];
    for (int i = threadIdx.x; i < passes * 256; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * RS_THREADS + threadIdx.x;
        if (idx < n) {
            const uint64_t k = keys[idx];
            for (int p = 0; p < passes; p++) atomicAdd(&h[p * 256 + (int)((k >> (8 * p)) & 0xffu)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * 256; i += RS_THREADS)
        if (h[i]) atomicAdd(&ghist[i], h[i]);
}

// one block per pass: exclusive scan of the 256 digit totals
__global__ __launch_bounds__(256) void sort_gbase_kernel(const uint32_t* __restrict__ ghist, uint32_t* __restrict__ gbase)
{
    __shared__ uint32_t red[8];
    const uint32_t v = ghist[blockIdx.x * 256 + threadIdx.x];
    uint32_t total;
    gbase[blockIdx.x * 256 + threadIdx.x] = block256_exclusive_prefix(v, total, red);
}

__global__ __launch_bounds__(RS_THREADS) void sort_onesweep_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                                   uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, size_t n,
                                                                   int shift, const uint32_t* __restrict__ gbase /*[256]*/,
                                                                   unsigned long long* status /*[nblk][256], zeroed*/,
                                                                   uint32_t* ticket /*zeroed*/)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint32_t dstart[256];
    __shared__ uint32_t gofs[256];
    __shared__ uint32_t red[8];
    __shared__ uint32_t s_tile;
    __shared__ uint64_t skeys[RS_TILE];
    __shared__ uint32_t svals[RS_TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) cnt[w][tid] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;

    const size_t blk_base = (size_t)tile * RS_TILE;
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

    // thread t owns digit t: fold the four waves
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) {
        const uint32_t c = cnt[w][tid];
        cnt[w][tid] = tot;
        tot += c;
    }
    // publish this tile's count of digit t, then look back for the count over all preceding tiles
    unsigned long long* const my = status + (size_t)tile * 256 + tid;
    uint64_t excl = 0;
    if (tile == 0) {
        __hip_atomic_store(my, OS_FLAG_GLOBAL | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(my, OS_FLAG_LOCAL | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int64_t t = (int64_t)tile - 1; t >= 0; t--) {
            const unsigned long long* w = status + (size_t)t * 256 + tid;
            unsigned long long sv = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint32_t spins = 0; (sv >> 62) == 0ull && spins < (1u << 22); ++spins) {  // bounded: never hang the device
                __builtin_amdgcn_s_sleep(1);
                sv = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            excl += sv & OS_VALUE_MASK;
            if ((sv >> 62) == 2ull) break;
        }
        __hip_atomic_store(my, OS_FLAG_GLOBAL | (unsigned long long)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t block_total;
    const uint32_t ex = block256_exclusive_prefix(tot, block_total, red);
    dstart[tid] = ex;
    gofs[tid] = gbase[tid] + (uint32_t)excl;
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
        const size_t g = (size_t)gofs[d] + (j - dstart[d]);
        kout[g] = k;
        vout[g] = svals[j];
    }
}

size_t onesweep_state_bytes(const SortPlan& plan)
{
    // [passes] x { status u64[nblk*256] } + ghist u32[8*256] + gbase u32[8*256] + tickets u32[64]
    return (size_t)plan.passes * plan.nblk * 256 * sizeof(unsigned long long) + 2 * 8 * 256 * sizeof(uint32_t) + 64 * sizeof(uint32_t);
}

int radix_sort_pairs_onesweep(uint64_t* keys[2], uint32_t* vals[2], const SortPlan& plan, void* state, hipStream_t s)
{
    if (plan.n == 0) return GSLIC_OK;
    if (plan.passes > 8) return set_error(GSLIC_ERR_INVALID_ARG, "radix sort: more than 8 digit passes");
    const unsigned nblk = (unsigned)plan.nblk;
    unsigned long long* status = reinterpret_cast<unsigned long long*>(state);
    uint32_t* ghist = reinterpret_cast<uint32_t*>(status + (size_t)plan.passes * plan.nblk * 256);
    uint32_t* gbase = ghist + 8 * 256;
    uint32_t* tickets = gbase + 8 * 256;
    GS_HIP(hipMemsetAsync(state, 0, onesweep_state_bytes(plan), s));
    GS_LAUNCH(K_SORT_HIST, sort_ghist_kernel, dim3(nblk), dim3(RS_THREADS), 0, s, (const uint64_t*)keys[0], plan.n, plan.passes, ghist);
    GS_LAUNCH(K_SCAN_SPINE, sort_gbase_kernel, dim3(plan.passes), dim3(256), 0, s, (const uint32_t*)ghist, gbase);
    for (int p = 0; p < plan.passes; p++) {
        const int src = p & 1, dst = src ^ 1;
        GS_LAUNCH(K_SORT_SCATTER, sort_onesweep_kernel, dim3(nblk), dim3(RS_THREADS), 0, s, (const uint64_t*)keys[src],
                  (const uint32_t*)vals[src], keys[dst], vals[dst], plan.n, p * 8, (const uint32_t*)(gbase + p * 256),
                  status + (size_t)p * plan.nblk * 256, tickets + p);
    }
    return GSLIC_OK;
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
