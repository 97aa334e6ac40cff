// radix_sort.hip — stable LSD radix sort of u32 keys with one or two u32 payloads on a bit range
// (together with the depth-first emission order of api.hip this replaces cub::DeviceRadixSort::SortPairs on the
// 64-bit tile|depth keys, rasterizer_impl.cu:419-424 — see DESIGN.md "two-level binning").
//
// 8-bit digits.  A block owns GS_SORT_TILE = 4096 consecutive keys; inside a block the order is wave-major, then item, then
// lane.  Stability comes from ranking every key among the equal-digit keys that precede it: a 64-lane match-any (8 ballots)
// gives the rank inside one wave-wide item, per-wave LDS digit counters carry the rank across items, and a 256-thread pass
// folds the four waves.  Keys and payloads are then staged through LDS in block-sorted order so that the global writes are
// contiguous runs per digit.  Two ways of getting a block's global digit offsets:
//
//  classic  (default)  per pass: per-block digit histogram -> exclusive scan of the digit-major [256][nblk] matrix -> scatter.
//  onesweep (GSLIC_SORT_ONESWEEP bit) one histogram kernel for ALL digit positions, then ONE kernel per digit that ranks its
//           tile, publishes the tile's per-digit counts and resolves the tile's offsets by decoupled look-back over the
//           preceding tiles: 2 + passes launches instead of 5 * passes.
//
// Onesweep inter-workgroup protocol (MI355X: 8 XCDs with private, mutually non-coherent L2s — guide §6 G16): every shared word
// is an aligned 8-byte granule {flag:2 | value:62} written by ONE relaxed agent-scope atomic store and polled with relaxed
// agent-scope atomic loads (sc1: served past the non-coherent caches); the data IS the flag, so no fences are needed and
// nothing depends on block placement.  Tiles take their index from an atomic ticket, so every tile a block waits for has
// already started: the look-back cannot deadlock.  All words are zeroed by a memset ahead of the first pass.
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

// Per-block digit histogram.  Order inside the block is irrelevant here, so each thread takes 16 CONTIGUOUS keys as four
// 16-byte loads.  LDS atomics serialise on equal addresses, and high digits of nearly sorted keys are equal across a whole
// wave: that case is detected with one ballot and costs one atomic for the wave.
// n_dev != NULL (capacity mode): the element count lives on the device; `n` is then the capacity the launch was sized for.
__device__ __forceinline__ size_t sort_count(size_t n, const uint32_t* n_dev)
{
    if (!n_dev) return n;
    const size_t m = (size_t)*n_dev;
    return m < n ? m : n;
}
// Block 0 also clears the chained-scan state of its own pass (the scan is the next kernel in the stream): no memset launch per sort.
__global__ __launch_bounds__(RS_THREADS) void sort_hist_kernel(const uint32_t* __restrict__ keys, size_t n_cap, const uint32_t* __restrict__ n_dev,
                                                               int shift, uint32_t* __restrict__ hist, uint32_t nblk,
                                                               unsigned long long* __restrict__ scan_state, uint32_t scan_state_words)
{
    __shared__ uint32_t h[256];
    const size_t n = sort_count(n_cap, n_dev);
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < scan_state_words; i += RS_THREADS) scan_state[i] = 0ull;
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t t0 = (size_t)blockIdx.x * RS_TILE + (size_t)threadIdx.x * RS_ITEMS;
    uint32_t k[RS_ITEMS];
    if (t0 + RS_ITEMS <= n && ((reinterpret_cast<uintptr_t>(keys + t0) & 15) == 0)) {
        const uint4* p = reinterpret_cast<const uint4*>(keys + t0);
#pragma unroll
        for (int i = 0; i < RS_ITEMS / 4; i++) {
            const uint4 q = p[i];
            k[4 * i] = q.x; k[4 * i + 1] = q.y; k[4 * i + 2] = q.z; k[4 * i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) k[i] = (t0 + i < n) ? keys[t0 + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        if (t0 + i < n) {
            const uint32_t d = (k[i] >> shift) & 0xffu;
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            const uint64_t act = __ballot(true);
            if (__ballot(d != d0) == 0ull) {
                if (popc_below(act) == 0) atomicAdd(&h[d0], (uint32_t)__popcll(act));
            } else {
                atomicAdd(&h[d], 1u);
            }
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// digit totals of every pass in one sweep over the keys (onesweep)
__global__ __launch_bounds__(RS_THREADS) void sort_ghist_kernel(const uint32_t* __restrict__ keys, size_t n_cap, const uint32_t* __restrict__ n_dev,
                                                                int passes, uint32_t* __restrict__ ghist /*[passes][256], zeroed*/)
{
    const size_t n = sort_count(n_cap, n_dev);
    __shared__ uint32_t h[4 * 256];
    for (int i = threadIdx.x; i < passes * 256; i += RS_THREADS) h[i] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * RS_THREADS + threadIdx.x;
        if (idx < n) {
            const uint32_t k = keys[idx];
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

static constexpr unsigned long long OS_FLAG_LOCAL = 1ull << 62;   // value = this tile's count of the digit
static constexpr unsigned long long OS_FLAG_GLOBAL = 2ull << 62;  // value = inclusive count over tiles 0..this
static constexpr unsigned long long OS_VALUE_MASK = (1ull << 62) - 1;
static constexpr int OS_WINDOW = 8;

struct SortPassArgs {
    const uint32_t* kin;
    const uint32_t* vin[2];  // vin[0] == NULL: the first payload is the key's own index (first pass of an argsort)
    uint32_t* kout;
    uint32_t* vout[2];
    size_t n;
    const uint32_t* n_dev;       // capacity mode: the real count (<= n) on the device, NULL otherwise
    int shift;
    uint32_t nblk;
    const uint32_t* table;       // classic: scanned [256][nblk] histogram; onesweep: gbase[256] of this pass
    unsigned long long* status;  // onesweep: [nblk][256], zeroed
    uint32_t* ticket;            // onesweep: zeroed
    uint32_t* timeout;           // onesweep: device status word (bit 0 set when a look-back wait gave up)
};

template <int NV, bool ONESWEEP>
__global__ __launch_bounds__(RS_THREADS) void sort_scatter_kernel(const SortPassArgs a)
{
    __shared__ uint32_t cnt[RS_WAVES][256];
    __shared__ uint32_t dstart[256];
    __shared__ uint32_t gofs[256];
    __shared__ uint32_t red[8];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t skeys[RS_TILE];
    __shared__ uint32_t svals[RS_TILE];   // ONE payload at a time (see below)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (ONESWEEP && tid == 0) s_tile = atomicAdd(a.ticket, 1u);
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) cnt[w][tid] = 0;
    __syncthreads();
    const uint32_t tile = ONESWEEP ? s_tile : blockIdx.x;
    const size_t n = sort_count(a.n, a.n_dev);
    const int shift = a.shift;

    const size_t blk_base = (size_t)tile * RS_TILE;
    const size_t base = blk_base + (size_t)wave * RS_WAVE_TILE;
    uint32_t key[RS_ITEMS];
    uint32_t val[NV][RS_ITEMS];
    uint32_t rank[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * 64 + lane;
        const bool valid = idx < n;
        key[i] = valid ? a.kin[idx] : 0u;
        val[0][i] = valid ? (a.vin[0] ? a.vin[0][idx] : (uint32_t)idx) : 0u;
        if (NV > 1) val[NV - 1][i] = valid ? a.vin[NV - 1][idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const size_t idx = base + (size_t)i * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (key[i] >> shift) & 0xffu;
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
    uint32_t gl;
    if (ONESWEEP) {
        // publish this tile's count of digit t, then look back for the count over all preceding tiles
        unsigned long long* const my = a.status + (size_t)tile * 256 + tid;
        uint64_t excl = 0;
        if (tile == 0) {
            __hip_atomic_store(my, OS_FLAG_GLOBAL | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(my, OS_FLAG_LOCAL | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // windowed look-back: OS_WINDOW predecessors are polled with independent loads, then consumed nearest first — all
            // tiles of a pass start together, so a one-at-a-time walk would pay ~tile/2 dependent round trips
            bool done = false;
            for (int64_t t = (int64_t)tile - 1; t >= 0 && !done; t -= OS_WINDOW) {
                unsigned long long sv[OS_WINDOW];
#pragma unroll
                for (int j = 0; j < OS_WINDOW; j++)
                    sv[j] = (t - j >= 0) ? __hip_atomic_load(a.status + (size_t)(t - j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                         : OS_FLAG_GLOBAL;
#pragma unroll
                for (int j = 0; j < OS_WINDOW; j++) {
                    if (done) continue;
                    if ((sv[j] >> 62) == 0ull) {
                        const unsigned long long* w = a.status + (size_t)(t - j) * 256 + tid;
                        for (uint32_t spins = 0; (sv[j] >> 62) == 0ull && spins < (1u << 22); ++spins) {  // bounded: never hang the device
                            __builtin_amdgcn_s_sleep(1);
                            sv[j] = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if ((sv[j] >> 62) == 0ull && a.timeout) atomicOr(a.timeout, 1u);  // gave up: offsets of this pass are wrong
                    }
                    excl += sv[j] & OS_VALUE_MASK;
                    if ((sv[j] >> 62) == 2ull) done = true;
                }
            }
            __hip_atomic_store(my, OS_FLAG_GLOBAL | (unsigned long long)(excl + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gl = a.table[tid] + (uint32_t)excl;
    } else {
        gl = a.table[(size_t)tid * a.nblk + tile];
    }
    uint32_t block_total;
    const uint32_t ex = block256_exclusive_prefix(tot, block_total, red);
    dstart[tid] = ex;
    gofs[tid] = gl;
    __syncthreads();

    // Keys and payloads leave through LDS in block-sorted order, as contiguous runs per digit.  The payloads take turns in ONE staging
    // buffer: with two of them resident the kernel held 54 KB of LDS — two workgroups per CU — and it loses a third of its speed when
    // padding takes it from two to one (profiles/r03t_occupancy_sweep.log); 38 KB allow four.
    const size_t remain = n > blk_base ? n - blk_base : 0;
    const uint32_t nvalid = remain < (size_t)RS_TILE ? (uint32_t)remain : (uint32_t)RS_TILE;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        if (v > 0) __syncthreads();   // the previous payload has left the buffer
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) {
            const size_t idx = base + (size_t)i * 64 + lane;
            if (idx < n) {
                const uint32_t d = (key[i] >> shift) & 0xffu;
                const uint32_t pos = dstart[d] + cnt[wave][d] + rank[i];
                if (v == 0) skeys[pos] = key[i];
                svals[pos] = val[v][i];
            }
        }
        __syncthreads();
        for (uint32_t j = tid; j < nvalid; j += RS_THREADS) {
            const uint32_t k = skeys[j];
            const uint32_t d = (k >> shift) & 0xffu;
            const size_t g = (size_t)gofs[d] + (j - dstart[d]);
            if (v == 0) a.kout[g] = k;
            a.vout[v][g] = svals[j];
        }
    }
}

SortPlan sort_plan(size_t n, int end_bit)
{
    SortPlan p;
    p.n = n;
    p.passes = (end_bit + 7) / 8;
    if (p.passes > 4) p.passes = 4;
    if (p.passes < 1) p.passes = 1;
    p.nblk = div_up_sz(n, RS_TILE);
    if (p.nblk == 0) p.nblk = 1;
    p.hist_elems = 256 * p.nblk;
    return p;
}

// scratch layout: classic  u8 scan_state[4][scan_state_bytes(hist_elems)] (cleared by each pass's histogram kernel) | u32 hist[256*nblk]
//                 onesweep u64 status[4][nblk*256] | u32 ghist[4*256] | u32 gbase[4*256] | u32 tickets[64]
// (sized for 4 passes whatever the plan says, so that a buffer's layout depends on n alone)
static size_t classic_bytes(const SortPlan& plan) { return 4 * scan_state_bytes(plan.hist_elems) + plan.hist_elems * sizeof(uint32_t); }
static size_t onesweep_bytes(const SortPlan& plan)
{
    return (size_t)4 * plan.nblk * 256 * sizeof(unsigned long long) + (2 * 4 * 256 + 64) * sizeof(uint32_t);
}
size_t sort_scratch_bytes(const SortPlan& plan)
{
    const size_t a = classic_bytes(plan), b = onesweep_bytes(plan);
    return ((a > b ? a : b) + 255) & ~size_t(255);
}

template <int NV>
static int sort_impl(const SortBuffers& b, const SortPlan& plan, const uint32_t* n_dev, void* scratch, bool onesweep, int id_hist, int id_scatter,
                     hipStream_t s, uint32_t* fault)
{
    const unsigned nblk = (unsigned)plan.nblk;
    SortPassArgs a;
    a.n = plan.n; a.n_dev = n_dev; a.nblk = nblk; a.status = nullptr; a.ticket = nullptr; a.timeout = onesweep ? (fault ? fault : device_status_word()) : nullptr;
    const size_t ssb = scan_state_bytes(plan.hist_elems);
    uint32_t* hist = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + 4 * ssb);
    unsigned long long* status = reinterpret_cast<unsigned long long*>(scratch);
    uint32_t* ghist = reinterpret_cast<uint32_t*>(status + (size_t)4 * plan.nblk * 256);
    uint32_t* gbase = ghist + 4 * 256;
    uint32_t* tickets = gbase + 4 * 256;
    if (onesweep) {
        GS_HIP(hipMemsetAsync(scratch, 0, onesweep_bytes(plan), s));
        GS_LAUNCH(id_hist, sort_ghist_kernel, dim3(nblk), dim3(RS_THREADS), 0, s, (const uint32_t*)b.keys[0], plan.n, n_dev, plan.passes, ghist);
        GS_LAUNCH(K_SCAN_SPINE, sort_gbase_kernel, dim3(plan.passes), dim3(256), 0, s, (const uint32_t*)ghist, gbase);
    }
    for (int p = 0; p < plan.passes; p++) {
        const int src = p & 1, dst = src ^ 1;
        a.kin = b.keys[src]; a.kout = b.keys[dst]; a.shift = p * 8;
        a.vin[0] = (p == 0 && b.v0_identity) ? nullptr : b.v0[src];
        a.vout[0] = b.v0[dst];
        a.vin[1] = NV > 1 ? b.v1[src] : nullptr;
        a.vout[1] = NV > 1 ? b.v1[dst] : nullptr;
        if (onesweep) {
            a.table = gbase + p * 256; a.status = status + (size_t)p * plan.nblk * 256; a.ticket = tickets + p;
            GS_LAUNCH(id_scatter, (sort_scatter_kernel<NV, true>), dim3(nblk), dim3(RS_THREADS), 0, s, a);
        } else {
            unsigned long long* const state = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch) + (size_t)p * ssb);  // one chained-scan state per pass
            GS_LAUNCH(id_hist, sort_hist_kernel, dim3(nblk), dim3(RS_THREADS), 0, s, a.kin, plan.n, n_dev, a.shift, hist, nblk, state,
                      (uint32_t)(ssb / sizeof(unsigned long long)));
            GS_TRY(scan_u32_chained(hist, nullptr, hist, plan.hist_elems, true, state, s, fault));
            a.table = hist;
            GS_LAUNCH(id_scatter, (sort_scatter_kernel<NV, false>), dim3(nblk), dim3(RS_THREADS), 0, s, a);
        }
    }
    return GSLIC_OK;
}

// Runs of equal keys in a sorted (key, id) sequence re-ordered by rank[id] (launch_tie_fix).  One thread per element; the thread at the head
// of a run (its left neighbour differs, its right neighbour is equal) owns the run: runs are disjoint, so the in-place insertion sort needs no
// synchronisation.  Equal depth bits are rare (a few 1e4 pairs among 1e6 visible Gaussians) and runs short (2, seldom 3): the kernel is one
// coalesced read of the keys.
__global__ __launch_bounds__(256) void tie_fix_kernel(size_t n, const uint32_t* __restrict__ keys, uint32_t* ids, const uint32_t* __restrict__ rank,
                                                      uint32_t skip_key)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 >= n) return;
    const uint32_t k = keys[i];
    if (k == skip_key || keys[i + 1] != k || (i > 0 && keys[i - 1] == k)) return;
    size_t j = i + 1;
    while (j + 1 < n && keys[j + 1] == k) j++;          // run = [i, j]
    for (size_t a = i + 1; a <= j; a++) {                // insertion sort by rank
        const uint32_t id = ids[a], r = rank[id];
        size_t b = a;
        while (b > i && rank[ids[b - 1]] > r) { ids[b] = ids[b - 1]; b--; }
        ids[b] = id;
    }
}

int launch_tie_fix(size_t n, const uint32_t* sorted_keys, uint32_t* ids, const uint32_t* rank, uint32_t skip_key, hipStream_t s)
{
    if (n < 2) return GSLIC_OK;
    GS_LAUNCH(K_TIE_FIX, tie_fix_kernel, dim3((unsigned)div_up_sz(n, 256)), dim3(256), 0, s, n, sorted_keys, ids, rank, skip_key);
    return GSLIC_OK;
}

int radix_sort_u32(const SortBuffers& b, const SortPlan& plan, void* scratch, bool onesweep, int id_hist, int id_scatter, hipStream_t s,
                   const uint32_t* n_dev, uint32_t* fault)
{
    if (plan.n == 0) return GSLIC_OK;
    if (plan.n > 0xffffffffull) return set_error(GSLIC_ERR_INVALID_ARG, "radix sort: more than 2^32 elements");
    return b.v1[0] ? sort_impl<2>(b, plan, n_dev, scratch, onesweep, id_hist, id_scatter, s, fault)
                   : sort_impl<1>(b, plan, n_dev, scratch, onesweep, id_hist, id_scatter, s, fault);
}

}  // namespace gslic
