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
    const uint32_t* vin[3];  // vin[0] == NULL: the first payload is the key's own index (first pass of an argsort)
    uint32_t* kout;
    uint32_t* vout[3];
    size_t n;
    const uint32_t* n_dev;       // capacity mode: the real count (<= n) on the device, NULL otherwise
    int shift;
    uint32_t nblk;
    const uint32_t* table;       // classic: scanned [256][nblk] histogram; onesweep: gbase[256] of this pass
    unsigned long long* status;  // onesweep: [nblk][256], zeroed
    uint32_t* ticket;            // onesweep: zeroed
    uint32_t* timeout;           // onesweep: device status word (bit 0 set when a look-back wait gave up)
};

// (148 VGPRs = three waves per SIMD for the three-payload variant; capped at 128 it spills 40 and runs as fast: 0.0595 / 0.0597 ms, profiles/r06o_scatter_occupancy_ab.log)
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
#pragma unroll
        for (int v = 1; v < NV; v++) val[v][i] = valid ? a.vin[v][idx] : 0u;
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
        a.vin[2] = NV > 2 ? b.v2[src] : nullptr;
        a.vout[2] = NV > 2 ? b.v2[dst] : nullptr;
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

// ---------------------------------------------------------------------------------------------------------
// Per-tile depth sort (TileDepthSortArgs, gslic_common.h).  One workgroup of 256 threads per tile.  The sort is an LSD radix sort on the
// four bytes of the depth bits of (depth, local index) pairs; a pass whose digit is the same for the whole segment is skipped.  A pass
// ranks the segment in order, LS_CHUNK = 1024 elements at a time — wave w of the chunk owns 256 contiguous elements as four 64-lane
// items; match-any gives an element's rank among the equal digits of its item, per-wave LDS counters carry it across the wave's items,
// thread d folds the four waves of digit d and bumps the segment's running offset of digit d — so equal digits keep their order: stable.
// This is the path of the LONG lists (more than LW_CAP instances in a tile: tile_depth_sort_wave_kernel below does the others): keys and
// 16-bit indices ping-pong in LDS (LDS = true: up to LS_CAP instances) or in the four global scratch arrays (LDS = false: any length).
static constexpr int LS_THREADS = 256;
static constexpr int LS_CHUNK = 1024;

// What the two kinds of input look like to the sort (TileDepthSortArgs): BINNED = 16-byte rows {depth, tie key, Gaussian id, slot} grouped by
// tile in any order; otherwise three arrays sorted by tile, in index order inside a tile.
template <bool BINNED>
struct TileIn {
    const TileDepthSortArgs& a;
    const uint32_t x;
    __device__ __forceinline__ uint32_t depth(uint32_t e) const { return BINNED ? a.binned[x + e].x : __builtin_nontemporal_load(a.depth + x + e); }
    // the second key: ascending original index among equal depths (forward.cu / rasterizer_impl.cu:419: the reference's stable sort of an
    // index-ordered emission).  The stable tile sort's output is in row order already: without tie_rank its local index IS that key.
    __device__ __forceinline__ uint32_t tie(uint32_t e) const
    {
        return BINNED ? a.binned[x + e].y : (a.tie_rank ? a.tie_rank[a.gauss_in[x + e]] : e);
    }
    __device__ __forceinline__ uint2 out(uint32_t e) const
    {
        if (BINNED) return *reinterpret_cast<const uint2*>(&a.binned[x + e].z);   // (the upper half of the row: one 8-byte load)
        return make_uint2(a.gauss_in[x + e], a.slot_in[x + e]);
    }
    // equal depths need the second key unless the input is in row order and the row order is the original one
    __device__ __forceinline__ bool two_keys() const { return BINNED || a.tie_rank != nullptr; }
};

template <bool LDS, bool BINNED>
__device__ __forceinline__ void tile_depth_sort_body(const TileDepthSortArgs& a, const uint32_t x, const uint32_t n, uint32_t* const kL0, uint32_t* const kL1,
                                                     uint16_t* const iL0, uint16_t* const iL1, uint32_t* const hist, uint32_t* const run,
                                                     uint32_t (*cnt)[256], uint32_t* const red)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TileIn<BINNED> in{a, x};
    // ping-pong sides: side 0 = {kL0, iL0} or {key_a, idx_a}; side 1 = {kL1, iL1} or {key_b, idx_b} (four scratch arrays that alias no input)
    uint32_t* const kG0 = a.key_a + x;
    uint32_t* const kG1 = a.key_b + x;
    uint32_t* const iG0 = a.idx_a + x;
    uint32_t* const iG1 = a.idx_b + x;
    auto rdk = [&](int side, uint32_t e) -> uint32_t { return LDS ? (side ? kL1[e] : kL0[e]) : (side ? kG1[e] : kG0[e]); };
    auto rdi = [&](int side, uint32_t e) -> uint32_t { return LDS ? (uint32_t)(side ? iL1[e] : iL0[e]) : (side ? iG1[e] : iG0[e]); };
    auto wr = [&](int side, uint32_t e, uint32_t k, uint32_t i) {
        if (LDS) { if (side) { kL1[e] = k; iL1[e] = (uint16_t)i; } else { kL0[e] = k; iL0[e] = (uint16_t)i; } }
        else { if (side) { kG1[e] = k; iG1[e] = i; } else { kG0[e] = k; iG0[e] = i; } }
    };
    // A long list always takes both keys when it needs them at all (LSD: the tie key first, then the depth — both passes stable): lists this long
    // are rare, and a run of equal depths can be as long as the list (a wall seen head-on), which rules out fixing runs up afterwards.
    int cur = 0;
#pragma unroll 1
    for (int phase = in.two_keys() ? 0 : 1; phase < 2; phase++) {
        if (phase == 0) {
            for (uint32_t e = tid; e < n; e += LS_THREADS) wr(0, e, in.tie(e), e);
        } else if (!in.two_keys()) {
            for (uint32_t e = tid; e < n; e += LS_THREADS) wr(0, e, in.depth(e), e);
        } else {
            for (uint32_t p = tid; p < n; p += LS_THREADS) { const uint32_t i = rdi(cur, p); wr(cur, p, in.depth(i), i); }
        }
        __syncthreads();
#pragma unroll 1
        for (int shift = 0; shift < 32; shift += 8) {
            // ---- digit histogram of the segment (one LDS atomic per distinct digit of a 64-lane item)
            hist[tid] = 0;
            __syncthreads();
            for (uint32_t e0 = 0; e0 < n; e0 += LS_THREADS) {
                const uint32_t e = e0 + tid;
                const bool valid = e < n;
                const uint32_t d = valid ? ((rdk(cur, e) >> shift) & 0xffu) : 0u;
                const uint64_t peers = match_digit(d, valid);
                if (valid && popc_below(peers) == 0) atomicAdd(&hist[d], (uint32_t)__popcll(peers));
            }
            __syncthreads();
            const uint32_t d_first = (rdk(cur, 0) >> shift) & 0xffu;
            const bool uniform = hist[d_first] == n;   // (the same for every thread: LDS values behind a barrier)
            uint32_t total;
            const uint32_t ex = block256_exclusive_prefix(hist[tid], total, red);   // (two barriers inside)
            if (uniform) continue;                     // every key has this digit: the pass would be the identity
            run[tid] = ex;
            // ---- stable scatter, LS_CHUNK elements at a time
            for (uint32_t c0 = 0; c0 < n; c0 += LS_CHUNK) {
#pragma unroll
                for (int w = 0; w < 4; w++) cnt[w][tid] = 0;
                __syncthreads();
                uint32_t rk[4], dg[4], kk[4], ii[4];
                bool vl[4];
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const uint32_t e = c0 + (uint32_t)wave * 256u + (uint32_t)it * 64u + (uint32_t)lane;
                    vl[it] = e < n;
                    kk[it] = vl[it] ? rdk(cur, e) : 0u;
                    ii[it] = vl[it] ? rdi(cur, e) : 0u;
                    dg[it] = (kk[it] >> shift) & 0xffu;
                    const uint64_t peers = match_digit(dg[it], vl[it]);
                    const uint32_t lower = popc_below(peers);
                    uint32_t old = 0;
                    if (vl[it]) old = cnt[wave][dg[it]];
                    __builtin_amdgcn_wave_barrier();
                    if (vl[it] && lower == 0) cnt[wave][dg[it]] = old + (uint32_t)__popcll(peers);
                    __builtin_amdgcn_wave_barrier();
                    rk[it] = old + lower;
                }
                __syncthreads();
                {   // thread d: where each wave's run of digit d starts in the output, and the segment's running offset behind this chunk
                    uint32_t o = run[tid];
#pragma unroll
                    for (int w = 0; w < 4; w++) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = o; o += c; }
                    run[tid] = o;
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; it++)
                    if (vl[it]) wr(cur ^ 1, cnt[wave][dg[it]] + rk[it], kk[it], ii[it]);
                __syncthreads();
            }
            cur ^= 1;
        }
        __syncthreads();
    }
    for (uint32_t j = tid; j < n; j += LS_THREADS) {
        const uint2 o = in.out(rdi(cur, j));
        a.gauss_out[x + j] = o.x;
        a.slot_out[x + j] = o.y;
    }
}

// ONE WAVE per tile for segments of up to LW_CAP instances (the common case): no workgroup barrier anywhere, and no match-any either.
// The pairs live in ONE LDS buffer (two, ping-pong, until round 6: GS_LSORT_SINGLE below) in a BLOCKED layout: lane l owns the E = ceil(n / 64) consecutive elements [l E, (l + 1) E) —
// order = (lane, position in the lane's run).  A pass takes FOUR bits: every lane counts the sixteen digits of its run into its own column of a
// 16 x 64 counter matrix (no conflicts, no atomics between lanes), the matrix is scanned in (digit, lane) order — which IS the stable order —
// and every lane walks its run once more, taking each element's position from its column's counter.  Eight passes of ~25 instructions per
// element and lane against four of ~70 per element and WAVE-WIDE ITEM for match-any ranking (8 ballots per item): 3x fewer instructions.
// Element p of the sorted order lives at word phys(p) = (p / E) * (E | 1) + p % E: an odd stride per lane keeps the lanes' runs on distinct banks.
//
// Equal depths.  The reference's list has them in ascending original index (a stable sort of an index-ordered emission).  Input in row order
// of a map whose row order is the original one gets that from the passes' stability; any other input (rows grouped by tile in arrival order;
// a permuted map) gets it from a SECOND key, and only when the tile has two equal depths at all (3 % of the tiles of the 2M / 1080p scene):
// the list is then sorted by the tie key and by the depth again — LSD over (depth, tie key), whatever the length of the runs.
static constexpr int LW_CAP = 1024;
static constexpr int LW_PHYS = 64 * 17;
#define GS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// LDS accesses of ONE wave execute in program order, an instruction at a time for all its lanes: lane A's write is seen by lane B's later read
// without a wait of its own.  GS_WAVE_ORDER() keeps the COMPILER from moving accesses across it — a wavefront-scope fence plus the barrier, like
// the key build and the scans (the bare barrier intrinsic is not a memory fence for IR-level passes: ADVICE round 5); no instruction either way.
#define GS_WAVE_ORDER() GS_WAVE_SYNC()

// GS_LSORT_SINGLE: ONE pair buffer instead of two.  A pass holds every lane's whole run in registers before the first element is written back (the
// reads, the counting, the scan and the ranking all come first, and a wave's LDS instructions execute in program order), so the scatter may
// overwrite the buffer it was read from: 10.9 KB of LDS per wave instead of 17.4 — fourteen waves per CU instead of nine for a kernel that is
// bound by the latency of a pass's dependent chain — twelve in fact, at the 168 VGPRs of three waves per SIMD: 0.1011 -> 0.097 ms (four waves: 47 spilled,
// 0.115; profiles/r06n_lsort_single_ab.log).
#ifndef GS_LSORT_SINGLE
#define GS_LSORT_SINGLE 1
#endif
#ifndef GS_LSORT_WPE
#define GS_LSORT_WPE 3
#endif
static constexpr int LW_NB = GS_LSORT_SINGLE ? 1 : 2;
template <bool BINNED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GS_LSORT_WPE, GS_LSORT_WPE))) void tile_depth_sort_wave_kernel(const TileDepthSortArgs a)
{
    __shared__ uint32_t sk[LW_NB][LW_PHYS];
    __shared__ uint16_t si[LW_NB][LW_PHYS];
    __shared__ uint32_t C[16 * 64 + 64];   // counter (digit d, lane l) = word 64 d + l of the matrix, stored at f + (f >> 4): see the scan
    if (a.status[2] != 0u) return;   // capacity mode: the instance lists did not fit
    const uint2 range = a.ranges[blockIdx.x];
    const uint32_t x = range.x, n = range.y - range.x;
    const uint32_t lane = threadIdx.x;
    if (n > (uint32_t)LW_CAP) {   // a long list: queued for tile_depth_sort_kernel, launched behind this one
        if (lane == 0) a.long_tiles[atomicAdd(a.status + GS_FLAG_LONG, 1u)] = blockIdx.x;
        return;
    }
    if (n == 0) return;
    const TileIn<BINNED> in{a, x};
    if (n <= 2) {
        if (lane == 0) {
            bool swap = false;
            if (n == 2) {
                const uint32_t k0 = in.depth(0), k1 = in.depth(1);
                swap = k1 < k0 || (k1 == k0 && in.two_keys() && in.tie(1) < in.tie(0));
            }
            const uint2 o0 = in.out(swap ? 1u : 0u);
            a.gauss_out[x] = o0.x; a.slot_out[x] = o0.y;
            if (n == 2) { const uint2 o1 = in.out(swap ? 0u : 1u); a.gauss_out[x + 1] = o1.x; a.slot_out[x + 1] = o1.y; }
        }
        return;
    }
    const uint32_t E = (n + 63u) >> 6;     // elements per lane (1 .. 16)
    const uint32_t Eo = E | 1u;            // odd word stride of a lane's run
    const float rE = 1.0f / (float)E;
    // (p + 1/2) / E is at least 1 / (2 E) >= 1/32 away from every integer: the float quotient truncates to p / E exactly for p < 2^20
    auto phys = [&](uint32_t p) { const uint32_t q = (uint32_t)(((float)p + 0.5f) * rE); return q * Eo + (p - q * E); };
    const uint32_t lo = lane * E;
    const uint32_t mine = lo < n ? (n - lo < E ? n - lo : E) : 0u;   // elements of this lane's run
    const uint32_t run0 = lane * Eo;
    uint32_t* const col = C + lane + (lane >> 4);                      // this lane's counter column: col[68 d] (64 d + l + its pad words)
    int cur = 0;
    // phase 0: by depth.  Only a tile with two equal depths whose order the input does not settle goes on — phase 1: by tie key (from the input
    // order), phase 2: by depth again (from the tie-key order).
#pragma unroll 1
    for (int phase = 0; phase < 3; phase++) {
        uint32_t krange;   // largest key relative to the smallest (wave-uniform)
        {   // all of the wave's global loads in flight at once (a load -> LDS store loop pays one HBM round trip per 64 elements, and there are only two
            // or three waves per SIMD to hide it: that, not the sort, was 80 % of this kernel's first version)
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t e = 64u * (uint32_t)j + lane;
                const bool ok = (uint32_t)j < E && e < n;
                const uint32_t src = (ok && phase == 2) ? (uint32_t)si[cur][phys(e)] : e;
                v[j] = ok ? (phase == 1 ? in.tie(src) : in.depth(src)) : 0u;
            }
            // keys relative to the tile's smallest (order and ties unchanged): the digits above the tile's key RANGE are zero for every key and
            // their passes are skipped (26-27 significant depth bits on the 2M / 1080p scene: seven passes instead of eight)
            uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t e = 64u * (uint32_t)j + lane;
                if ((uint32_t)j < E && e < n) { kmin = v[j] < kmin ? v[j] : kmin; kmax = v[j] > kmax ? v[j] : kmax; }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t a0 = (uint32_t)__shfl_xor((int)kmin, d, 64), a1 = (uint32_t)__shfl_xor((int)kmax, d, 64);
                kmin = a0 < kmin ? a0 : kmin; kmax = a1 > kmax ? a1 : kmax;
            }
            krange = kmax - kmin;
            const int dst = phase == 2 ? cur : 0;
            GS_WAVE_ORDER();
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t e = 64u * (uint32_t)j + lane;
                if ((uint32_t)j < E && e < n) {
                    const uint32_t ph = phys(e);
                    sk[dst][ph] = v[j] - kmin;
                    if (phase != 2) si[dst][ph] = (uint16_t)e;   // (phase 2 re-keys the list in place: position e keeps its index)
                }
            }
            cur = dst;
        }
        GS_WAVE_ORDER();
#pragma unroll 1
        for (int shift = 0; shift < 32; shift += 4) {
            if ((krange >> shift) == 0u) break;   // every remaining digit of every key is zero
            // The lane's run in registers: sixteen independent LDS reads in flight at once (a runtime loop over the run pays one LDS round trip per
            // element and step — 36 per pass — with two waves per SIMD to hide them: measured 187 us for the 2M / 1080p scene).
            uint32_t k[16], ix[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                k[j] = 0u; ix[j] = 0u;
                if ((uint32_t)j < E && (uint32_t)j < mine) { k[j] = sk[cur][run0 + j]; ix[j] = si[cur][run0 + j]; }
            }
#pragma unroll
            for (int d = 0; d < 16; d++) col[68 * d] = 0u;
            GS_WAVE_ORDER();
#pragma unroll
            for (int j = 0; j < 16; j++)
                if ((uint32_t)j < E && (uint32_t)j < mine) atomicAdd(&col[68 * ((k[j] >> shift) & 15u)], 1u);
            GS_WAVE_ORDER();
            {   // exclusive scan of the matrix in (digit, lane) order: lane l takes the sixteen words [16 l, 16 l + 16).  One pad word per sixteen
                // (word f at f + (f >> 4)) puts lane l's run at 17 l: distinct banks for the 32 lanes of a half wave — unpadded, sixteen lanes hit the same
                // bank with every access and the scan alone kept the LDS busy for longer than the rest of the kernel (69 % conflict cycles)
                uint32_t* const cs = C + 17 * lane;
                uint32_t w[16];
#pragma unroll
                for (int i = 0; i < 16; i++) w[i] = cs[i];
                uint32_t tot = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) { const uint32_t t = w[i]; w[i] = tot; tot += t; }
                const uint32_t base = wave_inclusive_scan(tot) - tot;
                GS_WAVE_ORDER();
#pragma unroll
                for (int i = 0; i < 16; i++) cs[i] = w[i] + base;
            }
            GS_WAVE_ORDER();
            uint32_t pos[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {   // (this lane's own counters, in run order: no other lane touches them)
                pos[j] = 0u;
                if ((uint32_t)j < E && (uint32_t)j < mine) pos[j] = atomicAdd(&col[68 * ((k[j] >> shift) & 15u)], 1u);
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if ((uint32_t)j < E && (uint32_t)j < mine) {
                    const uint32_t ph = phys(pos[j]);
                    sk[(cur ^ 1) & (LW_NB - 1)][ph] = k[j];
                    si[(cur ^ 1) & (LW_NB - 1)][ph] = (uint16_t)ix[j];
                }
            }
            GS_WAVE_ORDER();
            cur = (cur ^ 1) & (LW_NB - 1);
        }
        GS_WAVE_SYNC();
        if (phase == 0) {
#ifdef GS_LSORT_NO_TIES
            break;
#endif
            if (!in.two_keys()) break;
            // two equal depths anywhere in the tile?  Every lane looks at its own run of the sorted list (consecutive words: no phys() arithmetic)
            // and at the first element of the next lane's run.
            uint32_t kr[16];
#pragma unroll
            for (int j = 0; j < 16; j++) kr[j] = ((uint32_t)j < E && (uint32_t)j < mine) ? sk[cur][run0 + j] : 0u;
            uint32_t pairs = 0;   // equal neighbours seen by this lane
            uint32_t last = kr[0];
#pragma unroll
            for (int j = 1; j < 16; j++)
                if ((uint32_t)j < E && (uint32_t)j < mine) { pairs += kr[j] == last ? 1u : 0u; last = kr[j]; }
            const uint32_t next_first = (uint32_t)__shfl_down((int)kr[0], 1, 64);
            const uint32_t next_mine = (uint32_t)__shfl_down((int)mine, 1, 64);
            if (lane < 63u && mine > 0u && next_mine > 0u && last == next_first) pairs++;
            if (__ballot(pairs != 0u) == 0ull) break;   // (wave-uniform)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) pairs += (uint32_t)__shfl_xor((int)pairs, d, 64);
            // A few pairs of equal neighbours = a few SHORT runs (eight pairs: at most nine elements in a run): every run is put into tie-key order
            // by the lane that finds its head, by insertion — a tile that sorts by both keys lives three times as long as its neighbours, and one
            // such wave among the last to start (1.6 % of the tiles of the 2M / 1080p scene tie somewhere) kept the whole launch waiting: + 34 us.
            if (pairs <= 8u) {
                for (uint32_t j = lane; j + 1u < n; j += 64u) {
                    const uint32_t k = sk[cur][phys(j)];
                    if (sk[cur][phys(j + 1u)] != k || (j > 0u && sk[cur][phys(j - 1u)] == k)) continue;
                    uint32_t e = j + 1u;
                    while (e + 1u < n && sk[cur][phys(e + 1u)] == k) e++;      // run = [j, e]: this lane owns it (runs are disjoint)
                    for (uint32_t p = j + 1u; p <= e; p++) {
                        const uint16_t ip = si[cur][phys(p)];
                        const uint32_t rp = in.tie(ip);
                        uint32_t q = p;
                        while (q > j) {
                            const uint16_t iq = si[cur][phys(q - 1u)];
                            if (in.tie(iq) <= rp) break;
                            si[cur][phys(q)] = iq;
                            q--;
                        }
                        si[cur][phys(q)] = ip;
                    }
                }
                GS_WAVE_SYNC();
                break;
            }
        }
    }
    {   // the tile's Gaussian ids and emission slots in list order: every gather of the wave in flight before the first store
        uint2 o[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t p = 64u * (uint32_t)j + lane;
            o[j] = make_uint2(0u, 0u);
            if ((uint32_t)j < E && p < n) o[j] = in.out(si[cur][phys(p)]);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t p = 64u * (uint32_t)j + lane;
            if ((uint32_t)j < E && p < n) { a.gauss_out[x + p] = o[j].x; a.slot_out[x + p] = o[j].y; }
        }
    }
}

// Longer segments: one workgroup of 256 threads; pairs in LDS up to LS_CAP instances, in the global scratch arrays beyond.
static constexpr int LS_CAP = 4096;
template <bool BINNED>
__global__ __launch_bounds__(LS_THREADS) void tile_depth_sort_kernel(const TileDepthSortArgs a)
{
    __shared__ uint32_t kL0[LS_CAP], kL1[LS_CAP];
    __shared__ uint16_t iL0[LS_CAP], iL1[LS_CAP];
    __shared__ uint32_t hist[256], run[256], cnt[4][256], red[8];
    if (a.status[2] != 0u) return;   // capacity mode: the instance lists did not fit
    const uint32_t n_long = a.status[GS_FLAG_LONG];   // (the queue's order depends on the run; what is written for a tile does not)
    for (uint32_t q = blockIdx.x; q < n_long; q += gridDim.x) {
        const uint2 range = a.ranges[a.long_tiles[q]];
        const uint32_t x = range.x, n = range.y - range.x;
        if (n <= (uint32_t)LS_CAP) tile_depth_sort_body<true, BINNED>(a, x, n, kL0, kL1, iL0, iL1, hist, run, cnt, red);
        else tile_depth_sort_body<false, BINNED>(a, x, n, kL0, kL1, iL0, iL1, hist, run, cnt, red);
        __syncthreads();
    }
}

int launch_tile_depth_sort(const TileDepthSortArgs& a, hipStream_t s)
{
    if (a.T <= 0) return GSLIC_OK;
    // a fixed grid walks the queue of long lists behind the one-wave kernel (usually empty: its workgroups read one word and leave)
    const dim3 lgrid((unsigned)(a.T < 512 ? a.T : 512));
    if (a.binned) {
        GS_LAUNCH(K_TILE_LSORT, tile_depth_sort_wave_kernel<true>, dim3((unsigned)a.T), dim3(64), 0, s, a);
        GS_LAUNCH(K_TILE_LSORT_LONG, tile_depth_sort_kernel<true>, lgrid, dim3(LS_THREADS), 0, s, a);
    } else {
        GS_LAUNCH(K_TILE_LSORT, tile_depth_sort_wave_kernel<false>, dim3((unsigned)a.T), dim3(64), 0, s, a);
        GS_LAUNCH(K_TILE_LSORT_LONG, tile_depth_sort_kernel<false>, lgrid, dim3(LS_THREADS), 0, s, a);
    }
    return GSLIC_OK;
}

int radix_sort_u32(const SortBuffers& b, const SortPlan& plan, void* scratch, bool onesweep, int id_hist, int id_scatter, hipStream_t s,
                   const uint32_t* n_dev, uint32_t* fault)
{
    if (plan.n == 0) return GSLIC_OK;
    if (plan.n > 0xffffffffull) return set_error(GSLIC_ERR_INVALID_ARG, "radix sort: more than 2^32 elements");
    if (b.v1[0] && b.v2[0]) return sort_impl<3>(b, plan, n_dev, scratch, onesweep, id_hist, id_scatter, s, fault);
    return b.v1[0] ? sort_impl<2>(b, plan, n_dev, scratch, onesweep, id_hist, id_scatter, s, fault)
                   : sort_impl<1>(b, plan, n_dev, scratch, onesweep, id_hist, id_scatter, s, fault);
}

}  // namespace gslic
