// scan.hip — u32 prefix sums (replaces cub::DeviceScan::InclusiveSum, rasterizer_impl.cu:395,441).
// 256 threads x 16 items = 4096 items per block; wave-level scans use 64-lane shuffles.
//  scan_u32          three launches: per-block reduce -> single-block spine scan -> per-block scan with carry-in.
//  scan_u32_chained  ONE launch (every kernel costs ~5 us of dispatch on this part whatever its size, and the forward runs
//                    seven scans): a block posts its tile total as one 8-byte {flag | sum} word (relaxed agent-scope atomic
//                    store, as in radix_sort.hip's onesweep protocol), then its 256 threads poll the words of ALL preceding
//                    tiles in parallel and reduce them to the carry-in.  Tiles are ticketed, so every tile a block waits for
//                    has started.  Reads grow as nb^2 / 2 words, so above 2048 tiles it falls back to the three launches.
#include "gslic_common.h"
#include <stdlib.h>

namespace gslic {

static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_ITEMS = 16;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ void load_raw(const uint32_t* in, size_t base, size_t n, uint32_t (&v)[SCAN_ITEMS])
{
    const size_t t0 = base + (size_t)threadIdx.x * SCAN_ITEMS;
    if (t0 + SCAN_ITEMS <= n && ((reinterpret_cast<uintptr_t>(in + t0) & 15) == 0)) {
        const uint4* p = reinterpret_cast<const uint4*>(in + t0);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; i++) {
            const uint4 q = p[i];
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) v[i] = (t0 + i < n) ? in[t0 + i] : 0u;
    }
}
__device__ __forceinline__ void load_items(const uint32_t* in, size_t base, size_t n, uint32_t (&v)[SCAN_ITEMS],
                                           const uint32_t* gather = nullptr)
{
    if (!gather) { load_raw(in, base, n, v); return; }
    const size_t t0 = base + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t g[SCAN_ITEMS];
    load_raw(gather, base, n, g);  // v[i] = in[gather[i]]
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) v[i] = (t0 + i < n) ? in[g[i]] : 0u;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ gather, size_t n,
                                                                   uint32_t* __restrict__ sums)
{
    __shared__ uint32_t lds[8];
    uint32_t v[SCAN_ITEMS];
    load_items(in, (size_t)blockIdx.x * SCAN_TILE, n, v, gather);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    uint32_t total;
    block256_exclusive_prefix(s, total, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// One block walks the whole array tile by tile (used for n <= one tile and for the spine of block sums).
__global__ __launch_bounds__(SCAN_THREADS) void scan_single_kernel(const uint32_t* in, const uint32_t* __restrict__ gather, uint32_t* out, size_t n,
                                                                   int exclusive)
{
    __shared__ uint32_t lds[8];
    uint32_t carry = 0;
    for (size_t base = 0; base < n; base += SCAN_TILE) {
        uint32_t v[SCAN_ITEMS];
        load_items(in, base, n, v, gather);
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
        uint32_t total;
        uint32_t run = carry + block256_exclusive_prefix(s, total, lds);
        const size_t t0 = base + (size_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const uint32_t x = v[i];
            if (t0 + i < n) out[t0 + i] = exclusive ? run : run + x;
            run += x;
        }
        carry += total;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t* in, const uint32_t* __restrict__ gather, uint32_t* out, size_t n,
                                                                  int exclusive, const uint32_t* __restrict__ block_base)
{
    __shared__ uint32_t lds[8];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
    uint32_t v[SCAN_ITEMS];
    load_items(in, base, n, v, gather);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    uint32_t total;
    uint32_t run = block_base[blockIdx.x] + block256_exclusive_prefix(s, total, lds);
    const size_t t0 = base + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t o[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const uint32_t x = v[i];
        o[i] = exclusive ? run : run + x;
        run += x;
    }
    if (t0 + SCAN_ITEMS <= n && ((reinterpret_cast<uintptr_t>(out + t0) & 15) == 0)) {
        uint4* p = reinterpret_cast<uint4*>(out + t0);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; i++) p[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++)
            if (t0 + i < n) out[t0 + i] = o[i];
    }
}

static constexpr unsigned long long SC_FLAG = 1ull << 63;
static constexpr size_t SC_MAX_TILES = 2048;   // 8.4M elements: the scans over P of a 5M-Gaussian map (1221 tiles) stay one launch — and one gather

// state: u32 ticket (in word 0) | u64 words[nb], all zero on entry
__global__ __launch_bounds__(SCAN_THREADS) void scan_chained_kernel(const uint32_t* in, const uint32_t* __restrict__ gather, uint32_t* out,
                                                                    size_t n, int exclusive, unsigned long long* state, uint32_t* status)
{
    __shared__ uint32_t lds[8];
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(reinterpret_cast<uint32_t*>(state), 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    unsigned long long* const words = state + 1;
    const size_t base = (size_t)tile * SCAN_TILE;
    uint32_t v[SCAN_ITEMS];
    load_items(in, base, n, v, gather);
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) sum += v[i];
    uint32_t total;
    const uint32_t excl = block256_exclusive_prefix(sum, total, lds);
    if (threadIdx.x == 0) __hip_atomic_store(words + tile, SC_FLAG | (unsigned long long)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t acc = 0;
    unsigned long long acc64 = 0;  // the same sum without wrap-around: a total beyond 2^31 is reported, not silently truncated
    for (uint32_t j = threadIdx.x; j < tile; j += SCAN_THREADS) {
        unsigned long long w = __hip_atomic_load(words + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t spins = 0; !(w & SC_FLAG) && spins < (1u << 22); ++spins) {  // bounded: never hang the device
            __builtin_amdgcn_s_sleep(1);
            w = __hip_atomic_load(words + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!(w & SC_FLAG) && status) atomicOr(status, 1u);  // gave up: the sums of this launch are wrong, tell the host
        acc += (uint32_t)w;
        acc64 += w & 0xffffffffull;
    }
    uint32_t carry;
    block256_exclusive_prefix(acc, carry, lds);
    if (status) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc64 += (unsigned long long)__shfl_xor((long long)acc64, d, 64);
        __shared__ unsigned long long s_sum64[SCAN_THREADS / 64];
        if ((threadIdx.x & 63) == 0) s_sum64[threadIdx.x >> 6] = acc64;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t64 = total;
            for (int w2 = 0; w2 < SCAN_THREADS / 64; w2++) t64 += s_sum64[w2];
            if (t64 > 0x7fffffffull) atomicOr(status, 2u);  // bit 1: prefix sum beyond 2^31
        }
    }
    uint32_t run = carry + excl;
    const size_t t0 = base + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t o[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const uint32_t x = v[i];
        o[i] = exclusive ? run : run + x;
        run += x;
    }
    if (t0 + SCAN_ITEMS <= n && ((reinterpret_cast<uintptr_t>(out + t0) & 15) == 0)) {
        uint4* p = reinterpret_cast<uint4*>(out + t0);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS / 4; i++) p[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++)
            if (t0 + i < n) out[t0 + i] = o[i];
    }
}

// tiles' bucket counts and their inclusive scan in one single-block launch (rasterizer_impl.cu:433-441): T is the tile count of an
// image, a few thousand.  1024 threads x 8 tiles: one trip for a 1080p image (8160 tiles) — the kernel is a chain of dependent memory
// round trips, so trips are what it costs (256 x 16 took two: 14 us).
static constexpr int BSC_THREADS = 1024, BSC_ITEMS = 8, BSC_TILE = BSC_THREADS * BSC_ITEMS;
__global__ __launch_bounds__(BSC_THREADS) void bucket_scan_kernel(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ bucket_offsets,
                                                                  uint32_t* __restrict__ max_contrib)
{
    __shared__ uint32_t wsum[BSC_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < T; base += BSC_TILE) {
        const int t0 = base + (int)threadIdx.x * BSC_ITEMS;
        uint32_t v[BSC_ITEMS];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < BSC_ITEMS; i++) {
            uint32_t c = 0;
            if (t0 + i < T) { const uint2 r = ranges[t0 + i]; c = (r.y - r.x + (GS_BUCKET - 1)) / GS_BUCKET; }
            v[i] = c; s += c;
        }
        const uint32_t inc = wave_inclusive_scan(s);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t wave_base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < BSC_THREADS / 64; w++) {
            const uint32_t x = wsum[w];
            if (w < wave) wave_base += x;
            total += x;
        }
        __syncthreads();
        uint32_t run = carry + wave_base + inc - s;
#pragma unroll
        for (int i = 0; i < BSC_ITEMS; i++) {
            run += v[i];
            if (t0 + i < T) { bucket_offsets[t0 + i] = run; max_contrib[t0 + i] = 0u; }  // (render_fwd's waves of a tile combine their maxima with atomicMax)
        }
        carry += total;
    }
}
// (A longest-first launch order of the tiles — counting sort by bucket count in this kernel, render_fwd's workgroups handed out in that
// order — was built and measured in round 3: render_fwd -5 % on a scene of long, faint lists, nothing on the default scene, and the sort
// itself costs this single-block kernel 7 us on every forward: a net loss on four of five workloads.  Not in the tree.)
int launch_bucket_scan(int T, const uint2* ranges, uint32_t* bucket_offsets, uint32_t* max_contrib, hipStream_t s)
{
    GS_LAUNCH(K_BUCKET_COUNT, bucket_scan_kernel, dim3(1), dim3(BSC_THREADS), 0, s, T, ranges, bucket_offsets, max_contrib);
    return GSLIC_OK;
}

size_t scan_temp_elems(size_t n) { return div_up_sz(n, SCAN_TILE) + 64; }
size_t scan_state_bytes(size_t n)
{
    const size_t nb = div_up_sz(n, SCAN_TILE);
    const size_t a = (nb + 2) * sizeof(unsigned long long), b = scan_temp_elems(n) * sizeof(uint32_t);
    return ((a > b ? a : b) + 255) & ~size_t(255);
}

int scan_u32_chained(const uint32_t* in, const uint32_t* gather, uint32_t* out, size_t n, bool exclusive, void* zeroed_state, hipStream_t s,
                     uint32_t* fault)
{
    if (n == 0) return GSLIC_OK;
    const size_t nb = div_up_sz(n, SCAN_TILE);
    if (nb == 1) {
        GS_LAUNCH(K_SCAN_SPINE, scan_single_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, in, gather, out, n, exclusive ? 1 : 0);
        return GSLIC_OK;
    }
    if (nb > SC_MAX_TILES) {
        uint32_t* temp = reinterpret_cast<uint32_t*>(zeroed_state);
        GS_LAUNCH(K_SCAN_REDUCE, scan_reduce_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, gather, n, temp);
        GS_LAUNCH(K_SCAN_SPINE, scan_single_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, (const uint32_t*)temp, (const uint32_t*)nullptr, temp, nb, 1);
        GS_LAUNCH(K_SCAN_APPLY, scan_apply_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, gather, out, n, exclusive ? 1 : 0,
                  (const uint32_t*)temp);
        return GSLIC_OK;
    }
    GS_LAUNCH(K_SCAN_APPLY, scan_chained_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, gather, out, n, exclusive ? 1 : 0,
              reinterpret_cast<unsigned long long*>(zeroed_state), fault ? fault : device_status_word());
    return GSLIC_OK;
}


int scan_u32(const uint32_t* in, uint32_t* out, size_t n, bool exclusive, uint32_t* temp, hipStream_t s)
{
    if (n == 0) return GSLIC_OK;
    const size_t nb = div_up_sz(n, SCAN_TILE);
    if (nb == 1) {
        GS_LAUNCH(K_SCAN_SPINE, scan_single_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, in, (const uint32_t*)nullptr, out, n, exclusive ? 1 : 0);
        return GSLIC_OK;
    }
    GS_LAUNCH(K_SCAN_REDUCE, scan_reduce_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, (const uint32_t*)nullptr, n, temp);
    GS_LAUNCH(K_SCAN_SPINE, scan_single_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, (const uint32_t*)temp, (const uint32_t*)nullptr, temp, nb, 1);
    GS_LAUNCH(K_SCAN_APPLY, scan_apply_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, (const uint32_t*)nullptr, out, n, exclusive ? 1 : 0,
              (const uint32_t*)temp);
    return GSLIC_OK;
}

}  // namespace gslic
