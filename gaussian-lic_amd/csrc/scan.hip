// scan.hip — u32 prefix sums (replaces cub::DeviceScan::InclusiveSum, rasterizer_impl.cu:395,441).
// Three launches: per-block reduce -> single-block spine scan -> per-block scan with carry-in.
// 256 threads x 16 items = 4096 items per block; wave-level scans use 64-lane shuffles.
#include "gslic_common.h"

namespace gslic {

static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_ITEMS = 16;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ void load_items(const uint32_t* in, size_t base, size_t n, uint32_t (&v)[SCAN_ITEMS])
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

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t lds[8];
    uint32_t v[SCAN_ITEMS];
    load_items(in, (size_t)blockIdx.x * SCAN_TILE, n, v);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += v[i];
    uint32_t total;
    block256_exclusive_prefix(s, total, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// One block walks the whole array tile by tile (used for n <= one tile and for the spine of block sums).
__global__ __launch_bounds__(SCAN_THREADS) void scan_single_kernel(const uint32_t* in, uint32_t* out, size_t n, int exclusive)
{
    __shared__ uint32_t lds[8];
    uint32_t carry = 0;
    for (size_t base = 0; base < n; base += SCAN_TILE) {
        uint32_t v[SCAN_ITEMS];
        load_items(in, base, n, v);
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

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t* in, uint32_t* out, size_t n, int exclusive,
                                                                  const uint32_t* __restrict__ block_base)
{
    __shared__ uint32_t lds[8];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
    uint32_t v[SCAN_ITEMS];
    load_items(in, base, n, v);
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

size_t scan_temp_elems(size_t n) { return div_up_sz(n, SCAN_TILE) + 64; }

int scan_u32(const uint32_t* in, uint32_t* out, size_t n, bool exclusive, uint32_t* temp, hipStream_t s)
{
    if (n == 0) return GSLIC_OK;
    const size_t nb = div_up_sz(n, SCAN_TILE);
    if (nb == 1) {
        GS_LAUNCH(K_SCAN_SPINE, scan_single_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, in, out, n, exclusive ? 1 : 0);
        return GSLIC_OK;
    }
    GS_LAUNCH(K_SCAN_REDUCE, scan_reduce_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, n, temp);
    GS_LAUNCH(K_SCAN_SPINE, scan_single_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, (const uint32_t*)temp, temp, nb, 1);
    GS_LAUNCH(K_SCAN_APPLY, scan_apply_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, n, exclusive ? 1 : 0,
              (const uint32_t*)temp);
    return GSLIC_OK;
}

}  // namespace gslic
