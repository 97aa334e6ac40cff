// gslic_common.h — shared declarations of libgslic_hip.so (gfx950 only; wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gslic_hip.h"

#define GS_TILE 16        // tile edge in pixels (config.h:16-17 of the reference; part of the tile-list contract)
#define GS_TILE_PIX 256   // pixels per tile
#ifndef GS_REC_F4
#define GS_REC_F4 3       // float4 slots per Gaussian record.  4 (64-byte stride: a record never straddles a 128-byte line) was measured:
                          // keybuild 0.0905 -> 0.089 ms, preprocess +2.5 %, render_bwd +1 % — the depth-order gather is not what the stride fixes
#endif
#ifndef GS_PROW
#define GS_PROW 9         // floats per partial-gradient row of an instance (BinningState::partials).  12 / 16 (48- / 64-byte rows: fewer line straddles for the
                          // scattered row stores of the blend backward, more bytes for the per-Gaussian kernel to read) were measured: render_bwd 0.449 -> 0.448 / 0.445 ms, no gain (profiles/r04y_bwd_scan_by_elimination.log)
#endif
#define GS_BUCKET 64      // checkpoint period = one wave of list entries (reference: 32 = one CUDA warp)
#define GS_WAVE 64

namespace gslic {

// ---------------------------------------------------------------------------------------------------------
// errors
int set_error(int code, const char* fmt, ...);
#define GS_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess)                                                                             \
            return ::gslic::set_error(GSLIC_ERR_HIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e__),      \
                                      __FILE__, __LINE__);                                                 \
    } while (0)
#define GS_TRY(expr)                                                                                       \
    do {                                                                                                   \
        int rc__ = (expr);                                                                                 \
        if (rc__ != GSLIC_OK) return rc__;                                                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// per-kernel profiling (HIP events on the launch stream)
enum KernelId {
    K_PREPROCESS = 0,
    K_SCAN_REDUCE,
    K_SCAN_SPINE,
    K_SCAN_APPLY,
    K_KEYBUILD,
    K_SORT_HIST,
    K_SORT_SCATTER,
    K_FINALIZE_LISTS,
    K_BUCKET_COUNT,
    K_RENDER_FWD,
    K_RENDER_BWD,
    K_PREPROCESS_BWD,
    K_ADAM,
    K_SSIM_FWD,
    K_SSIM_BWD,
    K_KNN_MINMAX,
    K_KNN_MORTON,
    K_KNN_BOXES,
    K_KNN_SEARCH,
    K_DEBUG_EXPORT,
    K_EXTEND,
    K_DSORT_HIST,
    K_DSORT_SCATTER,
    K_SH_REBUILD,
    K_TILE_LSORT,
    K_TILE_HIST,
    K_TILE_BIN,
    K_TILE_SCAN,
    K_TILE_LSORT_LONG,
    K_COUNT
};
void prof_begin(int id, hipStream_t s);
void prof_end(int id, hipStream_t s);
extern bool g_prof_on;
#define GS_FLAG_HITBITS 4   // index into GeomState::flags: set by a forward that recorded SampleState::hit
#define GS_FLAG_BIN_ATOMICS 10   // index into GeomState::flags: [10] global atomics of a SAMPLE of the binning kernel's workgroups (their distinct tiles), [11] the
                                 // instances of those workgroups: what grouping by tile without a sort costs on this map's row order — api.hip picks the path by it
#define GS_FLAG_LONG 9      // index into GeomState::flags: number of tiles whose instance list is longer than the one-wave depth sort takes (radix_sort.hip)
#define GS_FLAG_FAULT 8     // index into GeomState::flags: bit 0 a bounded look-back wait of THIS forward's scans / sorts gave up, bit 1 its instance count
                            // overflowed.  Per forward (the flags are zeroed at its start), so forwards running concurrently on several streams of
                            // one device cannot consume each other's bits — round 4 kept them in one device-global word (device_status_word)
extern int g_strict_math;  // gslic_set_math_mode(): 1 (default) = blend kernels in the reference's arithmetic (render.hip), 0 = fast (GSLIC_FAST_MATH=1)

// g_lds_pad[id]: extra dynamic LDS per workgroup of kernel class id — occupancy experiments only (GSLIC_LDS_PAD="render_bwd=4000,preprocess=2000",
// names as in the profiler's table; all zero by default): how much a kernel loses per wave of occupancy says what an LDS diet would buy.
extern uint32_t g_lds_pad[];
#define GS_LAUNCH(id, kernel, grid, block, shmem, stream, ...)                                             \
    do {                                                                                                   \
        if (::gslic::g_prof_on) ::gslic::prof_begin(id, stream);                                           \
        hipLaunchKernelGGL(kernel, grid, block, (shmem) + ::gslic::g_lds_pad[id], stream, __VA_ARGS__);    \
        if (::gslic::g_prof_on) ::gslic::prof_end(id, stream);                                             \
        GS_HIP(hipGetLastError());                                                                         \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// scratch-buffer carving: 256-B aligned bump allocation inside one caller-provided byte buffer
struct Carver {
    uintptr_t p;
    explicit Carver(const void* base) : p(reinterpret_cast<uintptr_t>(base)) {}
    template <typename T>
    T* take(size_t count)
    {
        p = (p + 255) & ~uintptr_t(255);
        T* r = reinterpret_cast<T*>(p);
        p += count * sizeof(T);
        return r;
    }
    size_t used(const void* base) const { return p - reinterpret_cast<uintptr_t>(base); }
};

static inline int div_up(int a, int b) { return (a + b - 1) / b; }
static inline size_t div_up_sz(size_t a, size_t b) { return (a + b - 1) / b; }

// rasterizer_impl.cu:42-57: number of bits needed for the tile id in the sort key
static inline uint32_t higher_msb(uint32_t n)
{
    uint32_t bits = 0;
    while (bits < 32 && (n >> bits) != 0) ++bits;
    return bits == 0 ? 1 : bits;
}

// Device status word of the current device (lazily allocated, zero): bit 0 = a chained-scan / look-back spin-wait gave up (the
// waits are bounded so that a preempted or debugged device never hangs; the prefix sums of that launch are then wrong and
// the forward reports GSLIC_ERR_HIP instead of using them).  NULL if the allocation failed.
uint32_t* device_status_word();

// scan.hip ------------------------------------------------------------------------------------------------
size_t scan_temp_elems(size_t n);  // u32 elements of scratch needed by scan_u32 for n inputs
// inclusive (or exclusive) prefix sum of n u32 values; in == out allowed
int scan_u32(const uint32_t* in, uint32_t* out, size_t n, bool exclusive, uint32_t* temp, hipStream_t s);
// the same in ONE launch (chained tiles); optional gather: scans in[gather[i]].  `zeroed_state`: scan_state_bytes(n) bytes the
// caller has zeroed on the stream (it is consumed: zero it again before the next scan)
size_t scan_state_bytes(size_t n);
// fault: the word that receives the timeout / overflow bits (NULL: the device-global word — callers outside a forward: knn, extend)
int scan_u32_chained(const uint32_t* in, const uint32_t* gather, uint32_t* out, size_t n, bool exclusive, void* zeroed_state, hipStream_t s,
                     uint32_t* fault = nullptr);

// radix_sort.hip ------------------------------------------------------------------------------------------
#define GS_SORT_ITEMS 16
#define GS_SORT_BLOCK 256
#define GS_SORT_TILE (GS_SORT_ITEMS * GS_SORT_BLOCK)
struct SortPlan {
    size_t n;
    int passes;      // ceil(end_bit / 8), at most 4
    size_t nblk;     // ceil(n / GS_SORT_TILE)
    size_t hist_elems;  // 256 * nblk
};
SortPlan sort_plan(size_t n, int end_bit);
// Stable LSD radix sort of u32 keys on bits [0, end_bit) carrying one (v1[0] == NULL) or two u32 payloads.  Ping-pongs between
// the two buffer sets starting from index 0; the result is in buffers [plan.passes & 1].  v0_identity: the first payload starts
// as the element index (argsort) and v0[0] is never read.  `scratch`: sort_scratch_bytes(plan) bytes.  onesweep selects the
// decoupled-look-back variant (same result bit for bit).  id_*: profiling slots the launches are booked under.
struct SortBuffers {
    uint32_t* keys[2];
    uint32_t* v0[2];
    uint32_t* v1[2];   // NULL: one payload
    uint32_t* v2[2];   // NULL: at most two payloads
    bool v0_identity;
};
size_t sort_scratch_bytes(const SortPlan& plan);
// n_dev != NULL: the real element count (<= plan.n, which is then the capacity the launches are sized for) is read on the device.
int radix_sort_u32(const SortBuffers& b, const SortPlan& plan, void* scratch, bool onesweep, int id_hist, int id_scatter, hipStream_t s,
                   const uint32_t* n_dev = nullptr, uint32_t* fault = nullptr);
// Level 2 of the binning (round 5): every tile's segment [ranges[t].x, ranges[t].y) of the tile-sorted instance list, which is in emission
// (= Gaussian index) order inside a tile, is sorted by depth bits — stable, so equal depths stay in index order, or in tie_rank order when the
// map's rows are stored permuted — by ONE workgroup per tile: an LSD radix sort of (depth, local index) pairs in LDS for segments of up to
// LS_CAP instances, through the four scratch arrays for longer ones.  gauss_out / slot_out receive the tile's Gaussian ids / emission slots in
// the reference's list order (rasterizer_impl.cu:419-424: stable 64-bit sort of tile << 32 | depth over the index-ordered emission).
// Grouping the instances by tile WITHOUT a sort (tile_bin.hip): block-aggregated atomics on the tiles' list cursors.  Any order inside a tile
// will do, because the per-tile depth sort orders a tile's instances by (depth, tie key) — a total order.
struct TileBinArgs {
    int T;
    uint32_t n_cap;             // instances the launch is sized for (capacity mode: the buffer's capacity; the exact R otherwise)
    const uint32_t* n_dev;      // capacity mode: the real instance count, on the device (NULL: n_cap)
    const uint32_t* tile;       // [R] tile id per emission slot (keybuild)
    const uint32_t* gid;        // [R] Gaussian id per emission slot
    const uint32_t* depth;      // [R] depth bits per emission slot
    const uint32_t* tie_rank;   // optional [P]: gslic_raster_params.tie_rank (the tie key of an instance is tie_rank[gid], or gid without it)
    uint2* ranges;              // [T] zero on entry (preprocess_kernel); the tiles' list ranges on exit, (0, 0) for an empty tile
    uint4* binned;              // [R] out, grouped by tile: {depth bits, tie key, Gaussian id, emission slot}
    uint8_t* dead;              // optional [R]: the backward's per-slot dead flags, cleared here (a coalesced byte store riding on the pass)
    uint32_t* bucket_offsets;   // optional [T] out: inclusive scan of the tiles' 64-entry bucket counts, and with it
    uint32_t* max_contrib;      // [T] out: zeroed (ImageState: what launch_bucket_scan does on the radix path)
    uint32_t* status;           // device status words: a non-zero [2] (capacity overflow) makes the kernels return
};
static constexpr int GS_TILE_BIN_MAX_T = 36864;   // tiles the block histogram of the binning kernels holds in LDS (4 bytes each, 144 of 160 KB)
int launch_tile_bin(const TileBinArgs& a, hipStream_t s);
bool tile_bin_lds_ok(int T);   // the device grants the dynamic LDS the binning kernels need at T tiles (asked once per device)

struct TileDepthSortArgs {
    int T;
    const uint2* ranges;
    const uint4* binned;        // non-NULL: the instances grouped by tile in ANY order inside a tile (TileBinArgs::binned) — depth, gauss_in and
                                // slot_in are then not read, and equal depths are put into tie-key order by a second key (no input order to keep);
                                // NULL: depth / gauss_in / slot_in sorted by tile, in index order inside a tile (the stable tile sort's output)
    const uint32_t* depth;      // [R] sorted by tile (the third payload of the tile sort)
    uint32_t* key_a;            // [R] each: scratch of the segments longer than LS_CAP (keys and indices, ping-pong); they alias no input and no output
    uint32_t* key_b;
    uint32_t* idx_a;
    uint32_t* idx_b;
    const uint32_t* gauss_in;   // [R] sorted by tile
    const uint32_t* slot_in;    // [R] sorted by tile
    uint32_t* gauss_out;        // [R] the point list
    uint32_t* slot_out;         // [R] emission slot of every list entry
    const uint32_t* tie_rank;   // optional [P]: gslic_raster_params.tie_rank
    uint32_t* status;           // device status words (GeomState::flags): a non-zero [2] (capacity overflow) makes the kernels return;
                                // [GS_FLAG_LONG] counts the tiles whose list is longer than one wave sorts (zero at the start of a forward)
    uint32_t* long_tiles;       // [T] scratch: ids of those tiles (ImageState::bucket_offsets, rewritten by the bucket scan afterwards)
};
int launch_tile_depth_sort(const TileDepthSortArgs& a, hipStream_t s);

// opaque scratch layouts ----------------------------------------------------------------------------------
struct GeomState {
    float4* rec;             // [3P] per Gaussian: {mx,my,conic.x,conic.y} {conic.z,opacity,r,g} {b,depth,clamp bits 0-2 | tile-test mask 16-31,radius}
    uint32_t* tiles_touched; // [P]
    float* cam_scratch;      // [128 ceil(P / 256) + 64] the camera-gradient backward's per-wave partial rows and its 35 results (gslic_rasterize_backward_camera)
    uint32_t* point_offsets; // [P] inclusive scan of tiles_touched in index order: the emission slots of Gaussian i end here
    uint32_t* gauss_start;   // [P] first emission slot of Gaussian g (written for visible Gaussians only)
    uint32_t* flags;         // [64] device-side status words ([0] = prefiltered violation), zeroed together with scan_state
    void* scan_state;        // chained-scan state of the emission-slot scan (directly behind flags)
    size_t zero_bytes;       // flags + scan_state
    static GeomState carve(const void* base, size_t P, size_t* bytes);
};
struct ImageState {
    uint2* ranges;            // [T]
    uint32_t* bucket_offsets; // [T] inclusive scan of ceil(n_t / GS_BUCKET)
    uint32_t* max_contrib;    // [T]
    float4* pix_final;        // [T*256] tile-major {C.r,C.g,C.b, n_contrib bits}
    static ImageState carve(const void* base, size_t T, size_t* bytes);
};
struct BinningState {
    // Emission order -> lists, on either grouping path (api.hip: binning_choice).  pp = plan.passes & 1 names the side the lists do NOT go to.
    //   radix path : keybuild -> tile_keys[0] / gauss[0] / lsort[0] (tile, Gaussian id, depth bits); the stable tile sort ping-pongs all of them with
    //                the slots and leaves tile / slot / id / depth sorted by tile in tile_keys[pp] / slots[pp] / gauss[pp] / lsort[pp]; the per-tile sort's
    //                long lists use lsort[pp ^ 1], tile_keys[pp ^ 1], lsort[2], lsort[3] as scratch
    //   atomic path: keybuild -> tile_keys[0] / gauss[pp] / slots[pp]; tile_bin -> binned() (16-byte rows grouped by tile); the per-tile sort's long lists
    //                use tile_keys[0], tile_keys[1], gauss[pp], slots[pp] (all dead by then) and queue their tiles in sort_scratch
    //   both       : the per-tile sort writes the point list to gauss[pp ^ 1] and the slot list to slots[pp ^ 1]
    uint32_t* tile_keys[2];   // [R] each
    uint32_t* slots[2];       // [R] each: emission slot u of an instance = where the backward writes its partial row
    uint32_t* gauss[2];       // [R] each
    uint32_t* lsort[4];       // [R] each, CONTIGUOUS (= the 16 R bytes of binned()).  They alias the first 16 R bytes of `partials` (dead during the
                              // forward) when there is a backward (!no_color); a transmittance-only forward carves them
    void* sort_scratch;
    float* partials;          // [9R] per emission slot: the instance's 9 partial gradients (36-byte rows), only when !no_color
    uint8_t* dead;            // [R] per emission slot: 1 = the instance lies in a bucket behind its tile's last contributor (its partial row is
                              // NOT written and must not be read: all nine gradients are exactly zero); zeroed by finalize_ranges_kernel / tile_bin_kernel
    SortPlan plan;
    uint32_t* point_list() const { return gauss[(plan.passes & 1) ^ 1]; }    // (written by the per-tile depth sort)
    uint32_t* inst_slot() const { return slots[(plan.passes & 1) ^ 1]; }
    uint4* binned() const { return reinterpret_cast<uint4*>(lsort[0]); }       // [R] 16-byte rows of the atomic binning (TileBinArgs): the four lsort arrays
    static BinningState carve(const void* base, size_t R, int end_bit, bool no_color, size_t* bytes);
};
struct SampleState {
    uint32_t* bucket_to_tile; // [B]
    float4* ckpt;             // [B*256] {T, C.r, C.g, C.b} at the start of each bucket, per pixel
    uint64_t* hit;            // [B*256] per bucket and pixel (tile-major element order, like ckpt): bit j = the pixel blended entry j of the
                              // bucket — written by the strict forward, read by the strict backward (its blend / skip decisions)
    static SampleState carve(const void* base, size_t B, size_t* bytes);
};

}  // namespace gslic

// =========================================================================================================
// Device-side canonical arithmetic for everything that decides an integer (radii, tiles_touched, keys).
// Must be compiled with -ffp-contract=off; mirrors oracle/gs_oracle.c operation by operation.
// =========================================================================================================
#if defined(__HIPCC__)
namespace gslic {

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Pixel of element i (0..255) of the tile-major per-pixel arrays (pix_final, checkpoints, decision masks), inside its 16x16 tile: four
// 8x8 quadrants of 64 elements each — quadrant Q = i >> 6 at (8 (Q & 1), 8 (Q >> 1)), element l = i & 63 of it at (l & 7, l >> 3).
// render_fwd's lane l owns element l of every quadrant of its wave (8x8 quadrants cull list entries better than 16x4 strips).
__device__ __forceinline__ int tile_pix_x(int i) { return ((i >> 6) & 1) * 8 + (i & 7); }
__device__ __forceinline__ int tile_pix_y(int i) { return (i >> 7) * 8 + ((i >> 3) & 7); }
__device__ __forceinline__ int tile_pix_index(int x, int y) { return ((y >> 3) * 2 + (x >> 3)) * 64 + (y & 7) * 8 + (x & 7); }


// 64-lane inclusive scan and 256-thread block exclusive prefix (lds: >= 4 u32, reusable after return)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ __forceinline__ uint32_t block256_exclusive_prefix(uint32_t thread_sum, uint32_t& block_total, uint32_t* lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(thread_sum);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t s = lds[w];
        if (w < wave) wave_base += s;
        total += s;
    }
    __syncthreads();
    block_total = total;
    return wave_base + inc - thread_sum;
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// Streaming accesses: data that is read once or written once per launch (Adam moments, per-instance partial gradients) goes past the
// L2's retention — measured on this part (tools/ubench/hbm_rate): copy 4.9 -> 5.7 TB/s, Adam-shaped read-modify-write 5.98 -> 6.29.
// GSLIC_NT = 0 at compile time turns them back into plain accesses (A/B runs).
#ifndef GSLIC_NT
#define GSLIC_NT 1
#endif
typedef float gs_v4f __attribute__((ext_vector_type(4)));
typedef float gs_v4f_u __attribute__((ext_vector_type(4), aligned(4)));   // four floats at a dword-aligned address (the 36-byte partial rows)
__device__ __forceinline__ float4 ld_stream(const float4* p)
{
#if GSLIC_NT
    const gs_v4f v = __builtin_nontemporal_load(reinterpret_cast<const gs_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(float4* p, const float4 v)
{
#if GSLIC_NT
    const gs_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<gs_v4f*>(p));
#else
    *p = v;
#endif
}

// The Adam update of adam.cu:26-37, one definition for every kernel that applies it (adam.hip and the fused backward):
// contraction is pinned off so that the two call sites round identically (their results are compared bit for bit).
__device__ __forceinline__ void adam_scalar(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps)
{
#pragma clang fp contract(off)
    m = b1 * m + (1.0f - b1) * g;
    v = b2 * v + (1.0f - b2) * g * g;
    p += -lr * m / (sqrtf(v) + eps);
}

// expf() as hipcc lowers it for gfx950 (the AMDGPU lowering of llvm.exp.f32: x log2(e) split into a rounded head PH and a tail PL by
// two FMAs, E = rint(PH), v_exp_f32((PH - E) + PL), ldexp by E — see the disassembly of any expf() call under -fno-fast-math), minus
// its two range checks (x < -103.28 -> 0, x > 88.72 -> inf).  Those never decide anything in a blend: a pair with power > 0 is skipped
// before exp is looked at, and below -103.28 both this sequence (a denormal or 0 out of v_ldexp_f32) and the library (0) give an
// alpha far under 1/255.  Everywhere else the two are the same instructions on the same operands: bit-identical results
// (tools/ubench/expf_replica checks every float in [-104, 0] on the device).  kL2E / kCC live in VGPRs at the call sites (a literal
// operand doubles the issue cost of a VALU instruction on this part).
__device__ __forceinline__ float expf_core(float x, float kL2E, float kCC)
{
#pragma clang fp contract(off)
    const float ph = x * kL2E;
    const float pl = __builtin_fmaf(x, kCC, __builtin_fmaf(x, kL2E, -ph));
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + pl;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
#define GS_EXP_L2E 0x1.715476p+0f   /* log2(e) rounded to float: 0x3fb8aa3b */
#define GS_EXP_CC 0x1.4ae0bep-26f   /* log2(e) - (float)log2(e): 0x32a5705f */

// Culling threshold (forward.cu:302, rasterizer_impl.cu:89): logf(opacity / (1/255)) with the toolchain's own logf, i.e. the very
// instruction sequence the reference's kernels get from hipcc (v_log_f32 + the two-FMA ln2 tail): a tile count can then never differ from
// theirs.  (The CPU oracle has a correctly rounded logf, <= 1 ulp away: oracle/gs_oracle.c, orc_logf.)
__device__ __forceinline__ float cull_threshold(float opacity) { return logf(opacity / (1.0f / 255.0f)); }

__device__ __forceinline__ int trunc_clamped(float v, int hi)
{
    if (!(v > -1.0f)) return 0;
    if (v >= (float)(hi + 1)) return hi;
    int i = (int)v;
    return i < 0 ? 0 : (i > hi ? hi : i);
}

// auxiliary.h:46-56
__device__ __forceinline__ void get_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1)
{
    const float r = (float)radius;
    x0 = trunc_clamped((px - r) / (float)GS_TILE, gx);
    y0 = trunc_clamped((py - r) / (float)GS_TILE, gy);
    // the reference's LITERAL order, ((p + radius) + BLOCK) - 1 (auxiliary.h:52-53: `p.x + max_radius + BLOCK_X - 1`), not p + radius + 15: the two
    // round differently when (p + radius) + 16 is a tie of the next binade — 215.99998 + 25 + 16 = 257.0 exactly, so the rectangle reaches tile column
    // 15; + 15 gives 255.99998 and stops at 14 (found by fuzz scene 845806 at the end of round 6: one Gaussian in about 3e7 lost a tile it blends into)
    x1 = trunc_clamped((((px + r) + (float)GS_TILE) - 1.0f) / (float)GS_TILE, gx);
    y1 = trunc_clamped((((py + r) + (float)GS_TILE) - 1.0f) / (float)GS_TILE, gy);
}

__device__ __forceinline__ float saturate_f(float v) { return (v > 0.0f) ? ((v < 1.0f) ? v : 1.0f) : 0.0f; }

// forward.h:39-78: min over the tile's pixel-centre rectangle of 1/2 d^T Q d; co = (A, B, C) conic.
// The rectangle's edge lengths are GS_TILE - 1 = 15 for every tile (rmax - rmin of exactly representable integers), so the two
// correctly-rounded divisions 1 / (s * s * A), 1 / (s * s * C) of the reference formula depend on the Gaussian alone: callers that
// test many tiles of one Gaussian compute them once (tile_power_prep) — same values bit for bit, 20 VALU instructions less per tile.
__device__ __forceinline__ void tile_power_prep(float cA, float cC, float& rcpx, float& rcpy)
{
    const float s = (float)(GS_TILE - 1);
    rcpx = 1.0f / (s * s * cA);
    rcpy = 1.0f / (s * s * cC);
}
__device__ __forceinline__ float tile_min_power_p(float cA, float cB, float cC, float mx, float my, float rcpx, float rcpy, int tx, int ty)
{
    // (operation order is part of the tile-list contract: every sum and product below is rounded where the reference's is)
    const float lo_x = (float)(tx * GS_TILE), lo_y = (float)(ty * GS_TILE);                          // first pixel centre of the tile
    const float hi_x = (float)((tx + 1) * GS_TILE - 1), hi_y = (float)((ty + 1) * GS_TILE - 1);      // last one
    const float gap_x = lo_x - mx;
    const float left_of = (gap_x > 0.0f) ? 1.0f : 0.0f;                    // the mean lies left of the tile
    const float outside_x = left_of + ((mx > hi_x) ? 1.0f : 0.0f);         // ... or right of it
    const float gap_y = lo_y - my;
    const float above = (gap_y > 0.0f) ? 1.0f : 0.0f;
    const float outside_y = above + ((my > hi_y) ? 1.0f : 0.0f);
    if (!((outside_y + outside_x) > 0.0f)) return 0.0f;                    // mean inside the rectangle: the minimum is 0 at the mean
    const float span_x = hi_x - lo_x, span_y = hi_y - lo_y;                // = GS_TILE - 1
    const float near_x = left_of * lo_x + (1.0f - left_of) * hi_x;         // the rectangle's corner nearest to the mean
    const float near_y = above * lo_y + (1.0f - above) * hi_y;
    const float step_x = copysignf(span_x, gap_x);                         // direction along each edge away from that corner
    const float step_y = copysignf(span_y, gap_y);
    const float off_x = mx - near_x;
    const float off_y = my - near_y;
    // slide along the nearest edge(s): parameter of the 1-D minimum, clamped to the edge (NaN -> 0 through saturate_f)
    const float u = outside_y * saturate_f((step_x * cA * off_x + step_x * cB * off_y) * rcpx);
    const float v = outside_x * saturate_f((step_y * cB * off_x + step_y * cC * off_y) * rcpy);
    const float qx = near_x + u * step_x, qy = near_y + v * step_y;        // closest point of the rectangle in the conic's metric
    const float ex = mx - qx, ey = my - qy;
    return 0.5f * (cA * ex * ex + cC * ey * ey) + cB * ex * ey;
}
__device__ __forceinline__ float tile_min_power(float cA, float cB, float cC, float mx, float my, int tx, int ty)
{
    float rcpx, rcpy;
    tile_power_prep(cA, cC, rcpx, rcpy);
    return tile_min_power_p(cA, cB, cC, mx, my, rcpx, rcpy, tx, ty);
}

}  // namespace gslic
#endif
