// api.hip — the C-ABI of include/gslic_hip.h: argument validation, scratch carving, stage sequencing,
// per-kernel HIP-event profiling and the test-only debug export.  Stage sequencing follows
// CudaRasterizer::Rasterizer::forward/backward (rasterizer_impl.cu:312-581); kernels live in the other .hip files.
#include "gslic_common.h"
#include "kernels.h"
#include <atomic>
#include <chrono>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace gslic {

// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---------------------------------------------------------------------------------------------------------
bool g_prof_on = false;
// Arithmetic of the two blend kernels: 1 (default) = the reference's operations in source order (bit-identical image / final_T /
// n_contrib to the reference kernels), 0 = the fast variant (opt-in: GSLIC_FAST_MATH=1, or the older GSLIC_STRICT_MATH=0).
int g_strict_math = [] {
    const char* s = getenv("GSLIC_STRICT_MATH");
    if (s) return atoi(s) != 0 ? 1 : 0;
    const char* f = getenv("GSLIC_FAST_MATH");
    return (f && atoi(f) != 0) ? 0 : 1;
}();
static uint32_t g_prof_mask = 0xffffffffu;
static bool g_prof_cur_on = false;
static const char* const k_names[K_COUNT] = {
    "preprocess", "scan_reduce", "scan_spine", "scan_apply", "keybuild", "sort_hist", "sort_scatter", "finalize_lists",
    "bucket_count", "render_fwd", "render_bwd", "preprocess_bwd", "adam", "ssim_fwd", "ssim_bwd", "knn_minmax", "knn_morton",
    "knn_boxes", "knn_search", "debug_export", "extend", "dsort_hist", "dsort_scatter", "sh_grad_from_rgb", "tile_lsort", "tile_hist", "tile_bin", "tile_scan", "tile_lsort_long"};
uint32_t g_lds_pad[K_COUNT] = {0};
static const bool g_lds_pad_parsed = [] {   // GSLIC_LDS_PAD="name=bytes,name=bytes"
    const char* e = getenv("GSLIC_LDS_PAD");
    while (e && *e) {
        const char* eq = strchr(e, '=');
        if (!eq) break;
        for (int k = 0; k < K_COUNT; k++)
            if (strlen(k_names[k]) == (size_t)(eq - e) && strncmp(k_names[k], e, (size_t)(eq - e)) == 0) g_lds_pad[k] = (uint32_t)atoi(eq + 1);
        const char* c = strchr(eq, ',');
        e = c ? c + 1 : nullptr;
    }
    return true;
}();
struct ProfRec { int id; hipEvent_t a, b; };
static std::vector<ProfRec> g_pending;
static std::vector<hipEvent_t> g_pool;
static double g_total_ms[K_COUNT];
static int64_t g_launches[K_COUNT];
static hipEvent_t g_cur = nullptr;

static hipEvent_t ev_get()
{
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void prof_begin(int id, hipStream_t s)
{
    g_prof_cur_on = (g_prof_mask >> id) & 1u;
    if (!g_prof_cur_on) return;
    g_cur = ev_get();
    (void)hipEventRecord(g_cur, s);
}
void prof_end(int id, hipStream_t s)
{
    if (!g_prof_cur_on) return;
    hipEvent_t b = ev_get();
    (void)hipEventRecord(b, s);
    g_pending.push_back({id, g_cur, b});
    g_cur = nullptr;
}

// ---------------------------------------------------------------------------------------------------------
GeomState GeomState::carve(const void* base, size_t P, size_t* bytes)
{
    Carver c(base);
    GeomState g;
    g.rec = c.take<float4>(GS_REC_F4 * P);
    g.tiles_touched = c.take<uint32_t>(P);
    g.cam_scratch = c.take<float>(128 * ((P + 255) / 256) + 64);   // 32 floats per wave of 4 ceil(P / 256) waves + the 35 results (rounds 2-4 kept the Gaussians' depth-sort arrays here: 16 P bytes)
    g.point_offsets = c.take<uint32_t>(P);
    g.gauss_start = c.take<uint32_t>(P);
    g.flags = c.take<uint32_t>(64);  // 256 B, so that scan_state follows directly
    g.scan_state = c.take<char>(scan_state_bytes(P));
    g.zero_bytes = 64 * sizeof(uint32_t) + scan_state_bytes(P);
    if (bytes) *bytes = c.used(base) + 256;
    return g;
}
ImageState ImageState::carve(const void* base, size_t T, size_t* bytes)
{
    Carver c(base);
    ImageState g;
    g.ranges = c.take<uint2>(T);
    g.bucket_offsets = c.take<uint32_t>(T);
    g.max_contrib = c.take<uint32_t>(T);
    g.pix_final = c.take<float4>(T * GS_TILE_PIX);
    if (bytes) *bytes = c.used(base) + 256;
    return g;
}
BinningState BinningState::carve(const void* base, size_t R, int end_bit, bool no_color, size_t* bytes)
{
    Carver c(base);
    BinningState g;
    g.plan = sort_plan(R, end_bit);  // the pass count only picks which ping-pong side holds the result: the layout depends on R alone
    for (int i = 0; i < 2; i++) {
        g.tile_keys[i] = c.take<uint32_t>(R);
        g.slots[i] = c.take<uint32_t>(R);
        g.gauss[i] = c.take<uint32_t>(R);
    }
    g.sort_scratch = c.take<char>(sort_scratch_bytes(g.plan));
    g.partials = no_color ? nullptr : c.take<float>((size_t)GS_PROW * R);
    g.dead = no_color ? nullptr : c.take<uint8_t>(R);
    // (36 R bytes of partial rows, unused until the backward: room for the four 4 R arrays — contiguous: they are also the 16 R bytes of binned())
    uint32_t* const ls = no_color ? c.take<uint32_t>(4 * R) : reinterpret_cast<uint32_t*>(g.partials);
    for (int i = 0; i < 4; i++) g.lsort[i] = ls + (size_t)i * R;
    if (bytes) *bytes = c.used(base) + 256;
    return g;
}
SampleState SampleState::carve(const void* base, size_t B, size_t* bytes)
{
    Carver c(base);
    SampleState g;
    g.bucket_to_tile = c.take<uint32_t>(B);
    g.ckpt = c.take<float4>(B * GS_TILE_PIX);
    g.hit = c.take<uint64_t>(B * GS_TILE_PIX);
    if (bytes) *bytes = c.used(base) + 256;
    return g;
}

// ---------------------------------------------------------------------------------------------------------
// Device -> host mailbox for the two counts the forward's host side needs (R to size the binning buffer, B to size the sample
// buffer: rasterizer_impl.cu:398,442 block on cudaMemcpy there).  A one-thread kernel stores {value pair, sequence number} into
// pinned, device-mapped host memory with system-scope release stores and the host spins on the sequence word: the GPU sits idle
// for the host's wake-up + the next launch, and a hipMemcpyAsync + hipStreamSynchronize pair costs ~40 us of that per count
// (measured from the kernel trace); the mailbox costs ~15.  Falls back to a stream synchronise if the word does not arrive.
struct Mailbox {
    volatile uint32_t* host = nullptr;  // [16]: value0, value1, seq, two extra words riding along, pad
    uint32_t* dev = nullptr;
    uint32_t seq = 0;
};
// One mailbox per (thread, device): the device pointer of mapped host memory belongs to the device it was resolved on
// (hipHostMallocPortable makes the allocation itself visible to every device).  Freed when the thread exits.
static constexpr int kMaxDevices = 16;
struct MailboxSet {
    Mailbox box[kMaxDevices];
    ~MailboxSet()
    {
        for (auto& m : box)
            if (m.host) (void)hipHostFree(const_cast<uint32_t*>(m.host));
    }
};
static thread_local MailboxSet t_mbox;
static uint32_t* g_status_word[kMaxDevices];
uint32_t* device_status_word()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    uint32_t* w = __atomic_load_n(&g_status_word[dev], __ATOMIC_ACQUIRE);
    if (w) return w;
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
    (void)hipMemset(p, 0, 256);
    uint32_t* expected = nullptr;
    if (!__atomic_compare_exchange_n(&g_status_word[dev], &expected, static_cast<uint32_t*>(p), false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
        (void)hipFree(p);  // another thread won
        return expected;
    }
    return static_cast<uint32_t*>(p);
}
__global__ void publish_kernel(const uint32_t* __restrict__ v0, const uint32_t* __restrict__ v1, const uint32_t* __restrict__ v2, uint32_t* status,
                               uint32_t* box, uint32_t seq)
{
    const uint32_t st = status ? atomicExch(status, 0u) : 0u;  // bit 0: a bounded spin-wait gave up somewhere upstream
    __hip_atomic_store(box + 0, v0 ? *v0 : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(box + 3, v2 ? v2[0] : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(box + 4, v2 ? v2[1] : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(box + 1, (v1 ? (*v1 & 1u) : 0u) | ((st & 3u) << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(box + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// out[0] = *v0, out[1] = (*v1 & 1) | scan-timeout << 1 | scan-overflow << 2; extra_out[0..1] = extra[0..1] (two optional words riding along)
static int fetch_counts(const uint32_t* v0, const uint32_t* v1, uint32_t out[2], hipStream_t s, uint32_t* fault = nullptr,
                        const uint32_t* extra = nullptr, uint32_t* extra_out = nullptr)
{
    if (extra_out) extra_out[0] = extra_out[1] = 0;
    int dev = 0;
    GS_HIP(hipGetDevice(&dev));
    static const bool use_mailbox = getenv("GSLIC_NO_MAILBOX") == nullptr;
    uint32_t* const status = fault ? fault : device_status_word();   // (a forward passes its own word: GS_FLAG_FAULT)
    if (use_mailbox && dev >= 0 && dev < kMaxDevices) {
        Mailbox& m = t_mbox.box[dev];
        if (!m.host) {
            void* h = nullptr;
            if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) == hipSuccess && h) {
                void* d = nullptr;
                if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
                    m.host = static_cast<volatile uint32_t*>(h);
                    m.dev = static_cast<uint32_t*>(d);
                    m.host[0] = m.host[1] = m.host[2] = 0;
                } else {
                    (void)hipHostFree(h);
                }
            }
            (void)hipGetLastError();
        }
        if (m.host) {
            const uint32_t seq = ++m.seq ? m.seq : ++m.seq;  // never 0
            hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, s, v0, v1, extra, status, m.dev, seq);
            GS_HIP(hipGetLastError());
            const auto t0 = std::chrono::steady_clock::now();
            uint32_t spins = 0;
            while (__atomic_load_n(const_cast<const uint32_t*>(m.host + 2), __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                    GS_HIP(hipStreamSynchronize(s));  // slow path: whatever is ahead of the publish kernel in the stream takes long
                    break;
                }
            }
            if (__atomic_load_n(const_cast<const uint32_t*>(m.host + 2), __ATOMIC_ACQUIRE) != seq)
                return set_error(GSLIC_ERR_HIP, "count mailbox: the publish kernel did not report");
            out[0] = m.host[0];
            out[1] = m.host[1];
            if (extra_out) { extra_out[0] = m.host[3]; extra_out[1] = m.host[4]; }
            return GSLIC_OK;
        }
    }
    uint32_t st = 0;
    out[0] = out[1] = 0;
    if (v0) GS_HIP(hipMemcpyAsync(&out[0], v0, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (v1) GS_HIP(hipMemcpyAsync(&out[1], v1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (status) GS_HIP(hipMemcpyAsync(&st, status, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (extra && extra_out) GS_HIP(hipMemcpyAsync(extra_out, extra, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GS_HIP(hipStreamSynchronize(s));
    if (st) GS_HIP(hipMemsetAsync(status, 0, sizeof(uint32_t), s));
    out[1] = (out[1] & 1u) | ((st & 3u) << 1);
    return GSLIC_OK;
}

static inline char* align256(char* p) { return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255)); }
static inline int tile_grid(int W, int H, int& gx, int& gy)
{
    gx = (W + GS_TILE - 1) / GS_TILE;
    gy = (H + GS_TILE - 1) / GS_TILE;
    return gx * gy;
}
static inline int sort_end_bit(int T) { return (int)higher_msb((uint32_t)T); }  // bits of a tile id (the depth half is sorted per Gaussian)
// GSLIC_SORT_ONESWEEP: bit 0 = depth sort, bit 1 = tile sort use the decoupled-look-back variant (same result bit for bit)
static inline int onesweep_mask() { static const int m = getenv("GSLIC_SORT_ONESWEEP") ? atoi(getenv("GSLIC_SORT_ONESWEEP")) : 0; return m; }

int adam_update(float*, const float*, float*, float*, const uint8_t*, float, float, float, float, uint32_t, uint32_t, hipStream_t);
int adam_update_groups(const gslic_adam_group*, int, const uint8_t*, float, float, float, uint32_t, hipStream_t);
int ssim_forward(int, int, int, int, float, float, const float*, const float*, float*, float*, float*, float*, hipStream_t);
int ssim_backward(int, int, int, int, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                  hipStream_t);
int knn_mean_dist2(int, const float*, float*, gslic_alloc_fn, void*, hipStream_t);
int loss_forward(int, int, int, int, float, float, const float*, const float*, float*, float*, float*, float*, float*, hipStream_t);
int loss_backward(int, int, int, int, float, const float*, const float*, const float*, const float*, const float*, float*, hipStream_t);
int loss_forward_backward(int, int, int, int, float, float, float, const float*, const float*, float*, float*, float*, float*, float*, float*, hipStream_t);
int64_t loss_partials_count(int, int, int, int);
int extend_select(int, const float*, const float*, const float*, const float*, float, float, float, float, int, int, const float*,
                  gslic_alloc_fn, void*, uint32_t**, uint32_t**, int32_t*, hipStream_t);
int extend_emit(int, const uint32_t*, const uint32_t*, const float*, const float*, const float*, float, float, int, float*, float*, float*,
                float*, float*, float*, hipStream_t);

static int check_params(const gslic_raster_params* p)
{
    if (!p) return set_error(GSLIC_ERR_INVALID_ARG, "params is NULL");
    if (p->P < 0 || p->width <= 0 || p->height <= 0) return set_error(GSLIC_ERR_INVALID_ARG, "bad sizes P=%d W=%d H=%d", p->P, p->width, p->height);
    if (p->D < 0 || p->D > 3) return set_error(GSLIC_ERR_INVALID_ARG, "SH degree %d out of range 0..3", p->D);
    if (p->M < 0 || (p->D > 0 && p->M < (p->D + 1) * (p->D + 1) - 1))
        return set_error(GSLIC_ERR_INVALID_ARG, "M=%d rest coefficients cannot hold SH degree %d", p->M, p->D);
    return GSLIC_OK;
}

#define DEBUG_SYNC(prm, s)                                                   \
    do {                                                                     \
        if ((prm)->debug) GS_HIP(hipStreamSynchronize(s));                   \
    } while (0)

}  // namespace gslic

using namespace gslic;

extern "C" {

int gslic_abi_version(void) { return GSLIC_ABI_VERSION; }
int gslic_set_math_mode(int32_t strict)
{
    const int old = g_strict_math;
    g_strict_math = strict != 0;
    return old;
}
const char* gslic_last_error(void) { return g_err; }

size_t gslic_geom_bytes(int32_t P) { size_t b; GeomState::carve(nullptr, (size_t)(P > 0 ? P : 0), &b); return b; }
size_t gslic_img_bytes(int32_t W, int32_t H)
{
    int gx, gy; size_t b;
    ImageState::carve(nullptr, (size_t)tile_grid(W, H, gx, gy), &b);
    return b;
}
size_t gslic_binning_bytes(int32_t R, int32_t no_color)
{
    size_t b;
    BinningState::carve(nullptr, (size_t)(R > 0 ? R : 0), 16, no_color != 0, &b);  // end_bit only changes the pass count
    return b;
}
size_t gslic_sample_bytes(int32_t B) { size_t b; SampleState::carve(nullptr, (size_t)(B > 0 ? B : 0), &b); return b; }

}  // extern "C"

namespace gslic {
// Capacity mode (gslic_rasterize_forward_capacity): the four buffers are the caller's, already sized; R and B never travel to the host.
struct ForwardCapacity {
    char *geom, *binning, *img, *sample;
    size_t geom_bytes, binning_bytes, img_bytes, sample_bytes;
    uint32_t* status_out;  // device [8]: see forward_status_kernel — written by the last kernel of the forward
};
// largest count whose carve fits into `bytes` (the carves are monotonic)
template <typename F>
static uint32_t capacity_for(size_t bytes, F bytes_needed)
{
    uint64_t lo = 0, hi = 0x7fffffffull;
    if (bytes_needed(0) > bytes) return 0;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (bytes_needed((uint32_t)mid) <= bytes) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}
// The last kernel of a capacity-mode forward.  out (device uint32[8], zeroed once by the caller):
//   [0] R, [1] B of this forward; [2] its bits: 1 instances did not fit, 2 buckets did not fit, 4 prefiltered violation, 8 a scan / sort
//   look-back wait timed out, 16 the instance count overflowed 2^31; [3] forwards that completed (no bit of 1 | 2 | 8 | 16);
//   [4] forwards issued; [5] bit (issue index mod 32) set for every forward that did NOT complete; [6] / [7] the largest R / B seen.
// The forward's own fault word (flags[GS_FLAG_FAULT]: timeout / overflow of ITS chained scans and sorts) is consumed here: the backward of a forward
// whose prefix sums cannot be trusted stands down through flags[2] like one that overflowed.  Per forward, so capacity-mode forwards may run
// concurrently on several streams of one device (round 4 kept these bits in one device-global word).
__global__ void forward_status_kernel(const uint32_t* __restrict__ R, const uint32_t* __restrict__ B, uint32_t* __restrict__ flags,
                                      uint32_t* __restrict__ dev_status, uint32_t* __restrict__ out)
{
    const uint32_t ds = dev_status ? atomicExch(dev_status, 0u) : 0u;
    uint32_t st = flags[2] | (flags[0] & 1u ? 4u : 0u) | (ds & 1u ? 8u : 0u) | (ds & 2u ? 16u : 0u);
    if (st & (8u | 16u)) flags[2] |= 4u;   // (any non-zero value makes the backward kernels of this step return at once)
    const uint32_t r = *R, b = B ? *B : 0u;
    out[0] = r;
    out[1] = b;
    out[2] = st;
    if (!(st & (3u | 8u | 16u))) out[3] += 1u;
    else out[5] |= 1u << (out[4] & 31u);
    out[4] += 1u;
    if (r > out[6]) out[6] = r;
    if (b > out[7]) out[7] = b;
}
}  // namespace gslic

// GSLIC_BINNING = auto (default) | atomic | radix.  auto: per host thread and per kind of map (rows in the caller's order / rows permuted by
// the library: tie_rank set), the path follows what the last binned forward measured — the global atomics its binning kernel needed per
// instance (0.08 with the rows of the 2M / 1080p scene in Morton order, 0.79 in random order; the two paths cost the same at about 0.5).  A map
// found incoherent is probed again every 64th forward.  Capacity-mode forwards read nothing back: they follow the last measurement of the
// thread, and before any, bin exactly when the library permuted the rows.
struct BinningAuto { int measured = 0; bool coherent = true; uint32_t since_probe = 0; uint32_t epoch = 0; };
static thread_local BinningAuto t_binning[2];
static thread_local int t_last_path = GSLIC_BINNING_PATH_NONE;       // gslic_get_binning_path
static thread_local uint32_t t_last_sample[2] = {0u, 0u};
// The mode is process-wide and may be set while other host threads run forwards: an atomic, and an epoch that makes every thread drop what it
// measured under the previous mode (round 5 reset the calling thread's state only).
static std::atomic<int> g_binning_mode{[] {
    const char* e = getenv("GSLIC_BINNING");
    if (e && strcmp(e, "radix") == 0) return 1;
    if (e && strcmp(e, "atomic") == 0) return 2;
    return 0;
}()};
static std::atomic<uint32_t> g_binning_epoch{0u};
static BinningAuto& binning_state(bool permuted)
{
    BinningAuto& st = t_binning[permuted ? 1 : 0];
    const uint32_t ep = g_binning_epoch.load(std::memory_order_acquire);
    if (st.epoch != ep) { st = BinningAuto(); st.epoch = ep; }
    return st;
}
// 1 = block-aggregated atomics, 0 = the radix sort, < 0 = an error code (the atomic path was forced and the device cannot run it)
static int binning_choice(int T, bool permuted, bool capacity)
{
    const int mode = g_binning_mode.load(std::memory_order_relaxed);
    if (T > gslic::GS_TILE_BIN_MAX_T || mode == 1) return 0;
    if (!gslic::tile_bin_lds_ok(T)) {   // more than 64 KB of dynamic LDS asked for and refused: same lists from the sort
        if (mode == 2) return set_error(GSLIC_ERR_HIP, "GSLIC_BINNING=atomic: the device does not grant the %d-tile histogram's dynamic LDS", T);
        return 0;
    }
    if (mode == 2) return 1;
    BinningAuto& st = binning_state(permuted);
    if (!st.measured) return (capacity ? permuted : true) ? 1 : 0;
    if (st.coherent) return 1;
    if (capacity) return 0;
    return ++st.since_probe >= 64u ? 1 : 0;   // (a probe: binning_feedback resets the count)
}
static void binning_feedback(bool permuted, bool used_bin, uint32_t atomics, uint32_t instances)
{
    t_last_sample[0] = used_bin ? atomics : 0u;
    t_last_sample[1] = used_bin ? instances : 0u;
    if (!used_bin || instances < 4096u) return;
    BinningAuto& st = binning_state(permuted);
    st.measured = 1;
    st.coherent = (double)atomics <= 0.4 * (double)instances;
    st.since_probe = 0;
}

static int rasterize_forward_impl(const gslic_raster_params* prm, gslic_alloc_fn geom_alloc, void* geom_ctx, gslic_alloc_fn binning_alloc,
                            void* binning_ctx, gslic_alloc_fn img_alloc, void* img_ctx, gslic_alloc_fn sample_alloc,
                            void* sample_ctx, const ForwardCapacity* cap, const float* background, const float* means3D, const float* dc, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                            float* out_color, float* out_final_T, int32_t* radii, int32_t* num_rendered, int32_t* num_buckets,
                            void* stream)
{
    (void)background;
    GS_TRY(check_params(prm));
    if (!num_rendered || !num_buckets) return set_error(GSLIC_ERR_INVALID_ARG, "num_rendered / num_buckets is NULL");
    *num_rendered = 0;
    *num_buckets = 0;
    const int P = prm->P;
    if (P == 0) return GSLIC_OK;  // rasterize_points.cu:110
    if (colors_precomp || cov3D_precomp)
        return set_error(GSLIC_ERR_UNSUPPORTED, "colors_precomp / cov3D_precomp are not supported (the reference host always passes empty tensors)");
    if (!cap && (!geom_alloc || !binning_alloc || !img_alloc || (!prm->no_color && !sample_alloc)))
        return set_error(GSLIC_ERR_INVALID_ARG, "allocator callback is NULL");
    if (!means3D || !dc || !opacities || !scales || !rotations || !viewmatrix || !projmatrix || !cam_pos || !out_final_T || !radii ||
        (!prm->no_color && !out_color) || (prm->M > 0 && !shs))
        return set_error(GSLIC_ERR_INVALID_ARG, "required tensor pointer is NULL");
    hipStream_t s = (hipStream_t)stream;
    const bool no_color = prm->no_color != 0;
    int gx, gy;
    const int T = tile_grid(prm->width, prm->height, gx, gy);

    size_t geom_bytes, img_bytes;
    GeomState::carve(nullptr, (size_t)P, &geom_bytes);
    ImageState::carve(nullptr, (size_t)T, &img_bytes);
    if (cap && (cap->geom_bytes < geom_bytes || cap->img_bytes < img_bytes || !cap->geom || !cap->img || !cap->binning ||
                (!no_color && !cap->sample) || !cap->status_out))
        return set_error(GSLIC_ERR_INVALID_ARG, "capacity mode: geometry / image buffer too small (need %zu / %zu bytes) or a NULL buffer", geom_bytes, img_bytes);
    char* geom_base = cap ? cap->geom : geom_alloc(geom_ctx, geom_bytes);
    if (!geom_base) return set_error(GSLIC_ERR_ALLOC, "geometry allocator returned NULL for %zu bytes", geom_bytes);
    GeomState geom = GeomState::carve(align256(geom_base), (size_t)P, nullptr);
    char* img_base = cap ? cap->img : img_alloc(img_ctx, img_bytes);
    if (!img_base) return set_error(GSLIC_ERR_ALLOC, "image allocator returned NULL for %zu bytes", img_bytes);
    ImageState img = ImageState::carve(align256(img_base), (size_t)T, nullptr);

    GS_HIP(hipMemsetAsync(geom.flags, 0, geom.zero_bytes, s));
    PreprocessArgs pa;
    pa.P = P; pa.D = prm->D; pa.M = prm->M; pa.W = prm->width; pa.H = prm->height; pa.gx = gx; pa.gy = gy;
    pa.focal_y = prm->height / (2.0f * prm->tan_fovy);  // rasterizer_impl.cu:348-349
    pa.focal_x = prm->width / (2.0f * prm->tan_fovx);
    pa.limx_neg = prm->limx_neg; pa.limx_pos = prm->limx_pos; pa.limy_neg = prm->limy_neg; pa.limy_pos = prm->limy_pos;
    pa.scale_modifier = prm->scale_modifier; pa.prefiltered = prm->prefiltered; pa.no_color = prm->no_color; pa.raw = prm->raw_params;
    pa.means = means3D; pa.scales = scales; pa.rots = rotations; pa.opac = opacities; pa.dc = dc; pa.shs = shs;
    pa.view = viewmatrix; pa.proj = projmatrix; pa.campos = cam_pos;
    pa.radii = radii; pa.rec = geom.rec; pa.tiles_touched = geom.tiles_touched; pa.depth_keys = nullptr; pa.flags = geom.flags; pa.ranges = img.ranges;
    GS_TRY(launch_preprocess(pa, s));
    DEBUG_SYNC(prm, s);

    // Emission slots in INDEX order, as the reference hands them out (rasterizer_impl.cu:395): one plain scan.  (Rounds 2-4 sorted the Gaussians
    // by depth first and emitted in that order; the per-tile depth sort below replaced that: DESIGN.md section 3.)
    GS_TRY(scan_u32_chained(geom.tiles_touched, nullptr, geom.point_offsets, (size_t)P, false, geom.scan_state, s, geom.flags + GS_FLAG_FAULT));
    uint32_t hostbuf[2] = {0, 0};
    const int end_bit = sort_end_bit(T);
    const uint32_t* const R_dev = cap ? geom.point_offsets + (P - 1) : nullptr;  // capacity mode: the count stays on the device
    uint32_t R;
    if (cap) {
        // R = the number of instances the caller's binning buffer holds; the kernels stop at the real count (R_dev) and raise
        // status bit 0 when it does not fit.  No host read (the reference blocks here, rasterizer_impl.cu:398).
        R = capacity_for(cap->binning_bytes, [&](uint32_t r) { size_t b; BinningState::carve(nullptr, (size_t)r, end_bit, no_color, &b); return b; });
        if (R == 0) return set_error(GSLIC_ERR_INVALID_ARG, "capacity mode: the binning buffer (%zu bytes) does not hold a single instance", cap->binning_bytes);
    } else {
        GS_TRY(fetch_counts(geom.point_offsets + (P - 1), geom.flags, hostbuf, s, geom.flags + GS_FLAG_FAULT));  // host needs R to size the binning buffer (rasterizer_impl.cu:398)
        if (hostbuf[1] & 2u) return set_error(GSLIC_ERR_HIP, "a chained-scan look-back wait timed out (device preempted?): the forward was abandoned");
        if (hostbuf[1] & 4u) return set_error(GSLIC_ERR_INVALID_ARG, "more than 2^31 (Gaussian, tile) instances");
        if (prm->prefiltered && (hostbuf[1] & 1u)) return set_error(GSLIC_ERR_PREFILTERED, "a point was culled although prefiltered is set");
        if (hostbuf[0] > 0x7fffffffu) return set_error(GSLIC_ERR_INVALID_ARG, "more than 2^31 (Gaussian, tile) instances");
        R = hostbuf[0];
    }

    size_t bin_bytes;
    BinningState::carve(nullptr, (size_t)R, end_bit, no_color, &bin_bytes);
    char* bin_base = cap ? cap->binning : binning_alloc(binning_ctx, bin_bytes);
    if (!bin_base) return set_error(GSLIC_ERR_ALLOC, "binning allocator returned NULL for %zu bytes", bin_bytes);
    BinningState bin = BinningState::carve(align256(bin_base), (size_t)R, end_bit, no_color, nullptr);

    // How the instances get grouped by tile (same lists either way, bit for bit): block-aggregated atomics on the tiles' cursors (tile_bin.hip) when
    // the map's row order keeps a block's instances on few tiles, the stable radix sort on the tile id otherwise (and above GS_TILE_BIN_MAX_T tiles).
    const int bin_choice = R > 0 ? binning_choice(T, prm->tie_rank != nullptr, cap != nullptr) : 0;
    if (bin_choice < 0) return bin_choice;
    const bool use_bin = bin_choice == 1;
    if (R > 0) {
        t_last_path = use_bin ? GSLIC_BINNING_PATH_ATOMIC : GSLIC_BINNING_PATH_RADIX;
        t_last_sample[0] = t_last_sample[1] = 0u;
        const int pp = bin.plan.passes & 1;   // the lists go to gauss[pp ^ 1] / slots[pp ^ 1] on either path (BinningState::point_list())
        KeybuildArgs ka;
        ka.P = P; ka.gx = gx; ka.gy = gy; ka.rec = geom.rec; ka.order = nullptr; ka.offsets = geom.point_offsets;
        ka.gauss_start = geom.gauss_start; ka.cap = R; ka.status = geom.flags;
        ka.tile_keys = bin.tile_keys[0];
        ka.gauss = use_bin ? bin.gauss[pp] : bin.gauss[0];
        ka.depth = use_bin ? bin.slots[pp] : bin.lsort[0];   // (binned() is the lsort arrays: the emission-order depths must not sit there)
        GS_TRY(launch_keybuild(ka, s));
        DEBUG_SYNC(prm, s);
        TileDepthSortArgs ts;
        ts.T = T; ts.ranges = img.ranges; ts.tie_rank = prm->tie_rank; ts.status = geom.flags; ts.long_tiles = img.bucket_offsets;
        ts.gauss_out = bin.gauss[pp ^ 1]; ts.slot_out = bin.slots[pp ^ 1];
        if (use_bin) {
            TileBinArgs tb;
            tb.T = T; tb.n_cap = R; tb.n_dev = R_dev; tb.tile = ka.tile_keys; tb.gid = ka.gauss; tb.depth = ka.depth; tb.tie_rank = prm->tie_rank;
            tb.ranges = img.ranges; tb.binned = bin.binned(); tb.dead = bin.dead; tb.status = geom.flags;
            tb.bucket_offsets = no_color ? nullptr : img.bucket_offsets; tb.max_contrib = img.max_contrib;   // (the bucket scan rides on the tile scan)
            ts.long_tiles = reinterpret_cast<uint32_t*>(bin.sort_scratch);   // (the radix sort's scratch, unused on this path: at least R / 4 bytes)
            GS_TRY(launch_tile_bin(tb, s));
            DEBUG_SYNC(prm, s);
            ts.binned = bin.binned(); ts.depth = nullptr; ts.gauss_in = nullptr; ts.slot_in = nullptr;
            ts.key_a = bin.tile_keys[0]; ts.key_b = bin.tile_keys[1]; ts.idx_a = bin.gauss[pp]; ts.idx_b = bin.slots[pp];   // (all dead by now)
        } else {
            // Level 1: stable sort of the instances on the tile id alone (ceil(log2(tiles)/8) digit passes, 2 at 1080p), carrying the emission
            // slot (identity at the start), the Gaussian id and the depth bits: instances grouped by tile, in index order inside a tile.
            SortBuffers sb;
            for (int i = 0; i < 2; i++) { sb.keys[i] = bin.tile_keys[i]; sb.v0[i] = bin.slots[i]; sb.v1[i] = bin.gauss[i]; sb.v2[i] = bin.lsort[i]; }
            sb.v0_identity = true;
            GS_TRY(radix_sort_u32(sb, bin.plan, bin.sort_scratch, (onesweep_mask() & 2) != 0, K_SORT_HIST, K_SORT_SCATTER, s, R_dev, geom.flags + GS_FLAG_FAULT));
            DEBUG_SYNC(prm, s);
            GS_TRY(launch_finalize_ranges(R, R_dev, bin.tile_keys[pp], img.ranges, bin.dead, s));
            DEBUG_SYNC(prm, s);
            ts.binned = nullptr; ts.depth = bin.lsort[pp]; ts.gauss_in = bin.gauss[pp]; ts.slot_in = bin.slots[pp];
            ts.key_a = bin.lsort[pp ^ 1]; ts.key_b = bin.tile_keys[pp ^ 1]; ts.idx_a = bin.lsort[2]; ts.idx_b = bin.lsort[3];
        }
        // Level 2: every tile's segment by (depth, original index) — the order of the reference's 64-bit sort of an index-ordered emission
        GS_TRY(launch_tile_depth_sort(ts, s));
        DEBUG_SYNC(prm, s);
    }

    SampleState smp;
    memset(&smp, 0, sizeof(smp));
    uint32_t B = 0;
    if (!no_color) {
        if (!use_bin) GS_TRY(launch_bucket_scan(T, img.ranges, img.bucket_offsets, img.max_contrib, s));
        if (cap) {
            B = capacity_for(cap->sample_bytes, [&](uint32_t b) { size_t n; SampleState::carve(nullptr, (size_t)b, &n); return n; });
        } else {
            uint32_t bin_sample[2] = {0, 0};   // {global atomics, instances} of the binning kernel's sampled workgroups
            GS_TRY(fetch_counts(img.bucket_offsets + (T - 1), nullptr, hostbuf, s, geom.flags + GS_FLAG_FAULT,
                                use_bin ? geom.flags + GS_FLAG_BIN_ATOMICS : nullptr, bin_sample));  // rasterizer_impl.cu:442
            if (hostbuf[1] & 2u) return set_error(GSLIC_ERR_HIP, "a sort look-back / chained-scan wait timed out (device preempted?): the forward was abandoned");
            B = hostbuf[0];
            binning_feedback(prm->tie_rank != nullptr, use_bin, bin_sample[0], bin_sample[1]);
        }
        size_t smp_bytes;
        SampleState::carve(nullptr, (size_t)B, &smp_bytes);
        char* smp_base = cap ? cap->sample : sample_alloc(sample_ctx, smp_bytes);
        if (!smp_base) return set_error(GSLIC_ERR_ALLOC, "sample allocator returned NULL for %zu bytes", smp_bytes);
        smp = SampleState::carve(align256(smp_base), (size_t)B, nullptr);
    }

    RenderFwdArgs ra;
    ra.W = prm->width; ra.H = prm->height; ra.gx = gx; ra.gy = gy; ra.no_color = prm->no_color;
    ra.ranges = img.ranges; ra.point_list = bin.point_list(); ra.rec = geom.rec; ra.bucket_offsets = img.bucket_offsets;
    ra.bucket_to_tile = smp.bucket_to_tile; ra.ckpt = smp.ckpt; ra.hit = smp.hit; ra.pix_final = img.pix_final; ra.max_contrib = img.max_contrib;
    ra.out_color = out_color; ra.out_final_T = out_final_T; ra.capB = B; ra.status = geom.flags; ra.tail4_from = T;   // (launch_render_fwd decides)
    GS_TRY(launch_render_fwd(ra, s));
    DEBUG_SYNC(prm, s);
    if (cap) {
        hipLaunchKernelGGL(forward_status_kernel, dim3(1), dim3(1), 0, s, (const uint32_t*)(geom.point_offsets + (P - 1)),
                           no_color ? (const uint32_t*)nullptr : (const uint32_t*)(img.bucket_offsets + (T - 1)), geom.flags, geom.flags + GS_FLAG_FAULT,
                           cap->status_out);
        GS_HIP(hipGetLastError());
    }

    *num_rendered = (int32_t)R;
    *num_buckets = (int32_t)B;
    return GSLIC_OK;
}

extern "C" {
int gslic_set_binning_mode(int32_t mode)
{
    if (mode < 0 || mode > 2) return g_binning_mode.load(std::memory_order_relaxed);
    const int old = g_binning_mode.exchange(mode, std::memory_order_relaxed);
    g_binning_epoch.fetch_add(1u, std::memory_order_release);   // every thread's BinningAuto starts over (binning_state)
    return old;
}
int gslic_get_binning_path(uint32_t* sampled_atomics, uint32_t* sampled_instances)
{
    if (sampled_atomics) *sampled_atomics = t_last_sample[0];
    if (sampled_instances) *sampled_instances = t_last_sample[1];
    return t_last_path;
}
size_t gslic_scratch_round_up(size_t n)
{
    if (n <= (size_t(1) << 20)) return n;
    size_t g = size_t(32) << 20;
    if (n > (size_t(64) << 20)) {
        size_t p = 1;
        while ((p << 1) <= n && (p << 1) != 0) p <<= 1;
        const size_t half = p >> 1, cap = size_t(256) << 20;
        g = half > cap ? cap : (half > g ? half : g);
    }
    return (n + g - 1) / g * g;
}
int gslic_rasterize_forward(const gslic_raster_params* prm, gslic_alloc_fn geom_alloc, void* geom_ctx, gslic_alloc_fn binning_alloc,
                            void* binning_ctx, gslic_alloc_fn img_alloc, void* img_ctx, gslic_alloc_fn sample_alloc,
                            void* sample_ctx, const float* background, const float* means3D, const float* dc, const float* shs,
                            const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                            const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                            float* out_color, float* out_final_T, int32_t* radii, int32_t* num_rendered, int32_t* num_buckets,
                            void* stream)
{
    return rasterize_forward_impl(prm, geom_alloc, geom_ctx, binning_alloc, binning_ctx, img_alloc, img_ctx, sample_alloc, sample_ctx, nullptr,
                                  background, means3D, dc, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                                  cam_pos, out_color, out_final_T, radii, num_rendered, num_buckets, stream);
}

int gslic_rasterize_forward_capacity(const gslic_raster_params* prm, char* geom_buffer, size_t geom_bytes, char* binning_buffer,
                                     size_t binning_bytes, char* img_buffer, size_t img_bytes, char* sample_buffer, size_t sample_bytes,
                                     const float* background, const float* means3D, const float* dc, const float* shs,
                                     const float* colors_precomp, const float* opacities, const float* scales, const float* rotations,
                                     const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                     float* out_color, float* out_final_T, int32_t* radii, int32_t* capacity_R, int32_t* capacity_B,
                                     uint32_t* status, void* stream)
{
    ForwardCapacity cap{geom_buffer, binning_buffer, img_buffer, sample_buffer, geom_bytes, binning_bytes, img_bytes, sample_bytes, status};
    return rasterize_forward_impl(prm, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &cap, background, means3D, dc, shs,
                                  colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, out_color,
                                  out_final_T, radii, capacity_R, capacity_B, stream);
}

static int rasterize_backward_impl(const gslic_raster_params* prm, int32_t R, int32_t B, const float* background, const float* means3D,
                             const float* dc, const float* shs, const float* colors_precomp, const float* scales,
                             const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                             const float* cam_pos, const int32_t* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                             char* sample_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                             float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_ddc, float* dL_dsh, float* dL_dscale,
                             float* dL_drot, float lambda_erank, const gslic_adam_fused* adam, float* const dL_dcam[3], void* stream,
                             float* dL_drgb = nullptr, int32_t row_begin = 0, int32_t row_end = -1, bool skip_blend = false, uint8_t* vis_out = nullptr,
                             float* campos_out = nullptr)
{
    (void)background; (void)dc;
    GS_TRY(check_params(prm));
    const int P = prm->P;
    if (P == 0) return GSLIC_OK;  // rasterize_points.cu:203
    if (colors_precomp || cov3D_precomp)
        return set_error(GSLIC_ERR_UNSUPPORTED, "colors_precomp / cov3D_precomp are not supported");
    if (prm->no_color) return set_error(GSLIC_ERR_INVALID_ARG, "backward of a no_color forward is undefined (no checkpoints were stored)");
    if (R < 0 || B < 0) return set_error(GSLIC_ERR_INVALID_ARG, "negative R / B");
    if (!means3D || !scales || !rotations || !viewmatrix || !projmatrix || !cam_pos || !radii || !geom_buffer || !binning_buffer ||
        !img_buffer || !sample_buffer || !dL_dpix || (prm->M > 0 && !shs))
        return set_error(GSLIC_ERR_INVALID_ARG, "required tensor pointer is NULL");
    if (!adam && !dL_drgb && (!dL_dopacity || !dL_dmean3D || !dL_ddc || !dL_dscale || !dL_drot || (prm->M > 0 && !dL_dsh)))
        return set_error(GSLIC_ERR_INVALID_ARG, "required gradient output pointer is NULL");
    if (dL_drgb && (adam || dL_ddc || dL_dsh || !dL_dopacity || !dL_dmean3D || !dL_dscale || !dL_drot))
        return set_error(GSLIC_ERR_INVALID_ARG, "dL_drgb mode: dL_ddc / dL_dsh / adam must be NULL, the four other parameter gradients are required");
    if (adam) {
        if (!prm->raw_params) return set_error(GSLIC_ERR_INVALID_ARG, "fused Adam needs raw_params = 1 (it updates the raw parameters)");
        for (int g = 0; g < 6; g++) {
            if (g == 2 && prm->M == 0) continue;
            if (!adam->param[g] || !adam->exp_avg[g] || !adam->exp_avg_sq[g]) return set_error(GSLIC_ERR_INVALID_ARG, "fused Adam: group %d has a NULL pointer", g);
        }
        if (adam->param[0] != means3D || adam->param[4] != scales || adam->param[5] != rotations || (prm->M > 0 && adam->param[2] != shs))
            return set_error(GSLIC_ERR_INVALID_ARG, "fused Adam: param[] must alias the tensors passed as means3D / shs / scales / rotations");
    }
    {   // (checked before anything is enqueued: a bad row range must not leave a blend backward behind)
        const int re = row_end < 0 ? P : row_end;
        if (row_begin < 0 || re > P || row_begin > re || (row_begin & 63))
            return set_error(GSLIC_ERR_INVALID_ARG, "row range [%d, %d) of %d Gaussians (row_begin must be a multiple of 64)", row_begin, re, P);
    }
    hipStream_t s = (hipStream_t)stream;
    int gx, gy;
    const int T = tile_grid(prm->width, prm->height, gx, gy);
    GeomState geom = GeomState::carve(align256(geom_buffer), (size_t)P, nullptr);
    ImageState img = ImageState::carve(align256(img_buffer), (size_t)T, nullptr);
    BinningState bin = BinningState::carve(align256(binning_buffer), (size_t)R, sort_end_bit(T), false, nullptr);
    SampleState smp = SampleState::carve(align256(sample_buffer), (size_t)B, nullptr);

    RenderBwdArgs rb;
    rb.W = prm->width; rb.H = prm->height; rb.gx = gx; rb.B = B;
    rb.ranges = img.ranges; rb.point_list = bin.point_list(); rb.inst_slot = bin.inst_slot(); rb.rec = geom.rec;
    rb.bucket_offsets = img.bucket_offsets; rb.bucket_to_tile = smp.bucket_to_tile; rb.ckpt = smp.ckpt; rb.hit = smp.hit; rb.pix_final = img.pix_final;
    rb.max_contrib = img.max_contrib; rb.dL_dpix = dL_dpix; rb.partials = bin.partials; rb.dead = bin.dead; rb.status = geom.flags; rb.T = T;
    if (!skip_blend) GS_TRY(launch_render_bwd(rb, s));   // (a chunked per-Gaussian backward runs the blend backward with its first chunk only)
    DEBUG_SYNC(prm, s);

    PreprocessBwdArgs pb;
    pb.P = P; pb.D = prm->D; pb.M = prm->M; pb.W = prm->width; pb.H = prm->height; pb.raw = prm->raw_params;
    pb.row_begin = row_begin; pb.row_end = row_end < 0 ? P : row_end;
    pb.focal_y = prm->height / (2.0f * prm->tan_fovy);
    pb.focal_x = prm->width / (2.0f * prm->tan_fovx);
    pb.limx_neg = prm->limx_neg; pb.limx_pos = prm->limx_pos; pb.limy_neg = prm->limy_neg; pb.limy_pos = prm->limy_pos;
    pb.scale_modifier = prm->scale_modifier; pb.lambda_erank = lambda_erank;
    pb.means = means3D; pb.scales = scales; pb.rots = rotations; pb.dc = dc; pb.shs = shs; pb.view = viewmatrix; pb.proj = projmatrix;
    pb.campos = cam_pos; pb.radii = radii; pb.rec = geom.rec; pb.tiles_touched = geom.tiles_touched; pb.gauss_start = geom.gauss_start; pb.partials = bin.partials; pb.dead = bin.dead;
    pb.dL_dmean2D = dL_dmean2D; pb.dL_dconic = dL_dconic; pb.dL_dopacity = dL_dopacity; pb.dL_dcolor = dL_dcolor;
    pb.dL_dmean3D = dL_dmean3D; pb.dL_dcov3D = dL_dcov3D; pb.dL_ddc = dL_ddc; pb.dL_dsh = dL_dsh; pb.dL_dscale = dL_dscale;
    pb.dL_drot = dL_drot; pb.dL_drgb = dL_drgb;
    memset(&pb.adam, 0, sizeof(pb.adam));
    if (adam) {
        for (int g = 0; g < 6; g++) { pb.adam.p[g] = adam->param[g]; pb.adam.m[g] = adam->exp_avg[g]; pb.adam.v[g] = adam->exp_avg_sq[g]; pb.adam.lr[g] = adam->lr[g]; }
        pb.adam.b1 = adam->b1; pb.adam.b2 = adam->b2; pb.adam.eps = adam->eps; pb.adam.on = 1;
    }
    pb.status = geom.flags;
    pb.cam_partials = nullptr; pb.cam_out = nullptr;
    pb.vis_out = vis_out; pb.campos_out = campos_out;
    if (dL_dcam) {
        // scratch of the geometry buffer: the per-wave partial rows (128 B per wave; the generic 256-thread path writes 4 rows per block even
        // when P < 256, i.e. up to 4 ceil(P / 256) rows), the 35 reduced terms behind them
        pb.cam_partials = geom.cam_scratch;
        pb.cam_out = geom.cam_scratch + 128 * (((size_t)P + 255) / 256);
    }
    GS_TRY(launch_preprocess_bwd(pb, s));
    if (dL_dcam) {
        GS_HIP(hipMemcpyAsync(dL_dcam[0], pb.cam_out, 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
        GS_HIP(hipMemcpyAsync(dL_dcam[1], pb.cam_out + 16, 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
        GS_HIP(hipMemcpyAsync(dL_dcam[2], pb.cam_out + 32, 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    DEBUG_SYNC(prm, s);
    return GSLIC_OK;
}

int gslic_rasterize_backward(const gslic_raster_params* prm, int32_t R, int32_t B, const float* background, const float* means3D,
                             const float* dc, const float* shs, const float* colors_precomp, const float* scales,
                             const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                             const float* cam_pos, const int32_t* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                             char* sample_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                             float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_ddc, float* dL_dsh, float* dL_dscale,
                             float* dL_drot, float lambda_erank, void* stream)
{
    return rasterize_backward_impl(prm, R, B, background, means3D, dc, shs, colors_precomp, scales, rotations, cov3D_precomp, viewmatrix,
                                   projmatrix, cam_pos, radii, geom_buffer, binning_buffer, img_buffer, sample_buffer, dL_dpix, dL_dmean2D,
                                   dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_ddc, dL_dsh, dL_dscale, dL_drot,
                                   lambda_erank, nullptr, nullptr, stream);
}

int gslic_rasterize_backward_rgb(const gslic_raster_params* prm, int32_t R, int32_t B, const float* background, const float* means3D,
                                 const float* dc, const float* shs, const float* colors_precomp, const float* scales,
                                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                 const float* cam_pos, const int32_t* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                                 char* sample_buffer, const float* dL_dpix, float* dL_dopacity, float* dL_dmean3D, float* dL_drgb,
                                 float* dL_dscale, float* dL_drot, float lambda_erank, void* stream)
{
    if (!dL_drgb) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_rasterize_backward_rgb: dL_drgb is NULL");
    return rasterize_backward_impl(prm, R, B, background, means3D, dc, shs, colors_precomp, scales, rotations, cov3D_precomp, viewmatrix,
                                   projmatrix, cam_pos, radii, geom_buffer, binning_buffer, img_buffer, sample_buffer, dL_dpix, nullptr,
                                   nullptr, dL_dopacity, nullptr, dL_dmean3D, nullptr, nullptr, nullptr, dL_dscale, dL_drot, lambda_erank,
                                   nullptr, nullptr, stream, dL_drgb);
}

int gslic_rasterize_backward_rgb_payload(const gslic_raster_params* prm, int32_t R, int32_t B, const float* background, const float* means3D,
                                         const float* dc, const float* shs, const float* colors_precomp, const float* scales,
                                         const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                         const float* cam_pos, const int32_t* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                                         char* sample_buffer, const float* dL_dpix, float* dL_dopacity, float* dL_dmean3D, float* dL_drgb,
                                         float* dL_dscale, float* dL_drot, float lambda_erank, uint8_t* vis_out, float* campos_out, void* stream)
{
    if (!dL_drgb) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_rasterize_backward_rgb_payload: dL_drgb is NULL");
    return rasterize_backward_impl(prm, R, B, background, means3D, dc, shs, colors_precomp, scales, rotations, cov3D_precomp, viewmatrix,
                                   projmatrix, cam_pos, radii, geom_buffer, binning_buffer, img_buffer, sample_buffer, dL_dpix, nullptr,
                                   nullptr, dL_dopacity, nullptr, dL_dmean3D, nullptr, nullptr, nullptr, dL_dscale, dL_drot, lambda_erank,
                                   nullptr, nullptr, stream, dL_drgb, 0, -1, false, vis_out, campos_out);
}

int gslic_rasterize_backward_rgb_rows(const gslic_raster_params* prm, int32_t R, int32_t B, const float* background, const float* means3D,
                                      const float* dc, const float* shs, const float* colors_precomp, const float* scales,
                                      const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                      const float* cam_pos, const int32_t* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                                      char* sample_buffer, const float* dL_dpix, float* dL_dopacity, float* dL_dmean3D, float* dL_drgb,
                                      float* dL_dscale, float* dL_drot, float lambda_erank, int32_t row_begin, int32_t row_end, int32_t skip_blend,
                                      void* stream)
{
    if (!dL_drgb) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_rasterize_backward_rgb_rows: dL_drgb is NULL");
    return rasterize_backward_impl(prm, R, B, background, means3D, dc, shs, colors_precomp, scales, rotations, cov3D_precomp, viewmatrix,
                                   projmatrix, cam_pos, radii, geom_buffer, binning_buffer, img_buffer, sample_buffer, dL_dpix, nullptr,
                                   nullptr, dL_dopacity, nullptr, dL_dmean3D, nullptr, nullptr, nullptr, dL_dscale, dL_drot, lambda_erank,
                                   nullptr, nullptr, stream, dL_drgb, row_begin, row_end, skip_blend != 0);
}

int gslic_sh_grad_from_rgb(int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D, const float* campos_all,
                           const float* rgb_all, int32_t input_is_ddc, float* dL_ddc, float* dL_dsh, int64_t view_stride, void* stream)
{
    if (view_stride < 0 || (view_stride > 0 && view_stride < 3 * (int64_t)P)) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb: bad view_stride");
    if (P < 0 || D < 0 || D > 3 || M < 0 || n_views < 1) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb: bad P / D / M / n_views");
    if (P == 0) return GSLIC_OK;
    if (!means3D || !campos_all || !rgb_all || !dL_ddc || (M > 0 && !dL_dsh)) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb: NULL pointer");
    ShGradFromRgbArgs a;
    a.P = P; a.D = D; a.M = M; a.n_views = n_views; a.input_is_ddc = input_is_ddc ? 1 : 0;
    a.means3D = means3D; a.campos_all = campos_all; a.rgb_all = rgb_all; a.dL_ddc = dL_ddc; a.dL_dsh = dL_dsh;
    a.rgb_stride = view_stride ? (size_t)view_stride : (size_t)3 * (size_t)P; a.campos_stride = view_stride ? (size_t)view_stride : 3;
    a.visible = nullptr; a.vis_stride = 0; a.vis_out = nullptr;
    for (auto& g : a.g_small) g = nullptr;
    memset(&a.adam, 0, sizeof(a.adam));
    return launch_sh_grad_from_rgb(a, (hipStream_t)stream);
}

int gslic_sh_grad_from_rgb_adam(int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D, const float* campos_all,
                                const float* rgb_all, int32_t input_is_ddc, const uint8_t* visible, const gslic_adam_fused* adam, float* dL_ddc,
                                float* dL_dsh, int64_t view_stride, void* stream)
{
    if (view_stride < 0 || (view_stride > 0 && view_stride < 3 * (int64_t)P)) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam: bad view_stride");
    if (P < 0 || D < 0 || D > 3 || M < 0 || n_views < 1) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam: bad P / D / M / n_views");
    if (P == 0) return GSLIC_OK;
    if (!means3D || !campos_all || !rgb_all || !visible || !adam) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam: NULL pointer");
    for (int g = 1; g <= 2; g++) {
        if (g == 2 && M == 0) continue;
        if (!adam->param[g] || !adam->exp_avg[g] || !adam->exp_avg_sq[g]) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam: group %d has a NULL pointer", g);
    }
    ShGradFromRgbArgs a;
    a.P = P; a.D = D; a.M = M; a.n_views = n_views; a.input_is_ddc = input_is_ddc ? 1 : 0;
    a.means3D = means3D; a.campos_all = campos_all; a.rgb_all = rgb_all; a.dL_ddc = dL_ddc; a.dL_dsh = M > 0 ? dL_dsh : nullptr;
    a.rgb_stride = view_stride ? (size_t)view_stride : (size_t)3 * (size_t)P; a.campos_stride = view_stride ? (size_t)view_stride : 3;
    a.visible = visible; a.vis_stride = 0; a.vis_out = nullptr;
    for (auto& g : a.g_small) g = nullptr;
    memset(&a.adam, 0, sizeof(a.adam));
    for (int g = 1; g <= 2; g++) { a.adam.p[g] = adam->param[g]; a.adam.m[g] = adam->exp_avg[g]; a.adam.v[g] = adam->exp_avg_sq[g]; a.adam.lr[g] = adam->lr[g]; }
    a.adam.b1 = adam->b1; a.adam.b2 = adam->b2; a.adam.eps = adam->eps; a.adam.on = 1;
    return launch_sh_grad_from_rgb(a, (hipStream_t)stream);
}

int gslic_sh_grad_from_rgb_adam_all(int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D, const float* campos_all,
                                    const float* rgb_all, int32_t input_is_ddc, const uint8_t* vis_all, int64_t vis_stride, uint8_t* vis_out,
                                    const gslic_adam_fused* adam, const float* dL_dmean3D, const float* dL_dopacity, const float* dL_dscale,
                                    const float* dL_drot, int64_t view_stride, void* stream)
{
    if (view_stride < 0 || (view_stride > 0 && view_stride < 3 * (int64_t)P)) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam_all: bad view_stride");
    if (P < 0 || D < 0 || D > 3 || M < 0 || n_views < 1 || vis_stride < 0 || (vis_stride > 0 && vis_stride < (int64_t)P))
        return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam_all: bad P / D / M / n_views / vis_stride");
    if (P == 0) return GSLIC_OK;
    if (!means3D || !campos_all || !rgb_all || !vis_all || !adam) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam_all: NULL pointer");
    const bool small = dL_dmean3D || dL_dopacity || dL_dscale || dL_drot;
    if (small && !(dL_dmean3D && dL_dopacity && dL_dscale && dL_drot))
        return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam_all: the four small gradients are given together or not at all");
    for (int g = 0; g < 6; g++) {
        if ((g == 2 && M == 0) || (!small && g != 1 && g != 2)) continue;
        if (!adam->param[g] || !adam->exp_avg[g] || !adam->exp_avg_sq[g]) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_sh_grad_from_rgb_adam_all: group %d has a NULL pointer", g);
    }
    ShGradFromRgbArgs a;
    a.P = P; a.D = D; a.M = M; a.n_views = n_views; a.input_is_ddc = input_is_ddc ? 1 : 0;
    a.means3D = means3D; a.campos_all = campos_all; a.rgb_all = rgb_all; a.dL_ddc = nullptr; a.dL_dsh = nullptr;
    a.rgb_stride = view_stride ? (size_t)view_stride : (size_t)3 * (size_t)P; a.campos_stride = view_stride ? (size_t)view_stride : 3;
    a.visible = vis_all; a.vis_stride = (size_t)vis_stride; a.vis_out = vis_out;
    a.g_small[0] = dL_dmean3D; a.g_small[1] = dL_dopacity; a.g_small[2] = dL_dscale; a.g_small[3] = dL_drot;
    memset(&a.adam, 0, sizeof(a.adam));
    for (int g = 0; g < 6; g++) { a.adam.p[g] = adam->param[g]; a.adam.m[g] = adam->exp_avg[g]; a.adam.v[g] = adam->exp_avg_sq[g]; a.adam.lr[g] = adam->lr[g]; }
    a.adam.b1 = adam->b1; a.adam.b2 = adam->b2; a.adam.eps = adam->eps; a.adam.on = 1;
    return launch_sh_grad_from_rgb(a, (hipStream_t)stream);
}

int gslic_rasterize_backward_adam(const gslic_raster_params* prm, int32_t R, int32_t B, const float* background, const float* means3D,
                                  const float* dc, const float* shs, const float* colors_precomp, const float* scales,
                                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                  const float* cam_pos, const int32_t* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                                  char* sample_buffer, const float* dL_dpix, float* dL_dopacity, float* dL_dmean3D, float* dL_ddc,
                                  float* dL_dsh, float* dL_dscale, float* dL_drot, float lambda_erank, const gslic_adam_fused* adam,
                                  void* stream)
{
    if (!adam) return set_error(GSLIC_ERR_INVALID_ARG, "gslic_rasterize_backward_adam: adam descriptor is NULL");
    return rasterize_backward_impl(prm, R, B, background, means3D, dc, shs, colors_precomp, scales, rotations, cov3D_precomp, viewmatrix,
                                   projmatrix, cam_pos, radii, geom_buffer, binning_buffer, img_buffer, sample_buffer, dL_dpix, nullptr,
                                   nullptr, dL_dopacity, nullptr, dL_dmean3D, nullptr, dL_ddc, dL_dsh, dL_dscale, dL_drot, lambda_erank,
                                   adam, nullptr, stream, nullptr, 0, -1, false, adam->visible_out);
}

int gslic_rasterize_backward_camera(const gslic_raster_params* prm, int32_t R, int32_t B, const float* background, const float* means3D,
                                    const float* dc, const float* shs, const float* colors_precomp, const float* scales,
                                    const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                    const float* cam_pos, const int32_t* radii, char* geom_buffer, char* binning_buffer, char* img_buffer,
                                    char* sample_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity,
                                    float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_ddc, float* dL_dsh, float* dL_dscale,
                                    float* dL_drot, float lambda_erank, float* dL_dviewmatrix, float* dL_dprojmatrix, float* dL_dcampos,
                                    void* stream)
{
    if (!dL_dviewmatrix || !dL_dprojmatrix || !dL_dcampos)
        return set_error(GSLIC_ERR_INVALID_ARG, "gslic_rasterize_backward_camera: a camera-gradient output pointer is NULL");
    float* const cam[3] = {dL_dviewmatrix, dL_dprojmatrix, dL_dcampos};
    if (prm && prm->P == 0) {  // nothing rendered: zero gradients
        hipStream_t s = (hipStream_t)stream;
        GS_HIP(hipMemsetAsync(dL_dviewmatrix, 0, 16 * sizeof(float), s));
        GS_HIP(hipMemsetAsync(dL_dprojmatrix, 0, 16 * sizeof(float), s));
        GS_HIP(hipMemsetAsync(dL_dcampos, 0, 3 * sizeof(float), s));
        return GSLIC_OK;
    }
    return rasterize_backward_impl(prm, R, B, background, means3D, dc, shs, colors_precomp, scales, rotations, cov3D_precomp, viewmatrix,
                                   projmatrix, cam_pos, radii, geom_buffer, binning_buffer, img_buffer, sample_buffer, dL_dpix, dL_dmean2D,
                                   dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_ddc, dL_dsh, dL_dscale, dL_drot,
                                   lambda_erank, nullptr, cam, stream);
}

int gslic_adam_update(float* param, const float* param_grad, float* exp_avg, float* exp_avg_sq, const uint8_t* visible, float lr,
                      float b1, float b2, float eps, uint32_t N, uint32_t M, void* stream)
{
    if ((size_t)N * M == 0) return GSLIC_OK;
    if (!param || !param_grad || !exp_avg || !exp_avg_sq || !visible) return set_error(GSLIC_ERR_INVALID_ARG, "adam: NULL tensor pointer");
    return adam_update(param, param_grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M, (hipStream_t)stream);
}

int gslic_adam_update_groups(const gslic_adam_group* groups, int32_t n_groups, const uint8_t* visible, float b1, float b2, float eps,
                             uint32_t N, void* stream)
{
    if (n_groups <= 0 || N == 0) return GSLIC_OK;
    if (!groups || !visible) return set_error(GSLIC_ERR_INVALID_ARG, "adam groups: NULL pointer");
    // a group with M == 0 (features_rest [P,0,3] at SH degree 0) is a no-op, as adamUpdate is for an empty tensor (adam.cu:48-52)
    std::vector<gslic_adam_group> live;
    live.reserve((size_t)n_groups);
    for (int i = 0; i < n_groups; i++) {
        if (groups[i].M == 0) continue;
        if (!groups[i].param || !groups[i].grad || !groups[i].exp_avg || !groups[i].exp_avg_sq)
            return set_error(GSLIC_ERR_INVALID_ARG, "adam group %d has a NULL pointer", i);
        live.push_back(groups[i]);
    }
    if (live.empty()) return GSLIC_OK;
    return adam_update_groups(live.data(), (int)live.size(), visible, b1, b2, eps, N, (hipStream_t)stream);
}

int gslic_fusedssim_forward(int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, const float* img1, const float* img2,
                            float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream)
{
    if (B < 0 || CH < 0 || H < 0 || W < 0) return set_error(GSLIC_ERR_INVALID_ARG, "ssim: negative size");
    if ((size_t)B * CH * H * W == 0) return GSLIC_OK;
    if (!img1 || !img2 || !ssim_map) return set_error(GSLIC_ERR_INVALID_ARG, "ssim: NULL tensor pointer");
    const int nd = (dm_dmu1 != nullptr) + (dm_dsigma1_sq != nullptr) + (dm_dsigma12 != nullptr);
    if (nd != 0 && nd != 3) return set_error(GSLIC_ERR_INVALID_ARG, "ssim: pass all three derivative maps or none");
    return ssim_forward(B, CH, H, W, C1, C2, img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, (hipStream_t)stream);
}

int gslic_fusedssim_backward(int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, const float* img1, const float* img2,
                             const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                             float* dL_dimg1, void* stream)
{
    (void)C1; (void)C2;
    if (B < 0 || CH < 0 || H < 0 || W < 0) return set_error(GSLIC_ERR_INVALID_ARG, "ssim: negative size");
    if ((size_t)B * CH * H * W == 0) return GSLIC_OK;
    if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1)
        return set_error(GSLIC_ERR_INVALID_ARG, "ssim backward: NULL tensor pointer");
    return ssim_backward(B, CH, H, W, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1, (hipStream_t)stream);
}

int64_t gslic_loss_partials_count(int32_t B, int32_t CH, int32_t H, int32_t W)
{
    return (B <= 0 || CH <= 0 || H <= 0 || W <= 0) ? 8 : loss_partials_count(B, CH, H, W);
}

int gslic_l1_ssim_loss_forward(int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, const float* img, const float* gt,
                               float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, float* partials, float* terms, void* stream)
{
    if (B <= 0 || CH <= 0 || H <= 0 || W <= 0) return set_error(GSLIC_ERR_INVALID_ARG, "loss: empty image");
    if (!img || !gt || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !partials || !terms)
        return set_error(GSLIC_ERR_INVALID_ARG, "loss forward: NULL pointer");
    return loss_forward(B, CH, H, W, C1, C2, img, gt, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, partials, terms, (hipStream_t)stream);
}

int gslic_l1_ssim_loss_backward(int32_t B, int32_t CH, int32_t H, int32_t W, float lambda_dssim, const float* img, const float* gt,
                                const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg, void* stream)
{
    if (B <= 0 || CH <= 0 || H <= 0 || W <= 0) return set_error(GSLIC_ERR_INVALID_ARG, "loss: empty image");
    if (!img || !gt || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg) return set_error(GSLIC_ERR_INVALID_ARG, "loss backward: NULL pointer");
    return loss_backward(B, CH, H, W, lambda_dssim, img, gt, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg, (hipStream_t)stream);
}

int gslic_l1_ssim_loss_forward_backward(int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, float lambda_dssim, const float* img,
                                        const float* gt, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, float* partials, float* terms,
                                        float* dL_dimg, void* stream)
{
    if (B <= 0 || CH <= 0 || H <= 0 || W <= 0) return set_error(GSLIC_ERR_INVALID_ARG, "loss: empty image");
    if (!img || !gt || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !partials || !terms || !dL_dimg)
        return set_error(GSLIC_ERR_INVALID_ARG, "loss forward_backward: NULL pointer");
    return loss_forward_backward(B, CH, H, W, C1, C2, lambda_dssim, img, gt, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, partials, terms, dL_dimg,
                                 (hipStream_t)stream);
}

int gslic_knn_mean_dist2(int32_t P, const float* points, float* mean_dists, gslic_alloc_fn scratch_alloc, void* scratch_ctx,
                         void* stream)
{
    if (P < 0) return set_error(GSLIC_ERR_INVALID_ARG, "knn: negative P");
    if (P == 0) return GSLIC_OK;
    if (!points || !mean_dists || !scratch_alloc) return set_error(GSLIC_ERR_INVALID_ARG, "knn: NULL pointer");
    return knn_mean_dist2(P, points, mean_dists, scratch_alloc, scratch_ctx, (hipStream_t)stream);
}

int gslic_extend_select(int32_t n, const float* points, const float* depths_rsp, const float* R_cw, const float* t_cw, float fx, float fy,
                        float cx, float cy, int32_t width, int32_t height, const float* final_T, gslic_alloc_fn scratch_alloc,
                        void* scratch_ctx, uint32_t** keep_flags, uint32_t** keep_pos, int32_t* count, void* stream)
{
    if (!count || !keep_flags || !keep_pos) return set_error(GSLIC_ERR_INVALID_ARG, "extend: NULL output pointer");
    *count = 0; *keep_flags = nullptr; *keep_pos = nullptr;
    if (n < 0 || width <= 0 || height <= 0) return set_error(GSLIC_ERR_INVALID_ARG, "extend: bad sizes");
    if (n == 0) return GSLIC_OK;
    if (!points || !depths_rsp || !R_cw || !t_cw || !final_T || !scratch_alloc) return set_error(GSLIC_ERR_INVALID_ARG, "extend: NULL pointer");
    return extend_select(n, points, depths_rsp, R_cw, t_cw, fx, fy, cx, cy, width, height, final_T, scratch_alloc, scratch_ctx, keep_flags,
                         keep_pos, count, (hipStream_t)stream);
}

int gslic_extend_emit(int32_t n, const uint32_t* keep_flags, const uint32_t* keep_pos, const float* points, const float* colors,
                      const float* depths_rsp, float scaling_scale, float focal, int32_t M, float* xyz, float* dc, float* rest,
                      float* opacity, float* scaling, float* rotation, void* stream)
{
    if (n <= 0) return GSLIC_OK;
    if (!keep_flags || !keep_pos || !points || !colors || !depths_rsp || !xyz || !dc || !opacity || !scaling || !rotation || (M > 0 && !rest))
        return set_error(GSLIC_ERR_INVALID_ARG, "extend emit: NULL pointer");
    return extend_emit(n, keep_flags, keep_pos, points, colors, depths_rsp, scaling_scale, focal, M, xyz, dc, rest, opacity, scaling,
                       rotation, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
int gslic_profile_enable(int32_t on)
{
    g_prof_on = on != 0;
    g_prof_mask = (on == 1 || on == -1) ? 0xffffffffu : (uint32_t)on;  // 1 / -1: every kernel; otherwise bit i = kernel id i
    return GSLIC_OK;
}
int gslic_profile_reset(void)
{
    GS_HIP(hipDeviceSynchronize());
    for (auto& r : g_pending) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_pending.clear();
    memset(g_total_ms, 0, sizeof(g_total_ms));
    memset(g_launches, 0, sizeof(g_launches));
    return GSLIC_OK;
}
int gslic_profile_collect(void)
{
    GS_HIP(hipDeviceSynchronize());
    for (auto& r : g_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { g_total_ms[r.id] += ms; g_launches[r.id] += 1; }
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_pending.clear();
    return GSLIC_OK;
}
int gslic_profile_num_kernels(void) { return K_COUNT; }
const char* gslic_profile_kernel_name(int32_t id) { return (id >= 0 && id < K_COUNT) ? k_names[id] : ""; }
int gslic_profile_get(int32_t id, double* total_ms, int64_t* launches)
{
    if (id < 0 || id >= K_COUNT) return set_error(GSLIC_ERR_INVALID_ARG, "bad kernel id %d", id);
    if (total_ms) *total_ms = g_total_ms[id];
    if (launches) *launches = g_launches[id];
    return GSLIC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// test-only export of stage boundaries
namespace gslic {
__global__ __launch_bounds__(256) void export_geom_kernel(int P, const float4* __restrict__ rec, const uint32_t* __restrict__ tiles,
                                                          uint32_t* o_tiles, float* o_m2d, float* o_depth, float* o_co, float* o_rgb)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const bool vis = tiles[i] > 0;
    const float4 r0 = vis ? rec[GS_REC_F4 * (size_t)i] : make_float4(0, 0, 0, 0);
    const float4 r1 = vis ? rec[GS_REC_F4 * (size_t)i + 1] : make_float4(0, 0, 0, 0);
    const float4 r2 = vis ? rec[GS_REC_F4 * (size_t)i + 2] : make_float4(0, 0, 0, 0);
    if (o_tiles) o_tiles[i] = tiles[i];
    if (o_m2d) { o_m2d[2 * i] = r0.x; o_m2d[2 * i + 1] = r0.y; }
    if (o_depth) o_depth[i] = r2.y;
    if (o_co) { o_co[4 * i] = r0.z; o_co[4 * i + 1] = r0.w; o_co[4 * i + 2] = r1.x; o_co[4 * i + 3] = r1.y; }
    if (o_rgb) { o_rgb[3 * i] = r1.z; o_rgb[3 * i + 1] = r1.w; o_rgb[3 * i + 2] = r2.x; }
}
// the reference's sorted 64-bit keys (tile << 32 | depth bits), rebuilt from the tiles' ranges and the point list (one workgroup per tile)
__global__ __launch_bounds__(256) void export_keys_kernel(uint32_t R, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                          const float4* __restrict__ rec, uint64_t* __restrict__ o)
{
    const uint2 r = ranges[blockIdx.x];
    for (uint32_t k = r.x + threadIdx.x; k < r.y && k < R; k += 256u)
        o[k] = ((uint64_t)blockIdx.x << 32) | __float_as_uint(rec[GS_REC_F4 * (size_t)point_list[k] + 2].y);
}
__global__ __launch_bounds__(256) void export_ncontrib_kernel(int W, int H, int gx, const float4* __restrict__ pix_final, uint32_t* o)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const int tile = blockIdx.y * gx + blockIdx.x;
    o[(size_t)y * W + x] = __float_as_uint(pix_final[(size_t)tile * GS_TILE_PIX + tile_pix_index((int)(threadIdx.x & 15), (int)(threadIdx.x >> 4))].w);
}
}  // namespace gslic

extern "C" int gslic_debug_export(const gslic_raster_params* prm, int32_t R, int32_t B, const char* geom_buffer,
                                  const char* binning_buffer, const char* img_buffer, const char* sample_buffer,
                                  uint32_t* tiles_touched, float* means2D, float* depths, float* conic_opacity, float* rgb,
                                  uint64_t* sorted_keys, uint32_t* point_list, uint32_t* ranges, uint32_t* n_contrib,
                                  uint32_t* max_contrib, void* stream)
{
    (void)B; (void)sample_buffer;
    GS_TRY(check_params(prm));
    hipStream_t s = (hipStream_t)stream;
    const int P = prm->P;
    int gx, gy;
    const int T = tile_grid(prm->width, prm->height, gx, gy);
    if (P == 0) return GSLIC_OK;
    GeomState geom = GeomState::carve(align256(const_cast<char*>(geom_buffer)), (size_t)P, nullptr);
    ImageState img = ImageState::carve(align256(const_cast<char*>(img_buffer)), (size_t)T, nullptr);
    BinningState bin = BinningState::carve(align256(const_cast<char*>(binning_buffer)), (size_t)R, sort_end_bit(T), prm->no_color != 0, nullptr);
    if (tiles_touched || means2D || depths || conic_opacity || rgb)
        GS_LAUNCH(K_DEBUG_EXPORT, export_geom_kernel, dim3(div_up(P, 256)), dim3(256), 0, s, P, (const float4*)geom.rec,
                  (const uint32_t*)geom.tiles_touched, tiles_touched, means2D, depths, conic_opacity, rgb);
    if (R > 0 && sorted_keys)
        GS_LAUNCH(K_DEBUG_EXPORT, export_keys_kernel, dim3((unsigned)T), dim3(256), 0, s, (uint32_t)R,
                  (const uint2*)img.ranges, (const uint32_t*)bin.point_list(), (const float4*)geom.rec, sorted_keys);
    if (R > 0 && point_list)
        GS_HIP(hipMemcpyAsync(point_list, bin.point_list(), (size_t)R * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    if (ranges) GS_HIP(hipMemcpyAsync(ranges, img.ranges, (size_t)T * sizeof(uint2), hipMemcpyDeviceToDevice, s));
    if (!prm->no_color) {
        if (max_contrib) GS_HIP(hipMemcpyAsync(max_contrib, img.max_contrib, (size_t)T * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        if (n_contrib)
            GS_LAUNCH(K_DEBUG_EXPORT, export_ncontrib_kernel, dim3(gx, gy), dim3(256), 0, s, prm->width, prm->height, gx,
                      (const float4*)img.pix_final, n_contrib);
    }
    return GSLIC_OK;
}
