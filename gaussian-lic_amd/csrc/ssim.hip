// ssim.hip — fused SSIM map forward/backward (replaces fusedssimCUDA / fusedssim_backwardCUDA,
// src/fused-ssim/ssim.cu:186-365).  11-tap separable Gaussian window, zero padding, x pass then y pass, taps
// accumulated 0..10 from 0.0f like the reference.
//
// One 256-thread block per 32x32 output tile and per (batch, channel) plane (grid.z), four outputs per thread.
// The 42x42 input halo tiles are staged once in LDS; the horizontal pass produces ALL statistics of a row in
// one sweep (forward: E[a], E[a^2], E[b], E[b^2], E[ab]; backward: the three dL-weighted maps) into LDS, then
// the vertical pass finishes them — 3 barriers per plane instead of the reference's ~25 per channel, and no
// scratch flush.  LDS: forward 14.1 KB + 21.5 KB, backward 21.2 KB (the horizontal sums overwrite the rows they came from).  Forward: the two images interleaved in LDS, statistics on
// float2 pairs (packed multiply + packed add); backward: three separate maps, scalar arithmetic (see the note above ssim_bwd_kernel).
// Compiled with -ffp-contract=off (build.py): no product is fused into a sum, so the maps do not depend on instruction selection.
#include "gslic_common.h"

namespace gslic {

static constexpr int ST = 32;        // output tile edge
static constexpr int SH_ = ST + 10;  // halo tile edge (42)
static constexpr int HALO_TRIPS = (SH_ * SH_ + 255) / 256;  // halo elements per thread (7)

__constant__ float c_G[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f,
                              0.21300552785396576f,  0.26601171493530273f,   0.21300552785396576f,  0.10936068743467331f,
                              0.036000773310661316f, 0.0075987582094967365f, 0.001028380123898387f};

// blockIdx -> (tile x, tile y, plane).  Workgroup i of a 1-D grid runs on XCD i % 8 (eight private L2s); a tile shares 10 of its 42 halo columns /
// rows with each neighbour.  With a plain dim3(W/32, H/32, planes) grid neighbouring tiles land on different XCDs and every shared halo line is
// fetched from HBM by both (loss_fwd 2.2x, loss_bwd 2.4x its algorithmic bytes, L2 hit rate 0.23 / 0.09: profiles/r03x_cache_lds.md).  Here
// XCD x works through the x-th contiguous eighth of the tiles (plane by plane, bands of four tile rows walked column by column), so that a
// tile's neighbours ran shortly before it on the same L2.  Returns false for the padding workgroups of the last eighth.
struct SsimGrid { int gx, gy, planes, per_xcd; };
__device__ __forceinline__ bool ssim_tile(const SsimGrid& g, int& tx, int& ty, int& plane, size_t& linear)
{
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int t = xcd * g.per_xcd + j;
    if (j >= g.per_xcd || t >= g.gx * g.gy * g.planes) return false;
    plane = t / (g.gx * g.gy);
    const int r = t - plane * (g.gx * g.gy);
    // inside a plane: bands of four tile rows, column by column — a tile's upper neighbour ran just before it, its left neighbour four tiles earlier
    const int band = r / (4 * g.gx), rem = r - band * (4 * g.gx);
    const int bh = (g.gy - 4 * band) < 4 ? (g.gy - 4 * band) : 4;
    tx = rem / bh; ty = 4 * band + (rem - tx * bh);
    linear = (size_t)t;
    return true;
}
static inline SsimGrid ssim_grid(int W, int H, int planes, unsigned& blocks)
{
    SsimGrid g;
    g.gx = div_up(W, 32); g.gy = div_up(H, 32); g.planes = planes;
    g.per_xcd = div_up(g.gx * g.gy * planes, 8);
    blocks = 8u * (unsigned)g.per_xcd;
    return g;
}

__device__ __forceinline__ float pix_or_zero(const float* __restrict__ img, int H, int W, int y, int x)
{
    return (x >= W || y >= H || x < 0 || y < 0) ? 0.0f : img[(size_t)y * W + x];
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- shared cores.  The two images (the three derivative maps) sit INTERLEAVED in LDS — one ds_read_b64 fetches the pixel pair — and
// the statistics are accumulated on float2 pairs with v_pk_mul_f32 + v_pk_add_f32 (fewer and wider LDS reads); every component sees the
// reference's operation sequence (taps 0..10 from 0.0f, product and sum rounded separately: the file is compiled with -ffp-contract=off).
struct FwdLds {
    v2f ab[SH_][SH_];   // {img1, img2} halo tile; after the horizontal pass the first 32 floats of row ly hold that row's E[ab] (h1)
    v4f h4[SH_][ST];    // horizontal pass: {E[a], E[b], E[a^2], E[b^2]}
    // E[ab] of the horizontal pass overwrites the head of the halo row it was computed from (128 of its 336 bytes): the wave that owns a
    // row has read all of it by then (two rows per wave, lockstep), and nothing reads the halo tile afterwards.  35.6 KB instead of 41.0:
    // four workgroups per CU instead of three (the kernel loses 14 % from three to two: profiles/r03t_occupancy_sweep.log).
    __device__ __forceinline__ float* h1(int ly) { return reinterpret_cast<float*>(&ab[ly][0]); }
    __device__ __forceinline__ const float* h1(int ly) const { return reinterpret_cast<const float*>(&ab[ly][0]); }
};
struct FwdStats { float mu1, mu2, e11, e22, e12; };

// l1: += |a - b| over this thread's elements of the 32 x 32 centre (0 outside the image: both read as zero there)
__device__ __forceinline__ void ssim_fwd_stage(FwdLds& L, const float* __restrict__ a, const float* __restrict__ b, int H, int W, int x0, int y0, float& l1)
{
    const int tid = threadIdx.x;
    // all of a thread's halo loads are issued before the first LDS store: a rolled loop pays one memory round trip per trip
    float va[HALO_TRIPS], vb[HALO_TRIPS];
#pragma unroll
    for (int k = 0; k < HALO_TRIPS; k++) {
        const int t = tid + 256 * k;
        const int ly = t / SH_, lx = t - ly * SH_;
        va[k] = (t < SH_ * SH_) ? pix_or_zero(a, H, W, y0 + ly - 5, x0 + lx - 5) : 0.f;
        vb[k] = (t < SH_ * SH_) ? pix_or_zero(b, H, W, y0 + ly - 5, x0 + lx - 5) : 0.f;
        if (t < SH_ * SH_ && ly >= 5 && ly < 5 + ST && lx >= 5 && lx < 5 + ST) l1 += fabsf(va[k] - vb[k]);
    }
#pragma unroll
    for (int k = 0; k < HALO_TRIPS; k++) {
        const int t = tid + 256 * k;
        if (t < SH_ * SH_) (&L.ab[0][0])[t] = (v2f){va[k], vb[k]};
    }
    __syncthreads();
    for (int t = tid; t < SH_ * ST; t += 256) {
        const int ly = t / ST, lx = t % ST;
        v2f r02 = {0.f, 0.f}, r13 = {0.f, 0.f};
        float r4 = 0.f;
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const v2f p = L.ab[ly][lx + i];
            const v2f g2 = {c_G[i], c_G[i]};
            r02 = r02 + g2 * p;                 // val += G_i * pixel            (ssim.cu:116-126; this file is compiled with -ffp-contract=off:
            r13 = r13 + g2 * (p * p);           // val += G_i * do_sq(pixel)      every product and every sum is rounded on its own, like the
            r4 = r4 + c_G[i] * (p.x * p.y);     // val += G_i * (pix1 * pix2)     reference's kernels under the same flag — bit-identical maps)
        }
        L.h4[ly][lx] = (v4f){r02.x, r02.y, r13.x, r13.y};
        L.h1(ly)[lx] = r4;
    }
    __syncthreads();
}
// Vertical pass for FOUR consecutive output rows ly0 .. ly0+3 of column lx in one sweep over the 14 rows they share: row r feeds tap r - k of
// output k, so every output still sees its taps 0..10 in ascending order from 0.0f (the reference's operation sequence), with 14 LDS
// row reads instead of 44 — the LDS pipe, one per CU, was what these kernels waited for.
__device__ __forceinline__ void ssim_fwd_column4(const FwdLds& L, int ly0, int lx, FwdStats (&out)[4])
{
    v2f v02[4], v13[4];
    float v4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { v02[k] = (v2f){0.f, 0.f}; v13[k] = (v2f){0.f, 0.f}; v4[k] = 0.f; }
#pragma unroll
    for (int r = 0; r < 14; r++) {
        const v4f q = L.h4[ly0 + r][lx];
        const float q1 = L.h1(ly0 + r)[lx];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = r - k;
            if (j >= 0 && j <= 10) {
                const v2f g2 = {c_G[j], c_G[j]};
                v02[k] = v02[k] + g2 * ((v2f){q.x, q.y});
                v13[k] = v13[k] + g2 * ((v2f){q.z, q.w});
                v4[k] = v4[k] + c_G[j] * q1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { out[k].mu1 = v02[k].x; out[k].mu2 = v02[k].y; out[k].e11 = v13[k].x; out[k].e22 = v13[k].y; out[k].e12 = v4[k]; }
}

// SSIM of one pixel from its five window statistics, and the three derivative maps the backward consumes (ssim.cu:261-282).
// SSIM = (lum_n * con_n) / (lum_d * con_d): luminance and contrast-structure terms, numerators and denominators; every quotient keeps the
// reference's association and, with contraction off, its roundings: the maps equal the reference kernels' (and the CPU oracle's) bit for bit
// (tests/test_vs_reference_kernels_gpu.py).  One definition for the drop-in kernel and the fused loss kernel.
struct SsimTerms { float map, d_mu1, d_sigma1_sq, d_sigma12; };
__device__ __forceinline__ SsimTerms ssim_terms(const FwdStats& st, float C1, float C2)
{
    const float mu1 = st.mu1, mu2 = st.mu2;
    const float sigma1_sq = st.e11 - mu1 * mu1;
    const float sigma2_sq = st.e22 - mu2 * mu2;
    const float sigma12 = st.e12 - mu1 * mu2;
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
    const float lum_n = (2.0f * mu1_mu2 + C1);
    const float con_n = (2.0f * sigma12 + C2);
    const float lum_d = (mu1_sq + mu2_sq + C1);
    const float con_d = (sigma1_sq + sigma2_sq + C2);
    SsimTerms t;
    t.map = (lum_n * con_n) / (lum_d * con_d);
    t.d_mu1 = ((mu2 * 2.0f * con_n) / (lum_d * con_d) - (mu2 * 2.0f * lum_n) / (lum_d * con_d) - (mu1 * 2.0f * lum_n * con_n) / (lum_d * lum_d * con_d) +
               (mu1 * 2.0f * lum_n * con_n) / (lum_d * con_d * con_d));
    t.d_sigma1_sq = ((-lum_n * con_n) / (lum_d * con_d * con_d));
    t.d_sigma12 = ((2 * lum_n) / (lum_d * con_d));
    return t;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(SsimGrid grid, int H, int W, float C1, float C2, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                       float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                       float* __restrict__ dm_dsigma12)
{
    __shared__ FwdLds L;
    int tile_x, tile_y, plane_i;
    size_t tile_linear;
    if (!ssim_tile(grid, tile_x, tile_y, plane_i, tile_linear)) return;   // (whole workgroup: before any barrier)
    const size_t plane = (size_t)plane_i * H * W;
    const int x0 = tile_x * ST, y0 = tile_y * ST;
    const int tid = threadIdx.x;
    float l1_unused = 0.f;
    ssim_fwd_stage(L, img1 + plane, img2 + plane, H, W, x0, y0, l1_unused);
    const int lx = tid & 31;
    FwdStats st4[4];
    ssim_fwd_column4(L, 4 * (tid >> 5), lx, st4);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int ly = 4 * (tid >> 5) + q;
        const FwdStats st = st4[q];
        const int px = x0 + lx, py = y0 + ly;
        if (px < W && py < H) {
            const SsimTerms t = ssim_terms(st, C1, C2);
            const size_t o = plane + (size_t)py * W + px;
            ssim_map[o] = t.map;
            if (dm_dmu1) {
                dm_dmu1[o] = t.d_mu1;
                dm_dsigma1_sq[o] = t.d_sigma1_sq;
                dm_dsigma12[o] = t.d_sigma12;
            }
        }
    }
}

// The backward keeps three separate LDS arrays and scalar arithmetic: the interleaved / packed variant of the forward was measured slower here
// (0.063 -> 0.071 ms: three maps do not pair up, the float2 rows of 42 conflict on the LDS banks).
__global__ __launch_bounds__(256) void ssim_bwd_kernel(SsimGrid grid, int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                       const float* __restrict__ dL_dmap, const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12,
                                                       float* __restrict__ dL_dimg1)
{
    // the horizontal pass overwrites the first 32 floats of each row with that row's sums (the wave that owns a row has read all of it
    // by then: two rows per wave, lockstep): 21.2 KB instead of 37.3 — six workgroups per CU instead of four
    __shared__ float s1[SH_][SH_];
    __shared__ float s2[SH_][SH_];
    __shared__ float s3[SH_][SH_];
    int tile_x, tile_y, plane_i;
    size_t tile_linear;
    if (!ssim_tile(grid, tile_x, tile_y, plane_i, tile_linear)) return;   // (whole workgroup: before any barrier)
    const size_t plane = (size_t)plane_i * H * W;
    const int x0 = tile_x * ST, y0 = tile_y * ST;
    const int tid = threadIdx.x;
    {
        float v0[HALO_TRIPS], v1[HALO_TRIPS], v2[HALO_TRIPS], v3[HALO_TRIPS];
#pragma unroll
        for (int k = 0; k < HALO_TRIPS; k++) {
            const int t = tid + 256 * k;
            const int ly = t / SH_, lx = t - ly * SH_;
            const int y = y0 + ly - 5, x = x0 + lx - 5;
            const bool in = t < SH_ * SH_;
            v0[k] = in ? pix_or_zero(dL_dmap + plane, H, W, y, x) : 0.f;
            v1[k] = in ? pix_or_zero(dm_dmu1 + plane, H, W, y, x) : 0.f;
            v2[k] = in ? pix_or_zero(dm_dsigma1_sq + plane, H, W, y, x) : 0.f;
            v3[k] = in ? pix_or_zero(dm_dsigma12 + plane, H, W, y, x) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < HALO_TRIPS; k++) {
            const int t = tid + 256 * k;
            if (t < SH_ * SH_) { (&s1[0][0])[t] = v1[k] * v0[k]; (&s2[0][0])[t] = v2[k] * v0[k]; (&s3[0][0])[t] = v3[k] * v0[k]; }
        }
    }
    __syncthreads();
    for (int t = tid; t < SH_ * ST; t += 256) {
        const int ly = t / ST, lx = t % ST;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const float g = c_G[i];
            r0 += g * s1[ly][lx + i]; r1 += g * s2[ly][lx + i]; r2 += g * s3[ly][lx + i];
        }
        s1[ly][lx] = r0; s2[ly][lx] = r1; s3[ly][lx] = r2;
    }
    __syncthreads();
    const int lx = tid & 31;
    // four consecutive rows per thread, one sweep over the 14 rows of hs they share (ssim_fwd_column4): taps in ascending order per output
    const int ly0 = 4 * (tid >> 5);
    float w0[4] = {0.f, 0.f, 0.f, 0.f}, w1[4] = {0.f, 0.f, 0.f, 0.f}, w2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 14; r++) {
        const float a0 = s1[ly0 + r][lx], a1 = s2[ly0 + r][lx], a2 = s3[ly0 + r][lx];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = r - k;
            if (j >= 0 && j <= 10) {
                const float g = c_G[j];
                w0[k] += g * a0; w1[k] += g * a1; w2[k] += g * a2;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int ly = ly0 + q;
        const float v0 = w0[q], v1 = w1[q], v2 = w2[q];
        const int px = x0 + lx, py = y0 + ly;
        if (px < W && py < H) {
            const size_t o = plane + (size_t)py * W + px;
            const float pix1 = img1[o], pix2 = img2[o];
            float acc = 0.0f;
            acc += v0;
            acc += pix1 * 2.0f * v1;
            acc += pix2 * v2;
            dL_dimg1[o] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Loss of optimize() (gaussian.cpp:685-691) with l1_loss (loss_utils.h:30-33) folded into the SSIM passes.
// loss_fwd_kernel = ssim_fwd_kernel without the ssim_map store, plus per-block sums of |a-b| and of the SSIM values;
// loss_reduce_kernel adds the per-block partials in a fixed order (bit-reproducible, no float atomics).
__device__ __forceinline__ float block256_sum(float v, float* red /*[4]*/)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void loss_fwd_kernel(SsimGrid grid, int H, int W, float C1, float C2, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, float* __restrict__ dm_dmu1,
                                                       float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12,
                                                       float* __restrict__ partials)
{
    __shared__ FwdLds L;
    __shared__ float red[4];
    int tile_x, tile_y, plane_i;
    size_t tile_linear;
    if (!ssim_tile(grid, tile_x, tile_y, plane_i, tile_linear)) return;   // (whole workgroup: before any barrier)
    const size_t plane = (size_t)plane_i * H * W;
    const int x0 = tile_x * ST, y0 = tile_y * ST;
    const int tid = threadIdx.x;
    float sum_l1 = 0.f, sum_ssim = 0.f;
    ssim_fwd_stage(L, img1 + plane, img2 + plane, H, W, x0, y0, sum_l1);
    const int lx = tid & 31;
    FwdStats st4[4];
    ssim_fwd_column4(L, 4 * (tid >> 5), lx, st4);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int ly = 4 * (tid >> 5) + q;
        const FwdStats st = st4[q];
        const int px = x0 + lx, py = y0 + ly;
        if (px < W && py < H) {
            const SsimTerms t = ssim_terms(st, C1, C2);
            const size_t o = plane + (size_t)py * W + px;
            sum_ssim += t.map;
            dm_dmu1[o] = t.d_mu1;
            dm_dsigma1_sq[o] = t.d_sigma1_sq;
            dm_dsigma12[o] = t.d_sigma12;
        }
    }
    const float bl1 = block256_sum(sum_l1, red);
    const float bss = block256_sum(sum_ssim, red);
    if (tid == 0) {
        const size_t blk = tile_linear;
        partials[2 * blk] = bl1;
        partials[2 * blk + 1] = bss;
    }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(size_t nblk, const float* __restrict__ partials, float inv_n, float* __restrict__ terms)
{
    __shared__ float red[4];
    float s0 = 0.f, s1 = 0.f;
    for (size_t i = threadIdx.x; i < nblk; i += 256) { s0 += partials[2 * i]; s1 += partials[2 * i + 1]; }
    const float t0 = block256_sum(s0, red);
    const float t1 = block256_sum(s1, red);
    if (threadIdx.x == 0) { terms[0] = t0 * inv_n; terms[1] = t1 * inv_n; }
}

// dL/dimg for dL/dloss = 1: the SSIM branch is ssim_bwd_kernel with the uniform upstream gradient dL_dmap = -lambda/N (multiplied into the
// derivative maps before the convolutions, exactly where fusedssim_backwardCUDA multiplies by dL_dmap: ssim.cu:318-350), plus the L1 branch
// (1-lambda)/N * sign(img - gt): bit-identical to fusedssim_backward(full(-lambda/N)) + (1-lambda)/N * sign(img - gt).
__global__ __launch_bounds__(256) void loss_bwd_kernel(SsimGrid grid, int H, int W, float w_l1, float w_ssim, const float* __restrict__ img1,
                                                       const float* __restrict__ img2, const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12,
                                                       float* __restrict__ dL_dimg1, size_t red_nblk, const float* __restrict__ red_partials,
                                                       float red_inv_n, float* __restrict__ red_terms)
{
    // the horizontal pass overwrites the first 32 floats of each row with that row's sums (the wave that owns a row has read all of it
    // by then: two rows per wave, lockstep): 21.2 KB instead of 37.3 — six workgroups per CU instead of four
    __shared__ float s1[SH_][SH_];
    __shared__ float s2[SH_][SH_];
    __shared__ float s3[SH_][SH_];
    if (red_terms && blockIdx.x == 0) {
        // gslic_l1_ssim_loss_forward_backward: the sums of the forward's per-block partials (loss_reduce_kernel's loop and order: the same bits)
        // ride on workgroup 0 of the backward instead of being a launch of their own between the two loss kernels
        float* const red = &s1[0][0];
        float r0 = 0.f, r1 = 0.f;
        for (size_t i = threadIdx.x; i < red_nblk; i += 256) { r0 += red_partials[2 * i]; r1 += red_partials[2 * i + 1]; }
        const float t0 = block256_sum(r0, red);
        const float t1 = block256_sum(r1, red);
        if (threadIdx.x == 0) { red_terms[0] = t0 * red_inv_n; red_terms[1] = t1 * red_inv_n; }
    }
    int tile_x, tile_y, plane_i;
    size_t tile_linear;
    if (!ssim_tile(grid, tile_x, tile_y, plane_i, tile_linear)) return;   // (whole workgroup: before any barrier)
    const size_t plane = (size_t)plane_i * H * W;
    const int x0 = tile_x * ST, y0 = tile_y * ST;
    const int tid = threadIdx.x;
    {
        float v1[HALO_TRIPS], v2[HALO_TRIPS], v3[HALO_TRIPS];
#pragma unroll
        for (int k = 0; k < HALO_TRIPS; k++) {
            const int t = tid + 256 * k;
            const int ly = t / SH_, lx = t - ly * SH_;
            const int y = y0 + ly - 5, x = x0 + lx - 5;
            const bool in = t < SH_ * SH_;
            v1[k] = in ? pix_or_zero(dm_dmu1 + plane, H, W, y, x) : 0.f;
            v2[k] = in ? pix_or_zero(dm_dsigma1_sq + plane, H, W, y, x) : 0.f;
            v3[k] = in ? pix_or_zero(dm_dsigma12 + plane, H, W, y, x) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < HALO_TRIPS; k++) {
            const int t = tid + 256 * k;
            if (t < SH_ * SH_) { (&s1[0][0])[t] = v1[k] * w_ssim; (&s2[0][0])[t] = v2[k] * w_ssim; (&s3[0][0])[t] = v3[k] * w_ssim; }
        }
    }
    __syncthreads();
    for (int t = tid; t < SH_ * ST; t += 256) {
        const int ly = t / ST, lx = t % ST;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int i = 0; i < 11; i++) {
            const float g = c_G[i];
            r0 += g * s1[ly][lx + i]; r1 += g * s2[ly][lx + i]; r2 += g * s3[ly][lx + i];
        }
        s1[ly][lx] = r0; s2[ly][lx] = r1; s3[ly][lx] = r2;
    }
    __syncthreads();
    const int lx = tid & 31;
    // four consecutive rows per thread, one sweep over the 14 rows of hs they share (ssim_fwd_column4): taps in ascending order per output
    const int ly0 = 4 * (tid >> 5);
    float w0[4] = {0.f, 0.f, 0.f, 0.f}, w1[4] = {0.f, 0.f, 0.f, 0.f}, w2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 14; r++) {
        const float a0 = s1[ly0 + r][lx], a1 = s2[ly0 + r][lx], a2 = s3[ly0 + r][lx];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = r - k;
            if (j >= 0 && j <= 10) {
                const float g = c_G[j];
                w0[k] += g * a0; w1[k] += g * a1; w2[k] += g * a2;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int ly = ly0 + q;
        const float v0 = w0[q], v1 = w1[q], v2 = w2[q];
        const int px = x0 + lx, py = y0 + ly;
        if (px < W && py < H) {
            const size_t o = plane + (size_t)py * W + px;
            const float pix1 = img1[o], pix2 = img2[o];
            const float d = pix1 - pix2;
            const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);  // torch.sign (abs backward)
            float acc = 0.0f;
            acc += v0;
            acc += pix1 * 2.0f * v1;
            acc += pix2 * v2;
            dL_dimg1[o] = w_l1 * sgn + acc;
        }
    }
}

int loss_forward(int B, int CH, int H, int W, float C1, float C2, const float* img, const float* gt, float* d1, float* d2, float* d3,
                 float* partials, float* terms, hipStream_t s)
{
    unsigned blocks;
    const SsimGrid grid = ssim_grid(W, H, B * CH, blocks);
    GS_LAUNCH(K_SSIM_FWD, loss_fwd_kernel, dim3(blocks), dim3(256), 0, s, grid, H, W, C1, C2, img, gt, d1, d2, d3, partials);
    const size_t nblk = (size_t)grid.gx * grid.gy * grid.planes;
    GS_LAUNCH(K_SSIM_FWD, loss_reduce_kernel, dim3(1), dim3(256), 0, s, nblk, (const float*)partials,
              1.0f / (float)((size_t)B * CH * H * W), terms);
    return GSLIC_OK;
}
int loss_backward(int B, int CH, int H, int W, float lambda_dssim, const float* img, const float* gt, const float* d1, const float* d2,
                  const float* d3, float* dL_dimg, hipStream_t s)
{
    unsigned blocks;
    const SsimGrid grid = ssim_grid(W, H, B * CH, blocks);
    const float n = (float)((size_t)B * CH * H * W);
    GS_LAUNCH(K_SSIM_BWD, loss_bwd_kernel, dim3(blocks), dim3(256), 0, s, grid, H, W, (1.0f - lambda_dssim) / n, -lambda_dssim / n, img, gt, d1, d2,
              d3, dL_dimg, (size_t)0, (const float*)nullptr, 0.0f, (float*)nullptr);
    return GSLIC_OK;
}
// forward + backward of the loss in TWO launches: the reduction of the forward's partial sums rides on workgroup 0 of the backward
int loss_forward_backward(int B, int CH, int H, int W, float C1, float C2, float lambda_dssim, const float* img, const float* gt, float* d1,
                          float* d2, float* d3, float* partials, float* terms, float* dL_dimg, hipStream_t s)
{
    unsigned blocks;
    const SsimGrid grid = ssim_grid(W, H, B * CH, blocks);
    const float n = (float)((size_t)B * CH * H * W);
    GS_LAUNCH(K_SSIM_FWD, loss_fwd_kernel, dim3(blocks), dim3(256), 0, s, grid, H, W, C1, C2, img, gt, d1, d2, d3, partials);
    const size_t nblk = (size_t)grid.gx * grid.gy * grid.planes;
    GS_LAUNCH(K_SSIM_BWD, loss_bwd_kernel, dim3(blocks), dim3(256), 0, s, grid, H, W, (1.0f - lambda_dssim) / n, -lambda_dssim / n, img, gt,
              (const float*)d1, (const float*)d2, (const float*)d3, dL_dimg, nblk, (const float*)partials, 1.0f / n, terms);
    return GSLIC_OK;
}
int64_t loss_partials_count(int B, int CH, int H, int W) { return 2 * (int64_t)div_up(W, ST) * div_up(H, ST) * B * CH + 8; }

int ssim_forward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, float* ssim_map,
                 float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, hipStream_t s)
{
    if (B * CH == 0 || H == 0 || W == 0) return GSLIC_OK;
    unsigned blocks;
    const SsimGrid grid = ssim_grid(W, H, B * CH, blocks);
    GS_LAUNCH(K_SSIM_FWD, ssim_fwd_kernel, dim3(blocks), dim3(256), 0, s, grid, H, W, C1, C2, img1, img2, ssim_map, dm_dmu1, dm_dsigma1_sq,
              dm_dsigma12);
    return GSLIC_OK;
}
int ssim_backward(int B, int CH, int H, int W, const float* img1, const float* img2, const float* dL_dmap, const float* dm_dmu1,
                  const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, hipStream_t s)
{
    if (B * CH == 0 || H == 0 || W == 0) return GSLIC_OK;
    unsigned blocks;
    const SsimGrid grid = ssim_grid(W, H, B * CH, blocks);
    GS_LAUNCH(K_SSIM_BWD, ssim_bwd_kernel, dim3(blocks), dim3(256), 0, s, grid, H, W, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
              dL_dimg1);
    return GSLIC_OK;
}

}  // namespace gslic
