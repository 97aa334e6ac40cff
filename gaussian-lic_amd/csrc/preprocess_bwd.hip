// preprocess_bwd.hip — one fused per-Gaussian backward pass.
//
// preprocess_bwd_kernel replaces, in ONE launch and one read of each input,
//   * the nine atomicAdd targets of PerGaussianRenderCUDA (backward.cu:585-596): here a contiguous segmented
//     sum over the Gaussian's emission slots [offsets[g-1], offsets[g]) of the partials render_bwd wrote;
//   * computeCov2DCUDA (backward.cu:138-255);
//   * preprocessCUDA<3> bwd + computeColorFromSH bwd + computeCov3D bwd (backward.cu:27-136,257-377);
//   * the ten torch::zeros() of RasterizeGaussiansBackwardCUDA (rasterize_points.cu:192-201): every output row
//     is written here, zeros for invisible Gaussians, so callers may pass uninitialised memory.
// Sigma_3D is recomputed from scale/rotation instead of being stored by the forward (24 B/Gaussian saved).
#include "gslic_common.h"
#include "kernels.h"
#include <stdlib.h>

namespace gslic {

#define SHC0 0.28209479177387814f
#define SHC1 0.4886025119029199f
__constant__ float b_SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                 0.5462742152960396f};
__constant__ float b_SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                 -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// View direction of the SH evaluation (forward.cu:29-35): normalize(p - campos).  Shared by the backward and by sh_grad_from_rgb_kernel,
// which must reproduce the backward's numbers bit for bit: no contraction in here (the backend would otherwise fuse differently at
// the two call sites).
__device__ __forceinline__ void sh_dir(float px, float py, float pz, const float* __restrict__ campos, float& dox, float& doy, float& doz, float& x,
                                       float& y, float& z)
{
#pragma clang fp contract(off)
    dox = px - campos[0]; doy = py - campos[1]; doz = pz - campos[2];
    const float len = sqrtf(dox * dox + doy * doy + doz * doz);
    x = dox / len; y = doy / len; z = doz / len;
}

// Direction-only SH coefficients (the factors multiplying sh[k] in forward.cu:37-66), k = 0..14.  (No contraction: see sh_dir.)
__device__ __forceinline__ void sh_coefs(int D, float x, float y, float z, float (&c)[15])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int k = 0; k < 15; k++) c[k] = 0.f;
    if (D > 0) {
        c[0] = -SHC1 * y; c[1] = SHC1 * z; c[2] = -SHC1 * x;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            c[3] = b_SH_C2[0] * xy; c[4] = b_SH_C2[1] * yz; c[5] = b_SH_C2[2] * (2.f * zz - xx - yy);
            c[6] = b_SH_C2[3] * xz; c[7] = b_SH_C2[4] * (xx - yy);
            if (D > 2) {
                c[8] = b_SH_C3[0] * y * (3.f * xx - yy); c[9] = b_SH_C3[1] * xy * z; c[10] = b_SH_C3[2] * y * (4.f * zz - xx - yy);
                c[11] = b_SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); c[12] = b_SH_C3[4] * x * (4.f * zz - xx - yy);
                c[13] = b_SH_C3[5] * z * (xx - yy); c[14] = b_SH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

struct ShOut {  // what phase B needs to write a Gaussian's dL_dsh row
    float x, y, z, dR, dG, dB;
    bool on;
};

struct PartialSums { float mx, my, cx, cy, cw, op, r, g, b; };   // the nine per-Gaussian sums of render_bwd's per-instance rows

__device__ __forceinline__ bool bwd_gather(const PreprocessBwdArgs& a, const int idx, const bool staged, PartialSums& ps);
template <bool CAM>
__device__ __forceinline__ void bwd_rest(const PreprocessBwdArgs& a, const int idx, const PartialSums& ps, const float* sh_row, const float* sk_row,
                                         ShOut& so, float* sg, float* cg);

// Cooperative sink of the 14 small gradients per Gaussian of one block (see lds_g in the kernel): coalesced float4 stores into the
// gradient tensors that were asked for (gout, group order of lds_g) and / or the in-place Adam update.  Group g of width w: the
// block's region is rows * w floats starting at param[g] + row0 * w, 16-byte aligned because BS is a multiple of 4.
template <int BS>
__device__ __forceinline__ void small_groups_sink(const AdamFusedArgs& A, float* const (&gout)[5], const float* lds_g, const uint8_t* lds_vis, int row0,
                                                  int rows)
{
    constexpr int NG = 5;
    constexpr int gid[NG] = {0, 1, 3, 4, 5};   // xyz, features_dc, opacity, scaling, rotation
    constexpr int wid[NG] = {3, 3, 1, 3, 4};
    constexpr int off[NG] = {0, 3 * BS, 6 * BS, 7 * BS, 10 * BS};
    const int t = threadIdx.x;
    if (rows == BS) {
        float4 g[NG], p[NG], m[NG], v[NG];
        bool on[NG], vis[NG][4];
#pragma unroll
        for (int k = 0; k < NG; k++) {
            on[k] = false;
            if (t < BS * wid[k] / 4) {
                g[k] = reinterpret_cast<const float4*>(lds_g + off[k])[t];
                if (gout[k]) reinterpret_cast<float4*>(gout[k] + (size_t)row0 * wid[k])[t] = g[k];
                if (A.on) {
                    const int e = 4 * t;
#pragma unroll
                    for (int c = 0; c < 4; c++) vis[k][c] = lds_vis[(e + c) / wid[k]];
                    on[k] = vis[k][0] | vis[k][1] | vis[k][2] | vis[k][3];
                }
            }
            if (on[k]) {
                const size_t base = (size_t)row0 * wid[k];
                p[k] = reinterpret_cast<const float4*>(A.p[gid[k]] + base)[t];
                m[k] = ld_stream(reinterpret_cast<const float4*>(A.m[gid[k]] + base) + t);
                v[k] = ld_stream(reinterpret_cast<const float4*>(A.v[gid[k]] + base) + t);
            }
        }
#pragma unroll
        for (int k = 0; k < NG; k++) {
            if (!on[k]) continue;
            const float lr = A.lr[gid[k]];
            if (vis[k][0]) adam_scalar(p[k].x, g[k].x, m[k].x, v[k].x, lr, A.b1, A.b2, A.eps);
            if (vis[k][1]) adam_scalar(p[k].y, g[k].y, m[k].y, v[k].y, lr, A.b1, A.b2, A.eps);
            if (vis[k][2]) adam_scalar(p[k].z, g[k].z, m[k].z, v[k].z, lr, A.b1, A.b2, A.eps);
            if (vis[k][3]) adam_scalar(p[k].w, g[k].w, m[k].w, v[k].w, lr, A.b1, A.b2, A.eps);
            const size_t base = (size_t)row0 * wid[k];
            st_stream(reinterpret_cast<float4*>(A.p[gid[k]] + base) + t, p[k]);
            st_stream(reinterpret_cast<float4*>(A.m[gid[k]] + base) + t, m[k]);
            st_stream(reinterpret_cast<float4*>(A.v[gid[k]] + base) + t, v[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const size_t base = (size_t)row0 * wid[k];
            for (int i = t; i < rows * wid[k]; i += BS) {
                if (gout[k]) gout[k][base + i] = lds_g[off[k] + i];
                if (A.on && lds_vis[i / wid[k]])
                    adam_scalar(A.p[gid[k]][base + i], lds_g[off[k] + i], A.m[gid[k]][base + i], A.v[gid[k]][base + i], A.lr[gid[k]], A.b1, A.b2, A.eps);
            }
        }
    }
}

// LDS_SH (M == 15, one wave per workgroup): the block's 64 x 45 SH floats are contiguous in memory and are read ONCE, as float4
// columns, by a cooperative pass that does everything that needs them:
//   * q_k = sum_ch sh[k][ch] dRGB[ch] per Gaussian and coefficient (all the direction gradient needs: dL/ddir = sum_k dc_k/ddir q_k) — the
//     three channels of a coefficient are adjacent elements, a thread takes the triples that START in its float4 (the two elements behind
//     it come with an 8-byte load of its neighbour's column: an L1 hit), so every q_k is one fixed-order sum written once;
//   * the gradient element g = c_k(dir) dRGB[ch] (rank-1: never materialised as a row), stored to dL_dsh and / or consumed by Adam on the
//     spot with the parameter value that is already in registers.
// What the pass needs per Gaussian — c_0..c_14 and the clamp-masked dRGB — sits in a 64 x 19 float table.  LDS per wave 8.6 KB (table 4.75,
// q 3.75) against the 11.3 KB of staging whole rows in and gradient rows out: 16 waves per CU instead of 13 (the kernel streams ~100 bytes
// per flop: occupancy IS its memory-level parallelism, profiles/r03p_pbwd_occupancy.log), and features_rest is fetched once, not twice.
#ifndef GS_PBWD_U
#define GS_PBWD_U 4   // float4 columns of features_rest a thread has in flight in the column pass (3 / 6 measured: profiles/r06p_pbwd_unroll_ab.log)
#endif
#ifndef GS_SHT
#define GS_SHT 19
#endif
static constexpr int SHT = GS_SHT;   // table row stride (odd: rows of one column sit in different banks).  18 / 20 / 21 / 23 measured in round 6 (-DGS_SHT=n): see DESIGN section 4

// (rem * 43) >> 7 == rem / 3 for 0 <= rem < 48
__device__ __forceinline__ int div3_small(int rem) { return (rem * 43) >> 7; }

template <int BS>
__device__ __forceinline__ void sh_columns_pass(const PreprocessBwdArgs& a, const float* __restrict__ tab, float* __restrict__ sk,
                                                const uint8_t* __restrict__ lds_vis, const int row0)
{
#pragma clang fp contract(off)   // q_k = (p0 + p1) + p2 of separately rounded products, here AND in the row-by-row path of a partial last block: a
                                 // Gaussian's direction gradient must not depend on which block of the map its row happens to sit in
    constexpr int NE = BS * 45, NV = NE / 4, U = GS_PBWD_U;
    const AdamFusedArgs& A = a.adam;
    const size_t base = (size_t)row0 * 45;
    const float* __restrict__ P = a.shs + base;   // (== A.p[2] + base when the update is on: api.hip checks the aliasing)
    const bool store_g = a.dL_dsh != nullptr;
    for (int i0 = threadIdx.x; i0 < NV; i0 += BS * U) {
        float4 x[U], m[U], v[U];
        float2 nx[U];
        int r0[U], rem0[U];
        bool any[U], upd[U], vis[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + BS * u;
            any[u] = upd[u] = false;
            if (i < NV) {
                const int e = 4 * i;
                r0[u] = e / 45; rem0[u] = e - 45 * r0[u];
                const int r1 = (r0[u] + 1 < BS) ? r0[u] + 1 : r0[u];
                const bool v0 = lds_vis[r0[u]] != 0, v1 = lds_vis[r1] != 0;
#pragma unroll
                for (int c = 0; c < 4; c++) vis[u][c] = (rem0[u] + c < 45) ? v0 : v1;
                any[u] = v0 | v1;   // some element of columns i, i + 1 belongs to a visible Gaussian
                upd[u] = A.on & (vis[u][0] | vis[u][3]);
                x[u] = make_float4(0.f, 0.f, 0.f, 0.f); nx[u] = make_float2(0.f, 0.f);
                if (any[u]) {
                    x[u] = reinterpret_cast<const float4*>(P)[i];
                    if (e + 4 < NE) nx[u] = *reinterpret_cast<const float2*>(P + e + 4);
                }
                if (upd[u]) {
                    m[u] = ld_stream(reinterpret_cast<const float4*>(A.m[2] + base) + i);
                    v[u] = ld_stream(reinterpret_cast<const float4*>(A.v[2] + base) + i);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + BS * u;
            if (i >= NV) continue;
            const float xe[6] = {x[u].x, x[u].y, x[u].z, x[u].w, nx[u].x, nx[u].y};
            float pr[6], g[4];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                int r = r0[u], rem = rem0[u] + c;
                if (rem >= 45) { rem -= 45; r += 1; }
                if (r >= BS) r = BS - 1;   // (elements behind the block: their products are never used)
                const int k = div3_small(rem), ch = rem - 3 * k;
                const float d = tab[r * SHT + 15 + ch];
                pr[c] = xe[c] * d;
                if (c < 4) g[c] = tab[r * SHT + k] * d;
            }
            // the triples that start in this float4: at c0 = (3 - ch(e)) % 3 and, when c0 == 0, at 3 as well
            const int ch0 = rem0[u] - 3 * div3_small(rem0[u]);
            const int c0 = ch0 == 0 ? 0 : 3 - ch0;
            const float t0 = c0 == 0 ? pr[0] : (c0 == 1 ? pr[1] : pr[2]);
            const float t1 = c0 == 0 ? pr[1] : (c0 == 1 ? pr[2] : pr[3]);
            const float t2 = c0 == 0 ? pr[2] : (c0 == 1 ? pr[3] : pr[4]);
            const int f = (4 * i + c0) / 3;   // = 15 r + k of the triple
            sk[f] = (t0 + t1) + t2;
            if (c0 == 0) sk[f + 1] = (pr[3] + pr[4]) + pr[5];
            const float4 g4 = make_float4(g[0], g[1], g[2], g[3]);
            if (store_g) reinterpret_cast<float4*>(a.dL_dsh + base)[i] = g4;
            if (upd[u]) {
                if (vis[u][0]) adam_scalar(x[u].x, g4.x, m[u].x, v[u].x, A.lr[2], A.b1, A.b2, A.eps);
                if (vis[u][1]) adam_scalar(x[u].y, g4.y, m[u].y, v[u].y, A.lr[2], A.b1, A.b2, A.eps);
                if (vis[u][2]) adam_scalar(x[u].z, g4.z, m[u].z, v[u].z, A.lr[2], A.b1, A.b2, A.eps);
                if (vis[u][3]) adam_scalar(x[u].w, g4.w, m[u].w, v[u].w, A.lr[2], A.b1, A.b2, A.eps);
                st_stream(reinterpret_cast<float4*>(A.p[2] + base) + i, x[u]);
                st_stream(reinterpret_cast<float4*>(A.m[2] + base) + i, m[u]);
                st_stream(reinterpret_cast<float4*>(A.v[2] + base) + i, v[u]);
            }
        }
    }
}

// Waves per SIMD the register allocation aims at.  The kernel streams ~100 bytes per flop and its phases (gather, table, column pass, small
// groups) run one after the other inside a wave: the waves in flight ARE its memory-level parallelism.  At 131 VGPRs it ran three waves per
// SIMD; capped at 128 (two spilled outside the column pass) it runs four — 0.50 -> 0.444 ms at 2M / 1080p, same box
// (profiles/r06n_pbwd_occupancy_ab.log); five (96 VGPRs, 44 spilled) gives the gain back.  -DGS_PBWD_WPE=n for A/B runs.
// GS_PBWD_STRICT_CHAIN=1: the per-Gaussian chain (cov2D / cov3D / projection backward, bwd_rest) without contraction, as the reference's kernels are
// built.  Left to the compiler, its contraction choices moved with the occupancy attribute below (round 6: dL_dcov3D / dL_dmean3D / dL_dscale /
// dL_drot changed in their last bits, profiles/r06ag_case20_digests.log).  Pinned, the four gradients are on average 10-30 % closer to the
// reference's kernels (profiles/r06ai_chain_contract_off_errors.log), the kernel needs 124 VGPRs without a spill and is as fast — but the raw-parameter
// path's activation derivatives then differ a little more from LibTorch's autograd of the same activations, and after three Adam steps (sign-like
// first steps) tests/test_shim_gpu.py::test_dropin_renderer_cpp's image is 5.2e-4 from the reference renderer's instead of < 2e-4.  Off by default.
#ifndef GS_PBWD_STRICT_CHAIN
#define GS_PBWD_STRICT_CHAIN 0
#endif
#ifndef GS_PBWD_WPE
#define GS_PBWD_WPE 4
#endif
#define GS_PBWD_OCC __attribute__((amdgpu_waves_per_eu(GS_PBWD_WPE, GS_PBWD_WPE)))
template <bool LDS_SH, int BS, bool CAM>
__global__ __launch_bounds__(BS) GS_PBWD_OCC void preprocess_bwd_kernel(PreprocessBwdArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds_tab[LDS_SH ? BS * SHT : 4];   // {c_0..c_14, dRGB} per Gaussian; later the 14 small gradients
    __shared__ float lds_sk[LDS_SH ? BS * 15 : 4];                                    // q_k per Gaussian
    __shared__ uint8_t lds_vis[BS];
    // the 14 small-group gradients of every Gaussian of the block, group-major (xyz | dc | opacity | scale | rotation), for the
    // cooperative float4 Adam below: per-thread 4-byte accesses at stride 12 / 16 B cost this kernel 0.29 ms of 0.86
    float* const lds_g = lds_tab;  // overlays the table once the column pass is through (14 * BS <= SHT * BS)
    if (a.status[2] != 0u) return;  // capacity overflow in the forward: nothing of this step is valid — no gradients, no Adam
    const int idx = a.row_begin + blockIdx.x * BS + threadIdx.x;
    const int M = a.M;
    const int row0 = a.row_begin + blockIdx.x * BS;
    const int rows = (a.row_end - row0) < BS ? (a.row_end - row0) : BS;
    ShOut so;
    so.x = so.y = so.z = so.dR = so.dG = so.dB = 0.f;
    so.on = false;
    float sg[14];
#pragma unroll
    for (int k = 0; k < 14; k++) sg[k] = 0.f;
    // camera gradient (CAM): 27 sums over the Gaussians — d/dviewmatrix rows 0..2, d/dprojmatrix rows 0, 1, 3, d/dcampos — with the
    // three matrices treated as the independent inputs they are at this boundary (the reference returns no camera gradient at all:
    // rasterizer.cpp:181).  Each wave reduces its 64 Gaussians and writes one row of partials; cam_reduce_kernel adds the rows.
    float cg[CAM ? 27 : 1];
    if constexpr (CAM) {
#pragma unroll
        for (int k = 0; k < 27; k++) cg[k] = 0.f;
    }
    PartialSums ps;
    bool visible = false;
    if (idx < a.row_end) visible = bwd_gather(a, idx, LDS_SH, ps);
    lds_vis[threadIdx.x] = visible ? 1 : 0;
    if (a.vis_out && idx < a.row_end) a.vis_out[idx] = visible ? 1 : 0;                                            // the exchange payload's mask ...
    if (a.campos_out && blockIdx.x == 0 && threadIdx.x < 3) a.campos_out[threadIdx.x] = a.campos[threadIdx.x];     // ... and camera centre
    if constexpr (LDS_SH) {
        // ---- this Gaussian's row of the table (zeros when invisible: its gradient elements and products are then exact zeros)
        float trow[18];
#pragma unroll
        for (int k = 0; k < 18; k++) trow[k] = 0.f;
        if (visible) {
            const uint32_t clamp_bits = __float_as_uint(a.rec[GS_REC_F4 * (size_t)idx + 2].z);
            float dox, doy, doz, x, y, z;
            sh_dir(a.means[3 * idx], a.means[3 * idx + 1], a.means[3 * idx + 2], a.campos, dox, doy, doz, x, y, z);
            float c[15];
            sh_coefs(a.D, x, y, z, c);
#pragma unroll
            for (int k = 0; k < 15; k++) trow[k] = c[k];
            trow[15] = (clamp_bits & 1u) ? 0.f : ps.r; trow[16] = (clamp_bits & 2u) ? 0.f : ps.g; trow[17] = (clamp_bits & 4u) ? 0.f : ps.b;
        }
#pragma unroll
        for (int k = 0; k < 18; k++) lds_tab[threadIdx.x * SHT + k] = trow[k];
        __syncthreads();
        // ---- the column pass over the block's SH parameters (full blocks); the last, partial block goes row by row
        const bool sh_sink = a.dL_dsh || a.adam.on;
        if (rows == BS) {
            sh_columns_pass<BS>(a, lds_tab, lds_sk, lds_vis, row0);
        } else {
            const size_t base = (size_t)row0 * 45;
            if ((int)threadIdx.x < rows) {
#pragma clang fp contract(off)   // (as sh_columns_pass rounds them)
                const float* __restrict__ sh = a.shs + base + 45 * threadIdx.x;
                const float* __restrict__ tr = lds_tab + threadIdx.x * SHT;
#pragma unroll
                for (int k = 0; k < 15; k++) lds_sk[threadIdx.x * 15 + k] = (sh[3 * k] * tr[15] + sh[3 * k + 1] * tr[16]) + sh[3 * k + 2] * tr[17];
            }
            __syncthreads();   // every q_k has been formed from the parameters as they were
            const AdamFusedArgs& A = a.adam;
            if (sh_sink)
                for (int i = threadIdx.x; i < rows * 45; i += BS) {
                    const int r = i / 45, rem = i - 45 * r, k = rem / 3, ch = rem - 3 * k;
                    const float g = lds_tab[r * SHT + k] * lds_tab[r * SHT + 15 + ch];
                    if (a.dL_dsh) a.dL_dsh[base + i] = g;
                    if (A.on && lds_vis[r]) adam_scalar(A.p[2][base + i], g, A.m[2][base + i], A.v[2][base + i], A.lr[2], A.b1, A.b2, A.eps);
                }
        }
        __syncthreads();   // q_k complete; the table is free
        if (visible) bwd_rest<CAM>(a, idx, ps, nullptr, lds_sk + threadIdx.x * 15, so, sg, cg);
    } else {
        const float* sh_row = a.shs ? a.shs + (size_t)3 * M * idx : nullptr;
        if (visible) bwd_rest<CAM>(a, idx, ps, sh_row, nullptr, so, nullptr, cg);
    }
    if constexpr (CAM) {
#pragma unroll
        for (int k = 0; k < 27; k++) {
            float v = cg[k];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            cg[k] = v;
        }
        if ((threadIdx.x & 63) == 0) {
            float* row = a.cam_partials + 32 * ((size_t)blockIdx.x * (BS / 64) + (threadIdx.x >> 6));
#pragma unroll
            for (int k = 0; k < 27; k++) row[k] = cg[k];
        }
    }
    if constexpr (LDS_SH) {
        // ---- small groups: the block's rows of xyz / dc / opacity / scale / rotation are contiguous in memory, so the Adam update runs
        // on float4 columns of those five regions (at most one float4 per thread and group, all fifteen loads issued before the math)
        const int t = threadIdx.x;
        lds_g[3 * t] = sg[0]; lds_g[3 * t + 1] = sg[1]; lds_g[3 * t + 2] = sg[2];
        lds_g[3 * BS + 3 * t] = sg[3]; lds_g[3 * BS + 3 * t + 1] = sg[4]; lds_g[3 * BS + 3 * t + 2] = sg[5];
        lds_g[6 * BS + t] = sg[6];
        lds_g[7 * BS + 3 * t] = sg[7]; lds_g[7 * BS + 3 * t + 1] = sg[8]; lds_g[7 * BS + 3 * t + 2] = sg[9];
        reinterpret_cast<float4*>(lds_g + 10 * BS)[t] = make_float4(sg[10], sg[11], sg[12], sg[13]);
        __syncthreads();
        float* const gout[5] = {a.dL_dmean3D, a.dL_drgb ? a.dL_drgb : a.dL_ddc, a.dL_dopacity, a.dL_dscale, a.dL_drot};
        small_groups_sink<BS>(a.adam, gout, lds_g, lds_vis, row0, rows);
    } else if (idx < a.row_end && M > 0 && (a.dL_dsh || a.adam.on)) {
        // generic row width: per-thread strided rows (dL_dsh zeros when invisible, when shs == NULL, above the active degree)
        const AdamFusedArgs& A = a.adam;
        float c[15];
        sh_coefs(a.D, so.x, so.y, so.z, c);
        const int nk = M < 15 ? M : 15;
        const float dR[3] = {so.dR, so.dG, so.dB};
        const size_t rb = (size_t)3 * M * idx;
        for (int k = 0; k < 3 * M; k++) {
            const float g = (so.on && k < 3 * nk) ? c[k / 3] * dR[k % 3] : 0.f;
            if (a.dL_dsh) a.dL_dsh[rb + k] = g;
            if (A.on && lds_vis[threadIdx.x]) adam_scalar(A.p[2][rb + k], g, A.m[2][rb + k], A.v[2][rb + k], A.lr[2], A.b1, A.b2, A.eps);
        }
    }
}

// Zero rows of an invisible Gaussian / the segmented sum of a visible one's per-instance rows.  Returns the visibility.
// staged: the five parameter-gradient rows leave through the block's staged float4 stores (their zeros too).
__device__ __forceinline__ bool bwd_gather(const PreprocessBwdArgs& a, const int idx, const bool staged, PartialSums& ps)
{
    ps.mx = ps.my = ps.cx = ps.cy = ps.cw = ps.op = ps.r = ps.g = ps.b = 0.f;
    const bool visible = a.radii[idx] > 0;

    if (!visible) {
        if (a.dL_dmean2D) { a.dL_dmean2D[3 * idx] = 0; a.dL_dmean2D[3 * idx + 1] = 0; a.dL_dmean2D[3 * idx + 2] = 0; }
        if (a.dL_dconic) reinterpret_cast<float4*>(a.dL_dconic)[idx] = make_float4(0, 0, 0, 0);
        if (a.dL_dcolor) { a.dL_dcolor[3 * idx] = 0; a.dL_dcolor[3 * idx + 1] = 0; a.dL_dcolor[3 * idx + 2] = 0; }
        if (a.dL_dcov3D)
            for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * idx + k] = 0;
        if (staged) return false;
        if (a.dL_dopacity) a.dL_dopacity[idx] = 0;
        if (a.dL_dmean3D) { a.dL_dmean3D[3 * idx] = 0; a.dL_dmean3D[3 * idx + 1] = 0; a.dL_dmean3D[3 * idx + 2] = 0; }
        if (a.dL_ddc) { a.dL_ddc[3 * idx] = 0; a.dL_ddc[3 * idx + 1] = 0; a.dL_ddc[3 * idx + 2] = 0; }
        if (a.dL_drgb) { a.dL_drgb[3 * idx] = 0; a.dL_drgb[3 * idx + 1] = 0; a.dL_drgb[3 * idx + 2] = 0; }
        if (a.dL_dscale) { a.dL_dscale[3 * idx] = 0; a.dL_dscale[3 * idx + 1] = 0; a.dL_dscale[3 * idx + 2] = 0; }
        if (a.dL_drot) reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(0, 0, 0, 0);
        return false;
    }

    // ---- segmented sum of the per-instance partial gradients (ascending tile order) ----
    // One 36-byte row per instance, written by render_bwd for the instances of live buckets only: a set flag byte means the instance lies
    // behind its tile's last contributor — nine exact zeros that were never written and are not fetched here (58 % of the instances of the
    // 2M / 1080p scene).  Four rows in flight per lane (clamped loads, masked adds): same summation order as one at a time.
    float s_mx = 0, s_my = 0, s_cx = 0, s_cy = 0, s_cw = 0, s_op = 0, s_r = 0, s_g = 0, s_b = 0;
    {
        const uint32_t u0 = a.gauss_start[idx];
        const uint32_t u1 = u0 + a.tiles_touched[idx];
        constexpr int UP = 4;
        for (uint32_t u = u0; u < u1; u += UP) {
            // the four flag bytes as one (unaligned) dword: bytes past u1 - 1 belong to the next Gaussian (or to the 256 bytes of slack
            // behind the array) and are masked off
            uint32_t flags;
            __builtin_memcpy(&flags, a.dead + u, 4);
            bool live[UP];
#pragma unroll
            for (int j = 0; j < UP; j++) live[j] = (u + j < u1) & (((flags >> (8 * j)) & 0xffu) == 0u);
            gs_v4f_u q0[UP], q1[UP];
            float q2[UP];
#pragma unroll
            for (int j = 0; j < UP; j++) {
                if (live[j]) {
                    const float* p = a.partials + GS_PROW * (size_t)(u + j);
                    q0[j] = *reinterpret_cast<const gs_v4f_u*>(p);
                    q1[j] = *reinterpret_cast<const gs_v4f_u*>(p + 4);
                    q2[j] = p[8];
                }
            }
#pragma unroll
            for (int j = 0; j < UP; j++) {
                if (live[j]) {
                    s_mx += q0[j].x; s_my += q0[j].y; s_cx += q0[j].z; s_cy += q0[j].w;
                    s_cw += q1[j].x; s_op += q1[j].y; s_r += q1[j].z; s_g += q1[j].w;
                    s_b += q2[j];
                }
            }
        }
    }
    if (a.dL_dmean2D) { a.dL_dmean2D[3 * idx] = s_mx; a.dL_dmean2D[3 * idx + 1] = s_my; a.dL_dmean2D[3 * idx + 2] = 0; }
    if (a.dL_dconic) reinterpret_cast<float4*>(a.dL_dconic)[idx] = make_float4(s_cx, s_cy, 0.f, s_cw);
    if (a.dL_dcolor) { a.dL_dcolor[3 * idx] = s_r; a.dL_dcolor[3 * idx + 1] = s_g; a.dL_dcolor[3 * idx + 2] = s_b; }
    ps.mx = s_mx; ps.my = s_my; ps.cx = s_cx; ps.cy = s_cy; ps.cw = s_cw; ps.op = s_op; ps.r = s_r; ps.g = s_g; ps.b = s_b;
    return true;
}

// Everything behind the sums for a VISIBLE Gaussian.  The SH block needs, per coefficient k, only q_k = sum_ch sh[k][ch] dRGB[ch]
// (dL/ddir = sum_k dc_k/ddir q_k): sk_row hands them over when the block computed them cooperatively, else they are formed from sh_row.
template <bool CAM>
__device__ __forceinline__ void bwd_rest(const PreprocessBwdArgs& a, const int idx, const PartialSums& ps, const float* sh_row, const float* sk_row,
                                         ShOut& so, float* sg, float* cg)
{
#if GS_PBWD_STRICT_CHAIN
#pragma clang fp contract(off)   // the per-Gaussian chain rounds every product and sum on its own, as the reference's kernels are built (oracle/ref_build: -ffp-contract=off)
#endif
    const float s_mx = ps.mx, s_my = ps.my, s_cx = ps.cx, s_cy = ps.cy, s_cw = ps.cw, s_op = ps.op, s_r = ps.r, s_g = ps.g, s_b = ps.b;
    const float* __restrict__ V = a.view;
    const float* __restrict__ Pm = a.proj;
    const float mx3 = a.means[3 * idx], my3 = a.means[3 * idx + 1], mz3 = a.means[3 * idx + 2];

    // ---- Sigma_3D from scale / rotation (forward.cu:120-149) ----
    float qr = a.rots[4 * idx], qx = a.rots[4 * idx + 1], qy = a.rots[4 * idx + 2], qz = a.rots[4 * idx + 3];
    const float4 qraw = make_float4(qr, qx, qy, qz);
    if (a.raw) {
        const float nrm = fmaxf(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f);
        qr = qr / nrm; qx = qx / nrm; qy = qy / nrm; qz = qz / nrm;
    }
    float Rm[3][3];
    Rm[0][0] = 1.f - 2.f * (qy * qy + qz * qz); Rm[0][1] = 2.f * (qx * qy - qr * qz); Rm[0][2] = 2.f * (qx * qz + qr * qy);
    Rm[1][0] = 2.f * (qx * qy + qr * qz); Rm[1][1] = 1.f - 2.f * (qx * qx + qz * qz); Rm[1][2] = 2.f * (qy * qz - qr * qx);
    Rm[2][0] = 2.f * (qx * qz - qr * qy); Rm[2][1] = 2.f * (qy * qz + qr * qx); Rm[2][2] = 1.f - 2.f * (qx * qx + qy * qy);
    float sc[3] = {a.scales[3 * idx], a.scales[3 * idx + 1], a.scales[3 * idx + 2]};
    if (a.raw) { sc[0] = expf(sc[0]); sc[1] = expf(sc[1]); sc[2] = expf(sc[2]); }
    const float s[3] = {a.scale_modifier * sc[0], a.scale_modifier * sc[1], a.scale_modifier * sc[2]};
    float Mk[3][3];  // Mk[k][c] = s_k * Rm[c][k]  (= glm M[c][k])
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) Mk[k][c] = s[k] * Rm[c][k];
#define SIG(i, j) (Mk[0][i] * Mk[0][j] + Mk[1][i] * Mk[1][j] + Mk[2][i] * Mk[2][j])
    const float c6[6] = {SIG(0, 0), SIG(0, 1), SIG(0, 2), SIG(1, 1), SIG(1, 2), SIG(2, 2)};
#undef SIG

    // ---- computeCov2DCUDA (backward.cu:138-255) ----
    float t0 = V[0] * mx3 + V[4] * my3 + V[8] * mz3 + V[12];
    float t1 = V[1] * mx3 + V[5] * my3 + V[9] * mz3 + V[13];
    const float t2 = V[2] * mx3 + V[6] * my3 + V[10] * mz3 + V[14];
    const float tx_over_tz = t0 / t2, ty_over_tz = t1 / t2;
    t0 = fminf(a.limx_pos, fmaxf(a.limx_neg, tx_over_tz)) * t2;
    t1 = fminf(a.limy_pos, fmaxf(a.limy_neg, ty_over_tz)) * t2;
    const float keep_x = (tx_over_tz < a.limx_neg || tx_over_tz > a.limx_pos) ? 0.f : 1.f;
    const float keep_y = (ty_over_tz < a.limy_neg || ty_over_tz > a.limy_pos) ? 0.f : 1.f;
    const float fx = a.focal_x, fy = a.focal_y;
    const float J00 = fx / t2, J02 = -(fx * t0) / (t2 * t2), J11 = fy / t2, J12 = -(fy * t1) / (t2 * t2);
    float T0[3], T1[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        T0[i] = V[4 * i + 0] * J00 + V[4 * i + 2] * J02;
        T1[i] = V[4 * i + 1] * J11 + V[4 * i + 2] * J12;
    }
    const float Vr[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float VT0[3], VT1[3];  // Vr[k] . T0, Vr[k] . T1
#pragma unroll
    for (int k = 0; k < 3; k++) {
        VT0[k] = T0[0] * Vr[k][0] + T0[1] * Vr[k][1] + T0[2] * Vr[k][2];
        VT1[k] = T1[0] * Vr[k][0] + T1[1] * Vr[k][1] + T1[2] * Vr[k][2];
    }
    const float ca = (VT0[0] * T0[0] + VT0[1] * T0[1] + VT0[2] * T0[2]) + 0.3f;
    const float cb = VT1[0] * T0[0] + VT1[1] * T0[1] + VT1[2] * T0[2];
    const float cc = (VT1[0] * T1[0] + VT1[1] * T1[1] + VT1[2] * T1[2]) + 0.3f;
    const float denom = ca * cc - cb * cb;
    const float inv_det2 = 1.0f / ((denom * denom) + 0.0000001f);
    float g_ca = 0, g_cb = 0, g_cc = 0;
    float dcv[6] = {0, 0, 0, 0, 0, 0};
    if (inv_det2 != 0) {
        g_ca = inv_det2 * (-cc * cc * s_cx + 2 * cb * cc * s_cy + (denom - ca * cc) * s_cw);
        g_cc = inv_det2 * (-ca * ca * s_cw + 2 * ca * cb * s_cy + (denom - ca * cc) * s_cx);
        g_cb = inv_det2 * 2 * (cb * cc * s_cx - (denom + 2 * cb * cb) * s_cy + ca * cb * s_cw);
        dcv[0] = (T0[0] * T0[0] * g_ca + T0[0] * T1[0] * g_cb + T1[0] * T1[0] * g_cc);
        dcv[3] = (T0[1] * T0[1] * g_ca + T0[1] * T1[1] * g_cb + T1[1] * T1[1] * g_cc);
        dcv[5] = (T0[2] * T0[2] * g_ca + T0[2] * T1[2] * g_cb + T1[2] * T1[2] * g_cc);
        dcv[1] = 2 * T0[0] * T0[1] * g_ca + (T0[0] * T1[1] + T0[1] * T1[0]) * g_cb + 2 * T1[0] * T1[1] * g_cc;
        dcv[2] = 2 * T0[0] * T0[2] * g_ca + (T0[0] * T1[2] + T0[2] * T1[0]) * g_cb + 2 * T1[0] * T1[2] * g_cc;
        dcv[4] = 2 * T0[2] * T0[1] * g_ca + (T0[1] * T1[2] + T0[2] * T1[1]) * g_cb + 2 * T1[1] * T1[2] * g_cc;
    }
    if (a.dL_dcov3D)
#pragma unroll
        for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * idx + k] = dcv[k];

    const float gT0x = 2 * VT0[0] * g_ca + VT1[0] * g_cb;
    const float gT0y = 2 * VT0[1] * g_ca + VT1[1] * g_cb;
    const float gT0z = 2 * VT0[2] * g_ca + VT1[2] * g_cb;
    const float gT1x = 2 * VT1[0] * g_cc + VT0[0] * g_cb;
    const float gT1y = 2 * VT1[1] * g_cc + VT0[1] * g_cb;
    const float gT1z = 2 * VT1[2] * g_cc + VT0[2] * g_cb;
    const float gJ00 = V[0] * gT0x + V[4] * gT0y + V[8] * gT0z;
    const float gJ02 = V[2] * gT0x + V[6] * gT0y + V[10] * gT0z;
    const float gJ11 = V[1] * gT1x + V[5] * gT1y + V[9] * gT1z;
    const float gJ12 = V[2] * gT1x + V[6] * gT1y + V[10] * gT1z;
    const float tz = 1.f / t2, tz2 = tz * tz, tz3 = tz2 * tz;
    const float g_tx = keep_x * -fx * tz2 * gJ02;
    const float g_ty = keep_y * -fy * tz2 * gJ12;
    const float g_tz = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + (2 * fx * t0) * tz3 * gJ02 + (2 * fy * t1) * tz3 * gJ12;
    if constexpr (CAM) {
        const float pc[4] = {mx3, my3, mz3, 1.0f};
        // exact through the clamp (the parameter gradients keep the reference's convention, backward.cu:225-233: a clamped t.x = lim t.z
        // still moves with t.z; for the camera that term is kept: + lim * the unmasked dL/dt.x — oracle/gs_oracle.c, same lines)
        const float ex = (1.0f - keep_x) * (t0 * tz) * (-fx * tz2 * gJ02);
        const float ey = (1.0f - keep_y) * (t1 * tz) * (-fy * tz2 * gJ12);
        const float gt[3] = {g_tx, g_ty, g_tz + ex + ey};
        // t = V [p, 1]: d/dV[4c + r] = dL/dt_r * p_c
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 3; r++) cg[3 * c + r] += gt[r] * pc[c];
        // W = rotation part of V inside T = W J (T0[i] = V[4i] J00 + V[4i+2] J02, T1[i] = V[4i+1] J11 + V[4i+2] J12)
        const float gT0[3] = {gT0x, gT0y, gT0z}, gT1[3] = {gT1x, gT1y, gT1z};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            cg[3 * i + 0] += gT0[i] * J00;
            cg[3 * i + 1] += gT1[i] * J11;
            cg[3 * i + 2] += gT0[i] * J02 + gT1[i] * J12;
        }
    }
    float dmean[3];
    dmean[0] = V[0] * g_tx + V[1] * g_ty + V[2] * g_tz;
    dmean[1] = V[4] * g_tx + V[5] * g_ty + V[6] * g_tz;
    dmean[2] = V[8] * g_tx + V[9] * g_ty + V[10] * g_tz;

    // ---- mean2D -> mean3D through the projection (backward.cu:339-350) ----
    {
        const float hw = Pm[3] * mx3 + Pm[7] * my3 + Pm[11] * mz3 + Pm[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        const float mul1 = (Pm[0] * mx3 + Pm[4] * my3 + Pm[8] * mz3 + Pm[12]) * pw * pw;
        const float mul2 = (Pm[1] * mx3 + Pm[5] * my3 + Pm[9] * mz3 + Pm[13]) * pw * pw;
        dmean[0] += (Pm[0] * pw - Pm[3] * mul1) * s_mx + (Pm[1] * pw - Pm[3] * mul2) * s_my;
        dmean[1] += (Pm[4] * pw - Pm[7] * mul1) * s_mx + (Pm[5] * pw - Pm[7] * mul2) * s_my;
        dmean[2] += (Pm[8] * pw - Pm[11] * mul1) * s_mx + (Pm[9] * pw - Pm[11] * mul2) * s_my;
        if constexpr (CAM) {
            // p_proj = (hx, hy) * pw, pw = 1 / (hw + 1e-7): d/dPm[4c + r] = dL/dh_r * p_c for r = 0 (x), 1 (y), 3 (w)
            const float pc[4] = {mx3, my3, mz3, 1.0f};
            const float ghx = s_mx * pw, ghy = s_my * pw, ghw = -(mul1 * s_mx + mul2 * s_my);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                cg[12 + 3 * c + 0] += ghx * pc[c];
                cg[12 + 3 * c + 1] += ghy * pc[c];
                cg[12 + 3 * c + 2] += ghw * pc[c];
            }
        }
    }

    // ---- SH backward (backward.cu:27-136); skipped entirely when shs == NULL (backward.cu:352) ----
    float ddc[3] = {0.f, 0.f, 0.f};
    if (a.shs) {
        const uint32_t clamp_bits = __float_as_uint(a.rec[GS_REC_F4 * (size_t)idx + 2].z);
        float dox, doy, doz, x, y, z;
        sh_dir(mx3, my3, mz3, a.campos, dox, doy, doz, x, y, z);
        const float* __restrict__ sh = sh_row;
        const float dRGB[3] = {(clamp_bits & 1u) ? 0.f : s_r, (clamp_bits & 2u) ? 0.f : s_g, (clamp_bits & 4u) ? 0.f : s_b};
        so.x = x; so.y = y; so.z = z; so.dR = dRGB[0]; so.dG = dRGB[1]; so.dB = dRGB[2]; so.on = true;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) ddc[ch] = a.dL_drgb ? dRGB[ch] : SHC0 * dRGB[ch];   // (dL_drgb: the dc slot of the outputs carries the masked colour gradient itself)
        float q[15];
#pragma unroll
        for (int k = 0; k < 15; k++)
            q[k] = sk_row ? sk_row[k] : (k < a.M ? ((sh[3 * k] * dRGB[0] + sh[3 * k + 1] * dRGB[1]) + sh[3 * k + 2] * dRGB[2]) : 0.f);
        float ddir[3] = {0.f, 0.f, 0.f};
        if (a.D > 0) {
            float ddx = -SHC1 * q[2], ddy = -SHC1 * q[0], ddz = SHC1 * q[1];
            if (a.D > 1) {
                ddx += b_SH_C2[0] * y * q[3] + b_SH_C2[2] * 2.f * -x * q[5] + b_SH_C2[3] * z * q[6] + b_SH_C2[4] * 2.f * x * q[7];
                ddy += b_SH_C2[0] * x * q[3] + b_SH_C2[1] * z * q[4] + b_SH_C2[2] * 2.f * -y * q[5] + b_SH_C2[4] * 2.f * -y * q[7];
                ddz += b_SH_C2[1] * y * q[4] + b_SH_C2[2] * 2.f * 2.f * z * q[5] + b_SH_C2[3] * x * q[6];
                if (a.D > 2) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    ddx += (b_SH_C3[0] * q[8] * 3.f * 2.f * xy + b_SH_C3[1] * q[9] * yz + b_SH_C3[2] * q[10] * -2.f * xy +
                            b_SH_C3[3] * q[11] * -3.f * 2.f * xz + b_SH_C3[4] * q[12] * (-3.f * xx + 4.f * zz - yy) +
                            b_SH_C3[5] * q[13] * 2.f * xz + b_SH_C3[6] * q[14] * 3.f * (xx - yy));
                    ddy += (b_SH_C3[0] * q[8] * 3.f * (xx - yy) + b_SH_C3[1] * q[9] * xz +
                            b_SH_C3[2] * q[10] * (-3.f * yy + 4.f * zz - xx) + b_SH_C3[3] * q[11] * -3.f * 2.f * yz +
                            b_SH_C3[4] * q[12] * -2.f * xy + b_SH_C3[5] * q[13] * -2.f * yz + b_SH_C3[6] * q[14] * -3.f * 2.f * xy);
                    ddz += (b_SH_C3[1] * q[9] * xy + b_SH_C3[2] * q[10] * 4.f * 2.f * yz +
                            b_SH_C3[3] * q[11] * 3.f * (2.f * zz - xx - yy) + b_SH_C3[4] * q[12] * 4.f * 2.f * xz +
                            b_SH_C3[5] * q[13] * (xx - yy));
                }
            }
            ddir[0] = ddx; ddir[1] = ddy; ddir[2] = ddz;
        }
        // dnormvdv (auxiliary.h:119-129)
        const float sum2 = dox * dox + doy * doy + doz * doz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        const float sd0 = ((+sum2 - dox * dox) * ddir[0] - doy * dox * ddir[1] - doz * dox * ddir[2]) * invsum32;
        const float sd1 = (-dox * doy * ddir[0] + (sum2 - doy * doy) * ddir[1] - doz * doy * ddir[2]) * invsum32;
        const float sd2 = (-dox * doz * ddir[0] - doy * doz * ddir[1] + (sum2 - doz * doz) * ddir[2]) * invsum32;
        dmean[0] += sd0; dmean[1] += sd1; dmean[2] += sd2;
        if constexpr (CAM) { cg[24] -= sd0; cg[25] -= sd1; cg[26] -= sd2; }  // dir = p - campos
    }

    // ---- Sigma_3D -> scale, quaternion (backward.cu:257-310) ----
    // dL/dSigma symmetrised (1/2 on off-diagonals); dM[c][r] = 2 * sum_k M[k][r] * dS[c][k]  (glm column-major)
    const float dS[3][3] = {{dcv[0], 0.5f * dcv[1], 0.5f * dcv[2]}, {0.5f * dcv[1], dcv[3], 0.5f * dcv[4]}, {0.5f * dcv[2], 0.5f * dcv[4], dcv[5]}};
    float dMt[3][3];  // dMt[c][r] = dM[r][c];  glm M[c][r] = s_r * Rm[c][r] = Mk[r][c]
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            dMt[c][r] = 2.f * Mk[c][0] * dS[r][0] + 2.f * Mk[c][1] * dS[r][1] + 2.f * Mk[c][2] * dS[r][2];
    // dM_glm[rr][cc] = sum_k (2M)_glm[k][cc] * dS[rr][k] = sum_k 2*Mk[cc][k]*dS[rr][k]; dMt[c][r] = dM_glm[r][c] -> cc = c, rr = r. ok
    float dscale[3];
#pragma unroll
    for (int k = 0; k < 3; k++) dscale[k] = Rm[0][k] * dMt[k][0] + Rm[1][k] * dMt[k][1] + Rm[2][k] * dMt[k][2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int r = 0; r < 3; r++) dMt[k][r] *= s[k];
#define D_(c, r) dMt[c][r]
    float4 dq;
    dq.x = 2 * qz * (D_(0, 1) - D_(1, 0)) + 2 * qy * (D_(2, 0) - D_(0, 2)) + 2 * qx * (D_(1, 2) - D_(2, 1));
    dq.y = 2 * qy * (D_(0, 1) + D_(1, 0)) + 2 * qz * (D_(2, 0) + D_(0, 2)) + 2 * qr * (D_(1, 2) - D_(2, 1)) - 4 * qx * (D_(2, 2) + D_(1, 1));
    dq.z = 2 * qx * (D_(0, 1) + D_(1, 0)) + 2 * qr * (D_(2, 0) - D_(0, 2)) + 2 * qz * (D_(1, 2) + D_(2, 1)) - 4 * qy * (D_(2, 2) + D_(0, 0));
    dq.w = 2 * qr * (D_(0, 1) - D_(1, 0)) + 2 * qx * (D_(2, 0) + D_(0, 2)) + 2 * qy * (D_(1, 2) + D_(2, 1)) - 4 * qz * (D_(1, 1) + D_(0, 0));
#undef D_
    if (a.lambda_erank > 0) {  // backward.cu:358-375
        const float s1s1 = sc[0] * sc[0], s2s2 = sc[1] * sc[1], s3s3 = sc[2] * sc[2];
        const float sum = s1s1 + s2s2 + s3s3;
        const float q1 = sc[0] / sum, q2 = sc[1] / sum, q3 = sc[2] / sum;
        const float erank = expf(-q1 * logf(q1) - q2 * logf(q2) - q3 * logf(q3));
        if (-log((double)erank - 1 + 1e-5) > 0) {
            const float f = (float)((double)erank / ((double)erank - 1 + 1e-5));
            const float d1 = f * (-logf(q1) - 1), d2 = f * (-logf(q2) - 1), d3 = f * (-logf(q3) - 1);
            const float le = a.lambda_erank * 2.f / (sum * sum);
            dscale[0] += le * sc[0] * (d1 * (s2s2 + s3s3) - d2 * s2s2 - d3 * s3s3);
            dscale[1] += le * sc[1] * (-d1 * s1s1 + d2 * (s1s1 + s3s3) - d3 * s3s3);
            dscale[2] += le * sc[2] * (-d1 * s1s1 - d2 * s2s2 + d3 * (s1s1 + s2s2));
        }
        dscale[2] += 1;
    }
    float g_op = s_op;
    if (a.raw) {
        // chain through the activations, as LibTorch autograd does outside the reference's kernels (gaussian.cpp:147-175):
        // d exp = s; d sigmoid = o (1 - o); d normalize = dnormvdv (auxiliary.h:131-143)
        dscale[0] *= sc[0]; dscale[1] *= sc[1]; dscale[2] *= sc[2];
        const float o = a.rec[GS_REC_F4 * (size_t)idx + 1].y;
        g_op = s_op * o * (1.0f - o);
        const float sum2 = qraw.x * qraw.x + qraw.y * qraw.y + qraw.z * qraw.z + qraw.w * qraw.w;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        const float vd0 = qraw.x * dq.x, vd1 = qraw.y * dq.y, vd2 = qraw.z * dq.z, vd3 = qraw.w * dq.w;
        const float vds = vd0 + vd1 + vd2 + vd3;
        float4 dr;
        if (sqrtf(sum2) < 1e-12f) {
            // F::normalize divides by max(||q||, eps): below eps the divisor is the constant eps (its clamp_min passes no gradient), so the
            // Jacobian is I / eps; the closed form below would be 0 * inf there
            dr = make_float4(dq.x * 1e12f, dq.y * 1e12f, dq.z * 1e12f, dq.w * 1e12f);
        } else {
            dr.x = ((sum2 - qraw.x * qraw.x) * dq.x - qraw.x * (vds - vd0)) * invsum32;
            dr.y = ((sum2 - qraw.y * qraw.y) * dq.y - qraw.y * (vds - vd1)) * invsum32;
            dr.z = ((sum2 - qraw.z * qraw.z) * dq.z - qraw.z * (vds - vd2)) * invsum32;
            dr.w = ((sum2 - qraw.w * qraw.w) * dq.w - qraw.w * (vds - vd3)) * invsum32;
        }
        dq = dr;
    }
    // ---- sinks: gradient tensors (each optional) and / or the in-place Adam update of this Gaussian's 14 small scalars
    if (sg) {  // staged: the block stores / updates these rows with float4 columns (small_groups_sink)
        sg[0] = dmean[0]; sg[1] = dmean[1]; sg[2] = dmean[2];
        sg[3] = ddc[0]; sg[4] = ddc[1]; sg[5] = ddc[2];
        sg[6] = g_op;
        sg[7] = dscale[0]; sg[8] = dscale[1]; sg[9] = dscale[2];
        sg[10] = dq.x; sg[11] = dq.y; sg[12] = dq.z; sg[13] = dq.w;
        return;
    }
    if (a.dL_dopacity) a.dL_dopacity[idx] = g_op;
    if (a.dL_dmean3D) { a.dL_dmean3D[3 * idx] = dmean[0]; a.dL_dmean3D[3 * idx + 1] = dmean[1]; a.dL_dmean3D[3 * idx + 2] = dmean[2]; }
    if (a.dL_ddc) { a.dL_ddc[3 * idx] = ddc[0]; a.dL_ddc[3 * idx + 1] = ddc[1]; a.dL_ddc[3 * idx + 2] = ddc[2]; }
    if (a.dL_drgb) { a.dL_drgb[3 * idx] = ddc[0]; a.dL_drgb[3 * idx + 1] = ddc[1]; a.dL_drgb[3 * idx + 2] = ddc[2]; }
    if (a.dL_dscale) { a.dL_dscale[3 * idx] = dscale[0]; a.dL_dscale[3 * idx + 1] = dscale[1]; a.dL_dscale[3 * idx + 2] = dscale[2]; }
    if (a.dL_drot) reinterpret_cast<float4*>(a.dL_drot)[idx] = dq;
    if (a.adam.on) {
        const AdamFusedArgs& A = a.adam;
        const float dqv[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            adam_scalar(A.p[0][3 * idx + k], dmean[k], A.m[0][3 * idx + k], A.v[0][3 * idx + k], A.lr[0], A.b1, A.b2, A.eps);
            adam_scalar(A.p[1][3 * idx + k], ddc[k], A.m[1][3 * idx + k], A.v[1][3 * idx + k], A.lr[1], A.b1, A.b2, A.eps);
            adam_scalar(A.p[4][3 * idx + k], dscale[k], A.m[4][3 * idx + k], A.v[4][3 * idx + k], A.lr[4], A.b1, A.b2, A.eps);
        }
        adam_scalar(A.p[3][idx], g_op, A.m[3][idx], A.v[3][idx], A.lr[3], A.b1, A.b2, A.eps);
#pragma unroll
        for (int k = 0; k < 4; k++)
            adam_scalar(A.p[5][4 * idx + k], dqv[k], A.m[5][4 * idx + k], A.v[5][4 * idx + k], A.lr[5], A.b1, A.b2, A.eps);
    }
}

// rows of per-wave partials -> the 35 outputs, fixed summation order (bit-reproducible).  grid = 27 blocks, one per term.
__global__ __launch_bounds__(256) void cam_reduce_kernel(size_t rows, const float* __restrict__ partials, float* __restrict__ out)
{
    __shared__ float red[256];
    const int k = blockIdx.x;
    float s = 0.f;
    for (size_t r = threadIdx.x; r < rows; r += 256) s += partials[32 * r + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // term k -> element: view (k < 12): [4c + r], r = k % 3, c = k / 3; proj (12 <= k < 24): rows 0, 1, 3; campos (k >= 24)
        int dst;
        if (k < 12) dst = 4 * (k / 3) + (k % 3);
        else if (k < 24) { const int kk = k - 12; const int r = kk % 3; dst = 16 + 4 * (kk / 3) + (r == 2 ? 3 : r); }
        else dst = 32 + (k - 24);
        out[dst] = red[0];
    }
}

int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s)
{
    const bool cam = a.cam_partials != nullptr;
    const int nrows = a.row_end - a.row_begin;
    if (nrows <= 0) return GSLIC_OK;
    if (cam) GS_HIP(hipMemsetAsync(a.cam_out, 0, 35 * sizeof(float), s));  // row 3 of the view matrix, row 2 of the projection: untouched terms
    if (a.M == 15 && a.shs && (a.dL_dsh || a.adam.on || a.dL_drgb)) {
        // one wave per workgroup: 64 Gaussians' SH rows (11.25 KiB) + small-group gradients in LDS, the phase barriers are wave-level,
        // and ten workgroups per CU sit in different phases (measured 0.72 ms against 0.75 at 128 and 0.88 at 256 threads)
        if (cam) GS_LAUNCH(K_PREPROCESS_BWD, (preprocess_bwd_kernel<true, 64, true>), dim3(div_up(nrows, 64)), dim3(64), 0, s, a);
        else GS_LAUNCH(K_PREPROCESS_BWD, (preprocess_bwd_kernel<true, 64, false>), dim3(div_up(nrows, 64)), dim3(64), 0, s, a);
    } else {
        if (cam) GS_LAUNCH(K_PREPROCESS_BWD, (preprocess_bwd_kernel<false, 256, true>), dim3(div_up(nrows, 256)), dim3(256), 0, s, a);
        else GS_LAUNCH(K_PREPROCESS_BWD, (preprocess_bwd_kernel<false, 256, false>), dim3(div_up(nrows, 256)), dim3(256), 0, s, a);
    }
    if (cam) {
        const size_t rows = (a.M == 15 && a.shs && (a.dL_dsh || a.adam.on || a.dL_drgb)) ? (size_t)div_up(a.P, 64) : (size_t)div_up(a.P, 256) * 4;
        GS_LAUNCH(K_PREPROCESS_BWD, cam_reduce_kernel, dim3(27), dim3(256), 0, s, rows, (const float*)a.cam_partials, a.cam_out);
    }
    return GSLIC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The SH backward is linear in the clamp-masked colour gradient: dL_ddc = SH_C0 * dRGB, dL_dsh[k] = c_k(dir) * dRGB (backward.cu:27-136) with
// c_k depending only on the view direction normalize(p - campos).  So the N > 1 exchange does not have to all-reduce the 48 floats of
// dL_ddc / dL_dsh per Gaussian: every rank all-gathers the views' 3-float dRGB and rebuilds the summed rows here — per view the very
// products the backward forms (sh_dir / sh_coefs shared, multiply then add, no contraction), summed in view order.
// One wave per 64 Gaussians; the rows leave through LDS as contiguous runs (M == 15) like the backward's own dL_dsh rows.
static constexpr int SGR = 32;
template <bool ADAM>
__global__ __launch_bounds__(64) void sh_grad_from_rgb_kernel(ShGradFromRgbArgs a)
{
#pragma clang fp contract(off)
    // SGR = 32 Gaussians per one-wave workgroup: lanes 0..31 rebuild a row each, then all 64 lanes stream the block's moments.  With 64 rows
    // the staging buffer was 12 KB and the kernel ran at 12 waves per CU; it lives on its occupancy like preprocess_bwd (6 KB: 20 waves).
    __shared__ __attribute__((aligned(16))) float lds_rows[SGR * 48];
    const int t = threadIdx.x;
    const int row0 = blockIdx.x * SGR;
    const int idx = t < SGR ? row0 + t : a.P;   // (lanes 32..63 own no Gaussian)
    const int rows = (a.P - row0) < SGR ? (a.P - row0) : SGR;
    float acc[45], adc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 45; k++) acc[k] = 0.f;
    if (idx < a.P) {
        const float px = a.means3D[3 * (size_t)idx], py = a.means3D[3 * (size_t)idx + 1], pz = a.means3D[3 * (size_t)idx + 2];
        for (int v = 0; v < a.n_views; v++) {
            const float* g = a.rgb_all + (size_t)v * a.rgb_stride + (size_t)idx * 3;
            float d[3] = {g[0], g[1], g[2]};
            if (d[0] == 0.f && d[1] == 0.f && d[2] == 0.f) continue;   // invisible in this view (or all channels clamped): adds exact zeros
            float dc_v[3];
            if (a.input_is_ddc) {   // the view's dL_ddc = SH_C0 * dRGB was shipped (hosts that only have the reference's gradient tensors)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) { dc_v[ch] = d[ch]; d[ch] = d[ch] * (1.0f / SHC0); }
            } else {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) dc_v[ch] = SHC0 * d[ch];
            }
            float dox, doy, doz, x, y, z, c[15];
            sh_dir(px, py, pz, a.campos_all + (size_t)v * a.campos_stride, dox, doy, doz, x, y, z);
            sh_coefs(a.D, x, y, z, c);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) adc[ch] = adc[ch] + dc_v[ch];
#pragma unroll
            for (int k = 0; k < 15; k++) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    const float prod = c[k] * d[ch];
                    acc[3 * k + ch] = acc[3 * k + ch] + prod;
                }
            }
        }
    }
    // dc rows [32 x 3] then rest rows [32 x 45] through LDS: contiguous float runs instead of 48 strided 4-byte stores per thread
    if (t < SGR) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) lds_rows[3 * t + ch] = adc[ch];
#pragma unroll
        for (int k = 0; k < 45; k++) lds_rows[3 * SGR + 45 * t + k] = acc[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (a.dL_ddc)
        for (int i = t; i < rows * 3; i += 64) a.dL_ddc[(size_t)row0 * 3 + i] = lds_rows[i];
    if (a.dL_dsh) {
        if (a.M == 15) {
            for (int i = t; i < rows * 45; i += 64) a.dL_dsh[(size_t)row0 * 45 + i] = lds_rows[3 * SGR + i];
        } else if (a.M > 0 && idx < a.P) {   // generic row width (lanes 0..31): coefficients above 15 (and above the active degree) are zero
            for (int k = 0; k < 3 * a.M; k++) a.dL_dsh[(size_t)3 * a.M * idx + k] = k < 45 ? lds_rows[3 * SGR + 45 * t + k] : 0.f;
        }
    }
    if constexpr (ADAM) {
        // the masked Adam of optim_utils.h:102-137 on features_dc and features_rest straight from the rebuilt rows (adam_scalar: the one
        // definition every call site shares), on float4 columns of the block's contiguous regions like the fused backward's phase C
        __shared__ uint8_t lds_vis[64];
        uint8_t myvis = 0;
        if (idx < a.P) {
            if (a.vis_stride) {   // the views' masks of an all-gathered payload, OR-ed here (no MAX-reduce launch in front of the step)
                for (int v = 0; v < a.n_views; v++) myvis |= a.visible[(size_t)v * a.vis_stride + idx];
                myvis = myvis ? 1 : 0;
                if (a.vis_out) a.vis_out[idx] = myvis;
            } else {
                myvis = a.visible[idx] ? 1 : 0;
            }
        }
        lds_vis[t] = myvis;   // (entries 32..63: never read)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const AdamFusedArgs& A = a.adam;
        auto update = [&](int grp, int width, const float* rows_lds, size_t base) {
            // region of this block: rows * width floats at param + base.  The float4 path needs 16-byte alignment of the gradient rows, the
            // parameter and both moments: row0 is a multiple of 32, so that holds whenever the BASE pointers are 16-byte aligned — which a gradient
            // slab with P % 4 != 0 does not give the opacity / rotation runs: checked here (wave-uniform), scalar path otherwise
            const bool aligned16 = ((reinterpret_cast<uintptr_t>(rows_lds) | reinterpret_cast<uintptr_t>(A.p[grp] + base) |
                                     reinterpret_cast<uintptr_t>(A.m[grp] + base) | reinterpret_cast<uintptr_t>(A.v[grp] + base)) & 15) == 0;
            if (rows == SGR && aligned16) {
                const int nv = SGR * width / 4;
                const float4* s4 = reinterpret_cast<const float4*>(rows_lds);
                constexpr int U = 4;
                for (int i0 = t; i0 < nv; i0 += 64 * U) {
                    float4 g[U], p[U], m[U], v[U];
                    bool vis[U][4], any[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const int i = i0 + 64 * u;
                        any[u] = false;
                        if (i < nv) {
                            g[u] = s4[i];
                            const int e = 4 * i;
                            vis[u][0] = lds_vis[e / width]; vis[u][1] = lds_vis[(e + 1) / width];
                            vis[u][2] = lds_vis[(e + 2) / width]; vis[u][3] = lds_vis[(e + 3) / width];
                            any[u] = vis[u][0] | vis[u][1] | vis[u][2] | vis[u][3];
                        }
                        if (any[u]) {
                            p[u] = ld_stream(reinterpret_cast<const float4*>(A.p[grp] + base) + i);
                            m[u] = ld_stream(reinterpret_cast<const float4*>(A.m[grp] + base) + i);
                            v[u] = ld_stream(reinterpret_cast<const float4*>(A.v[grp] + base) + i);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (!any[u]) continue;
                        const int i = i0 + 64 * u;
                        if (vis[u][0]) adam_scalar(p[u].x, g[u].x, m[u].x, v[u].x, A.lr[grp], A.b1, A.b2, A.eps);
                        if (vis[u][1]) adam_scalar(p[u].y, g[u].y, m[u].y, v[u].y, A.lr[grp], A.b1, A.b2, A.eps);
                        if (vis[u][2]) adam_scalar(p[u].z, g[u].z, m[u].z, v[u].z, A.lr[grp], A.b1, A.b2, A.eps);
                        if (vis[u][3]) adam_scalar(p[u].w, g[u].w, m[u].w, v[u].w, A.lr[grp], A.b1, A.b2, A.eps);
                        st_stream(reinterpret_cast<float4*>(A.p[grp] + base) + i, p[u]);
                        st_stream(reinterpret_cast<float4*>(A.m[grp] + base) + i, m[u]);
                        st_stream(reinterpret_cast<float4*>(A.v[grp] + base) + i, v[u]);
                    }
                }
            } else {
                for (int i = t; i < rows * width; i += 64)
                    if (lds_vis[i / width]) adam_scalar(A.p[grp][base + i], rows_lds[i], A.m[grp][base + i], A.v[grp][base + i], A.lr[grp], A.b1, A.b2, A.eps);
            }
        };
        if (a.g_small[0]) {   // the four small groups from their all-reduced gradients (read in place from the slab): the whole optimiser step in this launch
            update(0, 3, a.g_small[0] + (size_t)row0 * 3, (size_t)row0 * 3);
            update(3, 1, a.g_small[1] + (size_t)row0, (size_t)row0);
            update(4, 3, a.g_small[2] + (size_t)row0 * 3, (size_t)row0 * 3);
            update(5, 4, a.g_small[3] + (size_t)row0 * 4, (size_t)row0 * 4);
        }
        update(1, 3, lds_rows, (size_t)row0 * 3);
        if (a.M == 15) {
            update(2, 45, lds_rows + 3 * SGR, (size_t)row0 * 45);
        } else if (a.M > 0 && idx < a.P && lds_vis[t]) {
            for (int k = 0; k < 3 * a.M; k++) {
                const size_t o = (size_t)3 * a.M * idx + k;
                adam_scalar(A.p[2][o], k < 45 ? lds_rows[3 * SGR + 45 * t + k] : 0.f, A.m[2][o], A.v[2][o], A.lr[2], A.b1, A.b2, A.eps);
            }
        }
    }
}

int launch_sh_grad_from_rgb(const ShGradFromRgbArgs& a, hipStream_t s)
{
    if (a.P <= 0) return GSLIC_OK;
    if (a.adam.on) GS_LAUNCH(K_SH_REBUILD, sh_grad_from_rgb_kernel<true>, dim3(div_up(a.P, SGR)), dim3(64), 0, s, a);
    else GS_LAUNCH(K_SH_REBUILD, sh_grad_from_rgb_kernel<false>, dim3(div_up(a.P, SGR)), dim3(64), 0, s, a);
    return GSLIC_OK;
}

}  // namespace gslic
