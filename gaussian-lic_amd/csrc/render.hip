// render.hip — the two alpha-blend kernels, designed for 64-lane waves.
//
// render_fwd_kernel  replaces renderCUDA<3> (forward.cu:321-481)
//   ONE WAVE PER 16x16 TILE, four pixels per lane (lane l owns pixels l, l+64, l+128, l+192 of the tile in
//   thread_rank order, i.e. column l&15, rows (l>>4)+{0,4,8,12}).  The wave fetches 64 list entries at a time
//   (one 48-byte record per lane, three dwordx4 loads) and broadcasts entry j with v_readlane into SGPRs, so the
//   inner loop has no LDS traffic, no barrier and no shared memory at all; a checkpoint {T, C} per pixel is
//   stored at every 64th entry (bucket = one wave of entries) as one coalesced 1-KiB dwordx4 store per quarter.
//
// render_bwd_kernel  replaces PerGaussianRenderCUDA<3> (backward.cu:379-597)
//   ONE WAVE PER BUCKET of 64 list entries: lane = Gaussian, the tile's 256 pixels stream through the lanes as a
//   64-deep systolic pipeline; the evolving per-pixel state {T, ar[3], n_contrib|index} moves lane -> lane+1 with one
//   v_mov_b32 DPP wave_shr:1 per value (no ds_bpermute), injected through the DPP's bound-lane `old` operand; the per-pixel
//   constants (dL/dpixel) are parked in LDS per chunk and fetched with one ds_read_b128 per step.  Only pixels whose n_contrib reaches this bucket are injected (a 64-bit ballot per chunk,
//   walked with s_ff1): pixels that terminated earlier cost no pipeline step at all.  Each lane accumulates its Gaussian's nine 2D gradients in registers and writes them
//   ONCE to its emission slot (plain 48-byte store): no atomics — the sum over a Gaussian's tiles is a
//   contiguous segmented reduction in preprocess_bwd_kernel, deterministic run to run.
#include "gslic_common.h"
#include "kernels.h"

namespace gslic {

__global__ __launch_bounds__(64) void render_fwd_kernel(RenderFwdArgs a)
{
    const int tile = blockIdx.x;
    const int lane = threadIdx.x;
    const int tx0 = (tile % a.gx) * GS_TILE, ty0 = (tile / a.gx) * GS_TILE;
    const uint2 range = a.ranges[tile];
    const int n = (int)(range.y - range.x);
    const bool color = !a.no_color;

    uint32_t bbm = 0;
    if (color) {
        bbm = (tile == 0) ? 0u : a.bucket_offsets[tile - 1];
        const int nb = (n + GS_BUCKET - 1) / GS_BUCKET;
        for (int b = lane; b < nb; b += 64) a.bucket_to_tile[bbm + b] = (uint32_t)tile;
    }

    const int px = tx0 + (lane & 15);
    const int pyb = ty0 + (lane >> 4);
    // The sign of T carries the `done` flag (forward.cu:352,439-443): T > 0 = still blending, T < 0 = finished with
    // transmittance |T| (T never reaches 0: blending stops below 1e-4).  One register and no flag bookkeeping per pixel.
    float T[4], Cr[4], Cg[4], Cb[4];
    uint32_t last[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int py = pyb + 4 * q;
        T[q] = (px < a.W && py < a.H) ? 1.0f : -1.0f;
        Cr[q] = Cg[q] = Cb[q] = 0.0f;
        last[q] = 0;
    }
    const float LOG2E = 1.4426950408889634f;

    for (int base = 0; base < n; base += GS_BUCKET) {
        if (__all(T[0] < 0.f && T[1] < 0.f && T[2] < 0.f && T[3] < 0.f)) break;
        if (color) {
            float4* ck = a.ckpt + ((size_t)(bbm + (uint32_t)(base / GS_BUCKET)) * GS_TILE_PIX) + lane;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (T[q] > 0.f) ck[q * 64] = make_float4(T[q], Cr[q], Cg[q], Cb[q]);
        }
        const int m = (n - base) < GS_BUCKET ? (n - base) : GS_BUCKET;
        // each lane fetches one record and pre-scales its conic: exponent in base 2, relative to this lane-independent tile origin
        float fdx = 0, fdy = 0, fhA = 0, fhC = 0, fnB = 0, fop = 0, fr = 0, fg = 0, fb = 0;
        if (lane < m) {
            const uint32_t g = a.point_list[range.x + (uint32_t)(base + lane)];
            const float4* rp = a.rec + 3 * (size_t)g;
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
            fdx = r0.x - (float)tx0; fdy = r0.y - (float)ty0;
            fhA = -0.5f * LOG2E * r0.z; fnB = -LOG2E * r0.w; fhC = -0.5f * LOG2E * r1.x;
            fop = r1.y; fr = r1.z; fg = r1.w; fb = r2.x;
        }
        const float lx = (float)(lane & 15), ly = (float)(lane >> 4);
        for (int j = 0; j < m; j++) {
            const float gdx = readlane_f(fdx, j), gdy = readlane_f(fdy, j);
            const float hA = readlane_f(fhA, j), nB = readlane_f(fnB, j), hC = readlane_f(fhC, j);
            const float op = readlane_f(fop, j);
            const float colr = readlane_f(fr, j), colg = readlane_f(fg, j), colb = readlane_f(fb, j);
            const uint32_t contributor = (uint32_t)(base + j + 1);
            const float dx = gdx - lx;
            const float pA = (hA * dx) * dx;   // log2(e) * (-1/2 A dx^2)
            const float pB = nB * dx;          // log2(e) * (-B dx)
            const float dy0 = gdy - ly;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float dy = dy0 - (float)(4 * q);
                const float p2 = __builtin_fmaf(pB, dy, __builtin_fmaf(hC * dy, dy, pA));  // log2(e) * power
                const float alpha = fminf(0.99f, op * __builtin_amdgcn_exp2f(p2));
                const float test_T = T[q] * (1.0f - alpha);  // negative (so < 1e-4) once the pixel is done
                if (!(p2 > 0.0f) && !(alpha < (1.0f / 255.0f)) && T[q] > 0.f) {
                    if (test_T < 0.0001f) {
                        T[q] = -T[q];  // done; this entry is NOT applied (forward.cu:438-443)
                    } else {
                        const float w = alpha * T[q];
                        Cr[q] = __builtin_fmaf(colr, w, Cr[q]); Cg[q] = __builtin_fmaf(colg, w, Cg[q]); Cb[q] = __builtin_fmaf(colb, w, Cb[q]);
                        T[q] = test_T;
                        last[q] = contributor;
                    }
                }
            }
        }
    }

    uint32_t mymax = 0;
    const size_t plane = (size_t)a.H * a.W;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int py = pyb + 4 * q;
        if (px < a.W && py < a.H) {
            const size_t pid = (size_t)py * a.W + px;
            a.out_final_T[pid] = fabsf(T[q]);
            if (color) {
                a.out_color[pid] = Cr[q];
                a.out_color[plane + pid] = Cg[q];
                a.out_color[2 * plane + pid] = Cb[q];
            }
        }
        if (color) {
            a.pix_final[(size_t)tile * GS_TILE_PIX + q * 64 + lane] = make_float4(Cr[q], Cg[q], Cb[q], __uint_as_float(last[q]));
            mymax = last[q] > mymax ? last[q] : mymax;
        }
    }
    if (color) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)mymax, d, 64);
            mymax = o > mymax ? o : mymax;
        }
        if (lane == 0) a.max_contrib[tile] = mymax;
    }
}

// Whole-wave shift by one lane with injection: lane l >= 1 receives v[l-1]; lane 0 has no source lane, so with
// bound_ctrl = 0 it keeps the DPP "old" operand — which we set to the (wave-uniform) value to inject.  2 VALU ops.
__device__ __forceinline__ float shift_in_f(float inject, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(inject), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t shift_in_u(uint32_t inject, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)inject, (int)v, 0x138, 0xf, 0xf, false);
}

// One pipeline step.  The evolving part of a pixel's state {T, ar[3]} and its tag (n_contrib << 8 | pixel index) move
// lane -> lane+1 with one DPP each; the injected values enter through the DPP's `old` operand (lane 0 has no source lane).
// The per-pixel CONSTANTS (dL/dpixel) do not travel: the wave parks every 64-pixel chunk in LDS once (coalesced
// ds_write_b128) and a lane fetches its current pixel's record with one ds_read_b128.  VALU is the bound of this kernel
// (profiles/r01_sq_counters_render.md), so every value taken off the conveyor is three VALU ops saved per step.
#define GS_BWD_STEP(inj /*float4 {T, ar0..2}*/, itag)                                                                \
    do {                                                                                                             \
        T = shift_in_f((inj).x, T); ar0 = shift_in_f((inj).y, ar0); ar1 = shift_in_f((inj).z, ar1);                  \
        ar2 = shift_in_f((inj).w, ar2); tag = shift_in_u(itag, tag);                                                 \
        if (kcmp < tag) { /* kit < n_contrib of this pixel (backward.cu:538) */                                     \
            const float4 gr = grec[tag & 255u];                                                                      \
            const float dx = dx0 - (float)(tag & 15u);                                                               \
            const float dy = dy0 - (float)((tag >> 4) & 15u);                                                        \
            float p2 = (hA * dx) * dx;                                                                               \
            p2 = __builtin_fmaf(hC * dy, dy, p2);                                                                    \
            p2 = __builtin_fmaf(nB * dx, dy, p2); /* = log2(e) * power */                                            \
            const float G = __builtin_amdgcn_exp2f(p2);                                                              \
            const float alpha = fminf(0.99f, op * G);                                                                \
            if (!(p2 > 0.0f) && !(alpha < (1.0f / 255.0f))) {                                                        \
                const float om = 1.0f - alpha;                                                                       \
                const float rinv = __builtin_amdgcn_rcpf(om);                                                        \
                const float Ta = T * alpha;                                                                          \
                ar0 = __builtin_fmaf(Ta, colr, ar0); ar1 = __builtin_fmaf(Ta, colg, ar1); ar2 = __builtin_fmaf(Ta, colb, ar2); \
                acc_r = __builtin_fmaf(Ta, gr.x, acc_r); acc_g = __builtin_fmaf(Ta, gr.y, acc_g); acc_b = __builtin_fmaf(Ta, gr.z, acc_b); \
                float dLda = __builtin_fmaf(rinv, ar0, colr * T) * gr.x;                                             \
                dLda = __builtin_fmaf(__builtin_fmaf(rinv, ar1, colg * T), gr.y, dLda);                              \
                dLda = __builtin_fmaf(__builtin_fmaf(rinv, ar2, colb * T), gr.z, dLda);                              \
                T *= om;                                                                                             \
                const float q = op * dLda; /* dL/dG */                                                               \
                const float gdx = G * dx, gdy = G * dy;                                                              \
                acc_mx = __builtin_fmaf(q, __builtin_fmaf(gdx, cA, gdy * cB), acc_mx); /* sign and 0.5*W applied at the end */ \
                acc_my = __builtin_fmaf(q, __builtin_fmaf(gdy, cC, gdx * cB), acc_my);                               \
                acc_cx = __builtin_fmaf(gdx * dx, q, acc_cx); /* -0.5 applied at the end */                          \
                acc_cy = __builtin_fmaf(gdx * dy, q, acc_cy);                                                        \
                acc_cw = __builtin_fmaf(gdy * dy, q, acc_cw);                                                        \
                acc_op = __builtin_fmaf(G, dLda, acc_op);                                                            \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)

__global__ __launch_bounds__(256) void render_bwd_kernel(RenderBwdArgs a)
{
    // per-wave LDS: the pixel records of the whole tile (dL/dpixel, by pixel index) and the current chunk's start states
    __shared__ float4 s_grec[4][GS_TILE_PIX];
    __shared__ float4 s_init[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float4* const grec = s_grec[wave];
    float4* const init = s_init[wave];
    const uint32_t bucket = blockIdx.x * 4u + (uint32_t)wave;
    if (bucket >= (uint32_t)a.B) return;
    const uint32_t tile = a.bucket_to_tile[bucket];
    const uint2 range = a.ranges[tile];
    const uint32_t n = range.y - range.x;
    const uint32_t bbm = (tile == 0) ? 0u : a.bucket_offsets[tile - 1];
    const uint32_t bit = bucket - bbm;
    const uint32_t bstart = bit * GS_BUCKET;
    const uint32_t kit = bstart + (uint32_t)lane;  // splat index in tile
    const bool valid = kit < n;
    const uint32_t slot = valid ? a.inst_slot[range.x + kit] : 0u;

    // bucket entirely behind every pixel's last contributor (backward.cu:428): gradients are exactly zero
    if (bstart >= a.max_contrib[tile]) {
        if (valid) {
            float4* o = a.partials + 3 * (size_t)slot;
            o[0] = o[1] = o[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

    const int tx0 = (int)(tile % (uint32_t)a.gx) * GS_TILE, ty0 = (int)(tile / (uint32_t)a.gx) * GS_TILE;
    float dx0 = 0, dy0 = 0, cA = 0, cB = 0, cC = 0, op = 0, colr = 0, colg = 0, colb = 0;
    if (valid) {
        const uint32_t g = a.point_list[range.x + kit];
        const float4* rp = a.rec + 3 * (size_t)g;
        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
        dx0 = r0.x - (float)tx0; dy0 = r0.y - (float)ty0;
        cA = r0.z; cB = r0.w; cC = r1.x; op = r1.y; colr = r1.z; colg = r1.w; colb = r2.x;
    }
    const float LOG2E = 1.4426950408889634f;
    const float hA = -0.5f * LOG2E * cA, hC = -0.5f * LOG2E * cC, nB = -LOG2E * cB;
    const uint32_t kcmp = (kit << 8) | 0xffu;  // kcmp < (n_contrib << 8 | idx)  <=>  kit < n_contrib
    float acc_mx = 0, acc_my = 0, acc_cx = 0, acc_cy = 0, acc_cw = 0, acc_op = 0, acc_r = 0, acc_g = 0, acc_b = 0;
    const size_t plane = (size_t)a.H * a.W;

    // evolving pixel state travelling through the lanes
    float T = 0, ar0 = 0, ar1 = 0, ar2 = 0;
    uint32_t tag = 0;

    // 64-pixel feed chunk (register double buffer: chunk c+1 is in flight while chunk c streams through)
    float4 ck, pf;
    float fg0, fg1, fg2;
    bool inside;
    auto load_chunk = [&](int c) {
        const int pidx = c * 64 + lane;
        ck = a.ckpt[(size_t)bucket * GS_TILE_PIX + pidx];
        pf = a.pix_final[(size_t)tile * GS_TILE_PIX + pidx];
        const int px = tx0 + (pidx & 15), py = ty0 + (pidx >> 4);
        inside = px < a.W && py < a.H;
        fg0 = fg1 = fg2 = 0.f;
        if (inside) {
            const size_t pid = (size_t)py * a.W + px;
            fg0 = a.dL_dpix[pid]; fg1 = a.dL_dpix[plane + pid]; fg2 = a.dL_dpix[2 * plane + pid];
        }
    };
    load_chunk(0);
#pragma unroll 1
    for (int c = 0; c < 4; c++) {
        // park this chunk in LDS, then start the next chunk's global loads
        const uint32_t ncp = inside ? __float_as_uint(pf.w) : 0u;
        const uint32_t ftag = (ncp << 8) | (uint32_t)(c * 64 + lane);
        grec[c * 64 + lane] = make_float4(fg0, fg1, fg2, 0.f);
        init[lane] = make_float4(ck.x, ck.y - pf.x, ck.z - pf.y, ck.w - pf.z);  // T, ar = checkpoint colour - final colour
        uint64_t active = __ballot(ncp > bstart);  // pixels that reach this bucket; the others contribute nothing here
        if (c < 3) load_chunk(c + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (active) {
            const int sl = __builtin_ctzll(active);
            active &= active - 1;
            const float4 inj = init[sl];  // wave-uniform address: LDS broadcast
            const uint32_t itag = readlane_u(ftag, sl);
            GS_BWD_STEP(inj, itag);
        }
        __builtin_amdgcn_wave_barrier();  // init[] is rewritten by the next chunk only after its last read above
    }
    // drain: the last injected pixel still has to pass the bucket's remaining (valid) lanes
    const int nvalid = (n - bstart) < (uint32_t)GS_BUCKET ? (int)(n - bstart) : GS_BUCKET;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int d = 1; d < nvalid; d++) GS_BWD_STEP(zero4, 0u);

    if (valid) {
        const float sx = -0.5f * (float)a.W, sy = -0.5f * (float)a.H;  // -(...) * ddelx_dx, ddelx_dx = 0.5 W (backward.cu:464-465)
        float4* o = a.partials + 3 * (size_t)slot;
        o[0] = make_float4(acc_mx * sx, acc_my * sy, -0.5f * acc_cx, -0.5f * acc_cy);
        o[1] = make_float4(-0.5f * acc_cw, acc_op, acc_r, acc_g);
        o[2] = make_float4(acc_b, 0.f, 0.f, 0.f);
    }
}

int launch_render_fwd(const RenderFwdArgs& a, hipStream_t s)
{
    GS_LAUNCH(K_RENDER_FWD, render_fwd_kernel, dim3(a.gx * a.gy), dim3(64), 0, s, a);
    return GSLIC_OK;
}
int launch_render_bwd(const RenderBwdArgs& a, hipStream_t s)
{
    if (a.B <= 0) return GSLIC_OK;
    GS_LAUNCH(K_RENDER_BWD, render_bwd_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a);
    return GSLIC_OK;
}

}  // namespace gslic
