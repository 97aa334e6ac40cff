// render.hip — the two alpha-blend kernels, designed for 64-lane waves.
//
// render_fwd_kernel  replaces renderCUDA<3> (forward.cu:321-481)
//   ONE WAVE PER 16x16 TILE, four pixels per lane (lane l owns pixels l, l+64, l+128, l+192 of the tile in
//   thread_rank order, i.e. column l&15, rows (l>>4)+{0,4,8,12}).  The wave fetches 64 list entries at a time
//   (one 48-byte record per lane, three dwordx4 loads), pre-scales them, parks them in LDS and fetches entry j with three
//   ds_read_b128 at a wave-uniform address (LDS broadcast, next entry prefetched): the kernel is VALU-issue bound and LDS reads cost
//   no issue slot, ten v_readlane per entry did.  A 4-bit mask per entry says which 16x4 pixel strips it can reach at all; a
//   checkpoint {T, C} per pixel is stored at every 64th entry (bucket = one wave of entries) as one coalesced 1-KiB dwordx4
//   store per quarter.  One wave per workgroup: no barrier anywhere.
//
// render_bwd_kernel  replaces PerGaussianRenderCUDA<3> (backward.cu:379-597)
//   ONE WAVE PER BUCKET of 64 list entries: lane = Gaussian, the tile's pixels stream through the lanes as a 64-deep systolic
//   pipeline; the evolving per-pixel state {ar0, ar1, T, ar2} and the pixel's tag move lane -> lane+1 with one in-place
//   v_mov_b32 DPP wave_shr:1 per value (no ds_bpermute, no copies); lane 0 is then re-loaded with the next pixel by one
//   ds_read_b128 + ds_read_b32 under a one-lane exec mask.  The per-pixel constants (dL/dpixel) are parked in LDS once per tile
//   and fetched with one ds_read_b96 when a lane actually blends.  Only pixels whose n_contrib reaches this bucket are injected
//   (a 64-bit ballot per 64-pixel chunk, walked with s_ff1): pixels that terminated earlier cost no pipeline step at all.  Each
//   lane accumulates its Gaussian's nine 2D gradients in registers and writes them ONCE to its emission slot (plain 48-byte
//   store): no atomics — the sum over a Gaussian's tiles is a contiguous segmented reduction in preprocess_bwd_kernel,
//   deterministic run to run.
#include "gslic_common.h"
#include "kernels.h"
#include <stdlib.h>

namespace gslic {

// Upper bound of p2(x, y) = hA dx^2 + hC dy^2 + nB dx dy (dx = gx - x, dy = gy - y; a negative-definite form scaled by log2 e)
// over the pixel rectangle [x0, x1] x [y0, y1], plus a rounding margin: the maximum of a concave quadratic over a rectangle that
// does not contain its centre lies on an edge facing the centre, where it is a 1-D parabola.  Used to skip whole 16x4 pixel
// strips that an entry cannot reach (alpha < 1/255 everywhere): conservative, so the image is unchanged bit for bit.
__device__ __forceinline__ float strip_max_p2(float hA, float hC, float nB, float gx, float gy, float x0, float x1, float y0, float y1)
{
    const bool in_x = gx >= x0 && gx <= x1, in_y = gy >= y0 && gy <= y1;
    if (in_x && in_y) return 0.0f;
    float best = -3.0e38f;
    if (!in_y) {  // horizontal edge y = ye facing the centre: maximise over x in [x0, x1]
        const float ye = gy < y0 ? y0 : y1;
        const float dy = gy - ye;
        float dx = -(nB * dy) / (2.0f * hA);            // stationary point of hA dx^2 + nB dy dx
        dx = fminf(fmaxf(dx, gx - x1), gx - x0);         // dx = gx - x with x in [x0, x1]
        best = fmaxf(best, (hA * dx) * dx + (hC * dy) * dy + (nB * dx) * dy);
    }
    if (!in_x) {  // vertical edge x = xe
        const float xe = gx < x0 ? x0 : x1;
        const float dx = gx - xe;
        float dy = -(nB * dx) / (2.0f * hC);
        dy = fminf(fmaxf(dy, gy - y1), gy - y0);
        best = fmaxf(best, (hA * dx) * dx + (hC * dy) * dy + (nB * dx) * dy);
    }
    // rounding margin: a few ulp of the largest term anywhere in the rectangle
    const float DX = fmaxf(fabsf(gx - x0), fabsf(gx - x1)), DY = fmaxf(fabsf(gy - y0), fabsf(gy - y1));
    return best + 1.0e-5f * (fabsf(hA) * DX * DX + fabsf(hC) * DY * DY + fabsf(nB) * DX * DY) + 1.0e-6f;
}

// STRICT = the reference's arithmetic operation for operation (forward.cu:424-445: absolute pixel coordinates, power in source
// order without contraction, exp() of the device library, separately rounded products): with bit-identical records the image,
// final_T and n_contrib then equal the reference kernels' bit for bit.  The default (fast) variant pre-scales the conic by log2(e),
// works in tile-relative coordinates and uses v_exp_f32 directly; it differs by rounding only, which flips the alpha < 1/255 /
// T < 1e-4 decisions of a few (pixel, Gaussian) pairs per million (counted in tests/test_fullsize_reference_gpu.py).
template <bool STRICT>
__global__ __launch_bounds__(64) void render_fwd_kernel(RenderFwdArgs a)
{
    __shared__ float4 s_rec[3 * GS_BUCKET];
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
        float fdx = 0, fdy = 0, fhA = 0, fhC = 0, fnB = 0, fop = 0, flop = -__builtin_inff(), fr = 0, fg = 0, fb = 0;
        uint32_t fmask = 0;
        if (lane < m) {
            const uint32_t g = a.point_list[range.x + (uint32_t)(base + lane)];
            const float4* rp = a.rec + 3 * (size_t)g;
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
            fdx = r0.x - (float)tx0; fdy = r0.y - (float)ty0;
            fhA = -0.5f * LOG2E * r0.z; fnB = -LOG2E * r0.w; fhC = -0.5f * LOG2E * r1.x;
            fop = r1.y; fr = r1.z; fg = r1.w; fb = r2.x;
            flop = __builtin_amdgcn_logf(fop);  // log2(opacity): alpha = exp2(p2 + log2 opacity), one multiply less per (pixel, entry)
            // which of the tile's four 16x4 strips (= the four pixels of every lane) can this entry reach at all
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float pm = strip_max_p2(fhA, fhC, fnB, fdx, fdy, 0.0f, 15.0f, (float)(4 * q), (float)(4 * q + 3));
                if (!(fop * __builtin_amdgcn_exp2f(pm) < 0.999f * (1.0f / 255.0f))) fmask |= 1u << q;
            }
        }
        const float lx = (float)(lane & 15), ly = (float)(lane >> 4);
        if constexpr (STRICT) {  // raw record: absolute mean, unscaled conic
            if (lane < m) {
                const uint32_t g = a.point_list[range.x + (uint32_t)(base + lane)];
                const float4* rp = a.rec + 3 * (size_t)g;
                const float4 r0 = rp[0], r1 = rp[1];
                fdx = r0.x; fdy = r0.y; fhA = r0.z; fnB = r0.w; fhC = r1.x;
            }
        }
        // the batch's 64 pre-scaled records are parked in LDS and entry j is fetched with three ds_read_b128 at a wave-uniform
        // address (LDS broadcast; the next entry is in flight while this one is blended): ten v_readlane per entry cost VALU
        // issue slots, which is what bounds this kernel — LDS reads do not
        s_rec[3 * lane] = make_float4(fdx, fdy, fhA, fnB);
        s_rec[3 * lane + 1] = make_float4(fhC, STRICT ? fop : flop, fr, fg);
        s_rec[3 * lane + 2] = make_float4(fb, __uint_as_float(fmask), 0.f, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4 n0 = s_rec[0], n1 = s_rec[1], n2 = s_rec[2];
        for (int j = 0; j < m; j++) {
            const float4 e0 = n0, e1 = n1, e2 = n2;
            if (j + 1 < m) { n0 = s_rec[3 * (j + 1)]; n1 = s_rec[3 * (j + 1) + 1]; n2 = s_rec[3 * (j + 1) + 2]; }
            const uint32_t smask = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(e2.y));
            if (smask == 0u) continue;
            const float gdx = e0.x, gdy = e0.y, hA = e0.z, nB = e0.w, hC = e1.x, op = e1.y, colr = e1.z, colg = e1.w, colb = e2.x;
            const uint32_t contributor = (uint32_t)(base + j + 1);
            if constexpr (STRICT) {
#pragma clang fp contract(off)
                const float dxs = gdx - (float)px;  // float2 d = { xy.x - pixf.x, xy.y - pixf.y } (forward.cu:424)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (!(smask & (1u << q))) continue;  // wave-uniform; conservative (no pixel of the strip reaches alpha >= 1/255)
                    const float dys = gdy - (float)(pyb + 4 * q);
                    const float power = -0.5f * (hA * dxs * dxs + hC * dys * dys) - nB * dxs * dys;
                    const float alpha = fminf(0.99f, op * expf(power));
                    const float test_T = T[q] * (1 - alpha);
                    if (!(power > 0.0f) && !(alpha < 1.0f / 255.0f) && T[q] > 0.f) {
                        if (test_T < 0.0001f) {
                            T[q] = -T[q];
                        } else {
                            Cr[q] += colr * alpha * T[q]; Cg[q] += colg * alpha * T[q]; Cb[q] += colb * alpha * T[q];
                            T[q] = test_T;
                            last[q] = contributor;
                        }
                    }
                }
                continue;
            }
            // e1.y holds log2(opacity) here.  The operation sequence below is repeated verbatim by the backward (GS_CH_BODY), so both
            // sides compute bit-identical alphas and take the same alpha < 1/255 decisions for every (pixel, entry) pair
            const float lop = op;
            const float dx = gdx - lx;
            const float pA = __builtin_fmaf(hA * dx, dx, lop);  // log2(e) * (-1/2 A dx^2) + log2(opacity)
            const float pB = nB * dx;                           // log2(e) * (-B dx)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!(smask & (1u << q))) continue;  // wave-uniform
                const float dy = gdy - (ly + (float)(4 * q));  // one rounding, as d0.y - py in the backward
                const float p2 = __builtin_fmaf(pB, dy, __builtin_fmaf(hC * dy, dy, pA));  // log2(e) * power + log2(opacity)
                const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(p2));
                const float test_T = T[q] * (1.0f - alpha);  // negative (so < 1e-4) once the pixel is done
                if (!(p2 > lop) && !(alpha < (1.0f / 255.0f)) && T[q] > 0.f) {
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

// Whole-wave shift by one lane: lane l >= 1 receives v[l-1], lane 0 receives 0 (bound_ctrl), so source and destination may be the same
// register (v_mov_b32 v, v wave_shr:1): one VALU op per value, no copies.
__device__ __forceinline__ float shift_zero_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ uint32_t shift_zero_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
// the same shift with lane 0 receiving `old` (bound_ctrl off: a lane without a source lane keeps the destination's previous value)
__device__ __forceinline__ float shift_old_f(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int32_t shift_old_i(int32_t old, int32_t v) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false); }

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define GS_PK_FMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))
#define GS_SPLAT(x) ((v2f){(x), (x)})

// =========================================================================================================
// Backward, strict variant (gslic_set_math_mode(1)): ONE WAVE PER BUCKET, the reference's arithmetic operation for operation.
// The evolving state of a pixel {ar0, ar1, T, ar2} and its tag (rel << 16 | py << 8 | 16 px) move lane -> lane+1 with one DPP
// each; the injected values enter at lane 0 from LDS under a one-lane exec mask; the per-pixel constants (dL/dpixel) are parked
// in LDS by pixel index and fetched with one ds_read_b128 when a lane blends.
#define GS_BWD_SHIFT(sl)                                                                                             \
    do {                                                                                                             \
        st.x = shift_zero_f(st.x); st.y = shift_zero_f(st.y); st.z = shift_zero_f(st.z); st.w = shift_zero_f(st.w);  \
        tag = shift_zero_u(tag);                                                                                     \
        if (lane == 0) { /* one ds_read_b128 under a one-lane exec mask, straight into the state registers */       \
            st = *reinterpret_cast<const v4f*>(&init[sl]);                                                           \
            tag = itags[sl];                                                                                         \
        }                                                                                                            \
    } while (0)
#define GS_BWD_SHIFT_ZERO()                                                                                          \
    do {                                                                                                             \
        st.x = shift_zero_f(st.x); st.y = shift_zero_f(st.y); st.z = shift_zero_f(st.z); st.w = shift_zero_f(st.w);  \
        tag = shift_zero_u(tag);                                                                                     \
    } while (0)
// backward.cu:538-581, contraction off, exp() and the IEEE divide of the device library, absolute pixel coordinates: the
// per-instance sums are the reference's Register_* values up to the order in which a lane meets its pixels.
#define GS_BWD_BODY_STRICT()                                                                                         \
    do {                                                                                                             \
        if (kcmp < tag) { /* lane < n_contrib - bucket start: this Gaussian precedes the pixel's last one (backward.cu:538) */ \
            _Pragma("clang fp contract(off)")                                                                        \
            const float pixx = (float)(tx0 + (int)((tag >> 4) & 15u)), pixy = (float)(ty0 + (int)((tag >> 8) & 0xffu)); \
            const float dx = mabs.x - pixx, dy = mabs.y - pixy;                                                      \
            const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;                                \
            const float G = expf(power);                                                                             \
            const float alpha = fminf(0.99f, op * G);                                                                \
            if (!(power > 0.0f) && !(alpha < 1.0f / 255.0f)) {                                                       \
                const float4 gr = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(grec) + (tag & 0xffffu)); \
                const float T = st.z;                                                                                \
                const float dchannel_dcolor = alpha * T;                                                             \
                const float alpha_inverse = 1.0f / (1.0f - alpha);                                                   \
                float dL_dalpha = 0.0f;                                                                              \
                st.x += T * alpha * col_rg.x; acc_rg.x += dchannel_dcolor * gr.x;                                    \
                dL_dalpha += ((col_rg.x * T) - alpha_inverse * (-st.x)) * gr.x;                                      \
                st.y += T * alpha * col_rg.y; acc_rg.y += dchannel_dcolor * gr.y;                                    \
                dL_dalpha += ((col_rg.y * T) - alpha_inverse * (-st.y)) * gr.y;                                      \
                st.w += T * alpha * colb; acc_b += dchannel_dcolor * gr.z;                                           \
                dL_dalpha += ((colb * T) - alpha_inverse * (-st.w)) * gr.z;                                          \
                st.z = T * (1.0f - alpha);                                                                           \
                const float dL_dG = op * dL_dalpha;                                                                  \
                const float gdx = G * dx, gdy = G * dy;                                                              \
                const float dG_ddelx = -gdx * cA - gdy * cB;                                                         \
                const float dG_ddely = -gdy * cC - gdx * cB;                                                         \
                acc_m.x += dL_dG * dG_ddelx * ddelx_dx;                                                              \
                acc_m.y += dL_dG * dG_ddely * ddely_dy;                                                              \
                acc_cxy.x += -0.5f * gdx * dx * dL_dG;                                                               \
                acc_cxy.y += -0.5f * gdx * dy * dL_dG;                                                               \
                acc_cw += -0.5f * gdy * dy * dL_dG;                                                                  \
                acc_op += G * dL_dalpha;                                                                             \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)

__global__ __launch_bounds__(64) void render_bwd_strict_kernel(RenderBwdArgs a)
{
    // LDS: the pixel records of the whole tile (dL/dpixel, by pixel index) and the current chunk's start states
    __shared__ float4 grec[GS_TILE_PIX];
    __shared__ float4 init[64];
    __shared__ uint32_t itags[64];
    const int lane = threadIdx.x;
    const uint32_t bucket = blockIdx.x;
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
    float cA = 0, cB = 0, cC = 0, op = 0, colb = 0;
    v2f col_rg = {0.f, 0.f}, mabs = {0.f, 0.f};
    if (valid) {
        const uint32_t g = a.point_list[range.x + kit];
        const float4* rp = a.rec + 3 * (size_t)g;
        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
        mabs.x = r0.x; mabs.y = r0.y;  // absolute coordinates like the reference
        cA = r0.z; cB = r0.w; cC = r1.x; op = r1.y; col_rg.x = r1.z; col_rg.y = r1.w; colb = r2.x;
    }
    const float ddelx_dx = (float)(0.5 * a.W), ddely_dy = (float)(0.5 * a.H);  // backward.cu:464-465
    // pixel tag = rel << 16 | py << 8 | 16 px, rel = min(n_contrib - bucket start, 64): the low half is the byte offset of the pixel's
    // float4 in grec[] (row stride 256 B); kcmp < tag  <=>  lane < rel
    const uint32_t kcmp = ((uint32_t)lane << 16) | 0xffffu;
    v2f acc_m = {0.f, 0.f}, acc_cxy = {0.f, 0.f}, acc_rg = {0.f, 0.f};
    float acc_cw = 0, acc_op = 0, acc_b = 0;
    const size_t plane = (size_t)a.H * a.W;

    v4f st = {0.f, 0.f, 0.f, 0.f};  // {ar0, ar1, T, ar2}
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
        const uint32_t ncp = inside ? __float_as_uint(pf.w) : 0u;
        const uint32_t pidx = (uint32_t)(c * 64 + lane);
        const uint32_t rel = ncp > bstart ? (ncp - bstart < 64u ? ncp - bstart : 64u) : 0u;
        const uint32_t ftag = (rel << 16) | ((pidx >> 4) << 8) | ((pidx & 15u) << 4);
        grec[c * 64 + lane] = make_float4(fg0, fg1, fg2, 0.f);
        {
#pragma clang fp contract(off)
            init[lane] = make_float4(-pf.x + ck.y, -pf.y + ck.z, ck.x, -pf.z + ck.w);  // ar = -final + sampled (backward.cu:522-523); T
        }
        itags[lane] = ftag;
        uint64_t active = __ballot(ncp > bstart);  // pixels that reach this bucket; the others contribute nothing here
        if (c < 3) load_chunk(c + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        while (active) {
            const int sl = __builtin_ctzll(active);
            active &= active - 1;
            GS_BWD_SHIFT(sl);
            GS_BWD_BODY_STRICT();
        }
        __builtin_amdgcn_wave_barrier();  // init[] is rewritten by the next chunk only after its last read above
    }
    // drain: the last injected pixel still has to pass the bucket's remaining (valid) lanes
    const int nvalid = (n - bstart) < (uint32_t)GS_BUCKET ? (int)(n - bstart) : GS_BUCKET;
#pragma unroll 1
    for (int dr = 1; dr < nvalid; dr++) {
        GS_BWD_SHIFT_ZERO();
        GS_BWD_BODY_STRICT();
    }
    if (valid) {  // every factor was applied term by term, as the reference does
        float4* o = a.partials + 3 * (size_t)slot;
        o[0] = make_float4(acc_m.x, acc_m.y, acc_cxy.x, acc_cxy.y);
        o[1] = make_float4(acc_cw, acc_op, acc_rg.x, acc_rg.y);
        o[2] = make_float4(acc_b, 0.f, 0.f, 0.f);
    }
}

// =========================================================================================================
// Backward, default (fast) variant: CHAINED buckets.  A wave takes `chain` consecutive global buckets and keeps the 64-deep pipeline
// full across their boundaries: between the pixels of bucket k and those of bucket k+1 (same tile) it injects one MARKER; the lane a
// marker reaches stores its nine sums, zeroes them and takes its next Gaussian's parameters from an LDS staging ring — so the 63 drain
// steps are paid once per run of buckets instead of once per bucket (profiles/r02_bwd_pipeline_model.txt: -18 % steps at chain = 8
// on the 2M / 1080p scene), and the per-instance results do not depend on `chain` at all (bit-identical for every value).
//
// What travels lane -> lane+1 is {T, A, tag}: A = sum_ch ar[ch] * dL/dpixel[ch] replaces the colour vector ar[3] of the reference's
// formulation (dL/dalpha only ever needs that dot product: dL/dalpha = A' / (1 - alpha) + T * (c . g), A' = A + T alpha (c . g)),
// three DPP moves per step instead of five.  alpha = min(0.99, exp2(p2 + log2 opacity)) with the SAME operation sequence as
// render_fwd (identical bits on both sides, so both make the same alpha < 1/255 decisions); the products with G = exp(power) that the
// reference forms are written on a = opacity * G, and dL/dopacity is divided by the opacity once per instance.
//
// tag = rel << 16 | py << 8 | 16 px for a pixel (rel = min(n_contrib - bucket start, 64) >= 1: the low half is the byte offset of the
// pixel's float4 in grec[] and two bytes v_cvt_f32_ubyte0/1 turn into coordinates), 0 for an empty slot, 0x80000000 | ring offset for a
// marker; the comparisons are signed, so `kcmp < tag` (lane < rel) is false for markers and `tag < 0` finds them.
struct BwdLane {
    v2f d0, hAC, col_rg;          // centre relative to the tile origin; log2(e)-scaled conic diagonal {-1/2 A, -1/2 C}; colour r, g
    float nB, lop, colb, rop;     // log2(e)-scaled -B; log2(opacity); colour b; 1 / opacity (0 for an empty lane)
    uint32_t slot;                // emission slot the sums go to; 0xffffffff: this lane holds no Gaussian
};

static constexpr int CH_RING = 1;  // parameter staging slots (three float4 columns of 64 lanes each): LDS per wave decides the occupancy here

// Injection: the entry of the NEXT step is fetched one step ahead (every lane reads the same LDS address: a broadcast) and enters at
// lane 0 through the DPP's `old` operand (lane 0 has no source lane and bound_ctrl is off, so it keeps `old`): no exec-mask
// juggling, and the LDS round trip is off the step's critical path.
#define GS_CH_PREFETCH(NST, NTAG, sl)                                                                                \
    do {                                                                                                             \
        NST = *reinterpret_cast<const v2f*>(&init[sl]);                                                              \
        NTAG = itags[sl];                                                                                            \
    } while (0)
// DST <- shift(SRC) with lane 0 <- INJ.  The DPP's destination is the register that held the injected value, so two steps per trip
// with the roles of the two register sets swapped need no copies at all.
#define GS_CH_SHIFT_INJ(DST, DTAG, INJ, ITAG, SRC, STAG)                                                             \
    do {                                                                                                             \
        DST.x = shift_old_f(INJ.x, SRC.x); DST.y = shift_old_f(INJ.y, SRC.y);                                        \
        DTAG = shift_old_i(ITAG, STAG);                                                                              \
        ++step;                                                                                                      \
    } while (0)
#define GS_CH_SHIFT_ZERO()                                                                                           \
    do {                                                                                                             \
        st.x = shift_zero_f(st.x); st.y = shift_zero_f(st.y);                                                        \
        tag = (int32_t)shift_zero_u((uint32_t)tag);                                                                  \
        ++step;                                                                                                      \
    } while (0)
// a marker has arrived: this lane's bucket is complete
#define GS_CH_SWITCH(TAG)                                                                                            \
    do {                                                                                                             \
        if (__builtin_expect(TAG < 0, 0)) {                                                                          \
            float4* cell = reinterpret_cast<float4*>(reinterpret_cast<char*>(&prm[0][0][0]) + ((uint32_t)TAG & 0xffffu)); \
            const BwdLane done = L;                                                                                  \
            bwd_take(L, cell, lane);                                                                                 \
            bwd_park(cell, lane, done, acc_S, acc_cxy, acc_rg, acc_cw, acc_op, acc_b, kx, ky);                       \
            acc_S = acc_cxy = acc_rg = (v2f){0.f, 0.f}; acc_cw = acc_op = acc_b = 0.f;                               \
        }                                                                                                            \
    } while (0)
#define GS_CH_BODY(ST, TAG)                                                                                          \
    do {                                                                                                             \
        if (__builtin_expect(kcmp < TAG, 1)) { /* lane < n_contrib - bucket start: this Gaussian precedes the pixel's last one (backward.cu:538) */ \
            const uint32_t utag = (uint32_t)TAG;                                                                     \
            const float4 gr = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(grec) + (utag & 0xffffu)); \
            const v2f pxy16 = {(float)(utag & 0xffu), (float)((utag >> 8) & 0xffu)}; /* v_cvt_f32_ubyte0 / ubyte1: {16 px, py} */ \
            const v2f d = GS_PK_FMA(pxy16, ((v2f){-0.0625f, -1.0f}), L.d0); /* exact: d0 - {px, py} */               \
            float p2 = __builtin_fmaf(L.hAC.x * d.x, d.x, L.lop); /* the operation sequence of render_fwd */         \
            p2 = __builtin_fmaf(L.hAC.y * d.y, d.y, p2);                                                             \
            p2 = __builtin_fmaf(L.nB * d.x, d.y, p2); /* = log2(e) * power + log2(opacity) */                        \
            const float araw = __builtin_amdgcn_exp2f(p2); /* opacity * G */                                         \
            const float alpha = fminf(0.99f, araw);                                                                  \
            if (__builtin_expect(!(p2 > L.lop) && !(alpha < (1.0f / 255.0f)), 1)) {                                  \
                const v2f grxy = {gr.x, gr.y};                                                                       \
                const float om = 1.0f - alpha;                                                                       \
                const float rinv = __builtin_amdgcn_rcpf(om);                                                        \
                const float Ta = ST.x * alpha;                                                                       \
                float cg = L.col_rg.x * gr.x;                                                                        \
                cg = __builtin_fmaf(L.col_rg.y, gr.y, cg);                                                           \
                cg = __builtin_fmaf(L.colb, gr.z, cg); /* c . dL/dpixel */                                           \
                acc_rg = GS_PK_FMA(GS_SPLAT(Ta), grxy, acc_rg); acc_b = __builtin_fmaf(Ta, gr.z, acc_b);             \
                ST.y = __builtin_fmaf(Ta, cg, ST.y);                                                                 \
                const float dLda = __builtin_fmaf(rinv, ST.y, ST.x * cg);                                            \
                ST.x *= om;                                                                                          \
                const float w = araw * dLda; /* opacity * G * dL/dalpha = G * dL/dG */                               \
                const v2f wd = GS_SPLAT(w) * d;                                                                      \
                acc_S += wd; /* the conic factors of dL/dmean2D are per-Gaussian constants: applied once, at the end */ \
                acc_cxy = GS_PK_FMA(GS_SPLAT(wd.x), d, acc_cxy); /* -0.5 applied at the end */                       \
                acc_cw = __builtin_fmaf(wd.y, d.y, acc_cw);                                                          \
                acc_op += w; /* divided by the opacity at the end */                                                 \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)
#define GS_CH_STEP_IDLE()   do { GS_CH_SHIFT_ZERO(); GS_CH_SWITCH(tag); GS_CH_BODY(st, tag); } while (0)

// sums of one instance -> its emission slot.  acc_S = sum of w d (w = G dL/dG): dL/dmean2D = -(0.5 W, 0.5 H) o (A S.x + B S.y, C S.y + B S.x)
// (backward.cu:566-573), written on the log2(e)-scaled conic the lane holds: A = -2 hA / log2 e, B = -nB / log2 e; kx = 0.5 W / log2 e.
__device__ __forceinline__ void bwd_store(float4* partials, const BwdLane& L, v2f acc_S, v2f acc_cxy, v2f acc_rg, float acc_cw, float acc_op,
                                          float acc_b, float kx, float ky)
{
    if (L.slot != 0xffffffffu) {
        float4* o = partials + 3 * (size_t)L.slot;
        const float gx = __builtin_fmaf(2.0f * L.hAC.x, acc_S.x, L.nB * acc_S.y) * kx;
        const float gy = __builtin_fmaf(2.0f * L.hAC.y, acc_S.y, L.nB * acc_S.x) * ky;
        o[0] = make_float4(gx, gy, -0.5f * acc_cxy.x, -0.5f * acc_cxy.y);
        o[1] = make_float4(-0.5f * acc_cw, acc_op * L.rop, acc_rg.x, acc_rg.y);  // acc_op = sum of opacity * G * dL/dalpha (backward.cu:580 sums G * dL/dalpha)
        o[2] = make_float4(acc_b, 0.f, 0.f, 0.f);
    }
}
// A lane that meets a marker does not store to global memory (a one-lane store occupies the memory pipeline like a full one: 192 store
// instructions per bucket boundary instead of 3): it swaps — takes its next parameters out of its three staging cells and parks the
// finished sums in the same cells; the wave writes a whole slot out (bwd_flush_ring, coalesced) once its marker has passed lane 63.
__device__ __forceinline__ void bwd_park(float4* c, int lane, const BwdLane& L, v2f acc_S, v2f acc_cxy, v2f acc_rg, float acc_cw, float acc_op,
                                         float acc_b, float kx, float ky)
{
    const float gx = __builtin_fmaf(2.0f * L.hAC.x, acc_S.x, L.nB * acc_S.y) * kx;
    const float gy = __builtin_fmaf(2.0f * L.hAC.y, acc_S.y, L.nB * acc_S.x) * ky;
    c[lane] = make_float4(gx, gy, -0.5f * acc_cxy.x, -0.5f * acc_cxy.y);
    c[64 + lane] = make_float4(-0.5f * acc_cw, acc_op * L.rop, acc_rg.x, acc_rg.y);
    c[128 + lane] = make_float4(acc_b, __uint_as_float(L.slot), 0.f, 0.f);
}
__device__ __forceinline__ void bwd_flush_ring(float4* partials, const float4* c, int lane)
{
    const float4 o0 = c[lane], o1 = c[64 + lane], o2 = c[128 + lane];
    const uint32_t slot = __float_as_uint(o2.y);
    if (slot != 0xffffffffu) {
        float4* o = partials + 3 * (size_t)slot;
        o[0] = o0; o[1] = o1; o[2] = make_float4(o2.x, 0.f, 0.f, 0.f);
    }
}
// staging slot: float4 c0[64] {d0.x, d0.y, hA, hC} | c1[64] {nB, lop, col.r, col.g} | c2[64] {col.b, slot bits, 1/opacity, -}: column-major, so
// the wave-wide writes are conflict-free and a lane reads its own three entries
__device__ __forceinline__ void bwd_put(const BwdLane& L, float4* c, int lane)
{
    c[lane] = make_float4(L.d0.x, L.d0.y, L.hAC.x, L.hAC.y);
    c[64 + lane] = make_float4(L.nB, L.lop, L.col_rg.x, L.col_rg.y);
    c[128 + lane] = make_float4(L.colb, __uint_as_float(L.slot), L.rop, 0.f);
}
__device__ __forceinline__ void bwd_take(BwdLane& L, const float4* c, int lane)
{
    const float4 p0 = c[lane], p1 = c[64 + lane], p2 = c[128 + lane];
    L.d0 = (v2f){p0.x, p0.y}; L.hAC = (v2f){p0.z, p0.w};
    L.nB = p1.x; L.lop = p1.y; L.col_rg = (v2f){p1.z, p1.w};
    L.colb = p2.x; L.slot = __float_as_uint(p2.y); L.rop = p2.z;
}
__device__ __forceinline__ void bwd_empty(BwdLane& L)
{
    L.d0 = L.hAC = L.col_rg = (v2f){0.f, 0.f};
    L.nB = L.colb = L.rop = 0.f;
    L.lop = -__builtin_inff();  // alpha = exp2(-inf) = 0: never blends
    L.slot = 0xffffffffu;
}

__global__ __launch_bounds__(64) void render_bwd_chain_kernel(RenderBwdArgs a, int chain)
{
    __shared__ float4 grec[GS_TILE_PIX];          // dL/dpixel of the current tile, by pixel index
    __shared__ float2 init[64 + 1];               // start state {T, A} of the current chunk's pixels (+1: the marker's entry)
    __shared__ int32_t itags[64 + 1];
    __shared__ float4 prm[CH_RING][3][64];        // parameter staging ring
    const int lane = threadIdx.x;
    const uint32_t gb0 = blockIdx.x * (uint32_t)chain;
    const uint32_t nb = ((uint32_t)a.B - gb0) < (uint32_t)chain ? ((uint32_t)a.B - gb0) : (uint32_t)chain;  // <= 64
    const int32_t kcmp = (int32_t)(((uint32_t)lane << 16) | 0xffffu);
    const float LOG2E = 1.4426950408889634f;
    const float kx = 0.5f * (float)a.W / LOG2E, ky = 0.5f * (float)a.H / LOG2E;
    const size_t plane = (size_t)a.H * a.W;

    // ---- metadata of the chain's buckets, one lane per bucket (two dependent rounds of loads for all of them at once)
    // (lane j keeps bucket j's tile, list start, list length and first entry; the wave reads them back with v_readlane)
    bool run_ = false;
    uint32_t t_ = 0, rx_ = 0, n_ = 0, bs_ = 0;
    {
        if ((uint32_t)lane < nb) {
            t_ = a.bucket_to_tile[gb0 + (uint32_t)lane];
            const uint2 rg = a.ranges[t_];
            const uint32_t bbm = (t_ == 0) ? 0u : a.bucket_offsets[t_ - 1];
            rx_ = rg.x; n_ = rg.y - rg.x; bs_ = (gb0 + (uint32_t)lane - bbm) * GS_BUCKET;
            run_ = bs_ < a.max_contrib[t_];  // else: entirely behind every pixel's last contributor (backward.cu:428)
        }
    }
    const uint64_t runmask = __ballot(run_);
    const uint64_t allmask = nb >= 64u ? ~0ull : ((1ull << nb) - 1ull);
#define GS_META(j, tile_v, rx_v, n_v, bs_v)                                \
    const uint32_t tile_v = readlane_u(t_, (j));                          \
    const uint32_t rx_v = readlane_u(rx_, (j));                           \
    const uint32_t n_v = readlane_u(n_, (j));                             \
    const uint32_t bs_v = readlane_u(bs_, (j))

    // ---- pass 1: buckets no pixel reaches get exact zeros (independent of the pipeline; their loads overlap freely)
    for (uint64_t z = allmask & ~runmask; z; z &= z - 1) {
        const int j = __builtin_ctzll(z);
        GS_META(j, zt, zrx, zn, zbs);
        (void)zt;
        const uint32_t kit = zbs + (uint32_t)lane;
        if (kit < zn) {
            float4* o = a.partials + 3 * (size_t)a.inst_slot[zrx + kit];
            o[0] = o[1] = o[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (!runmask) return;

    // ---- pass 2: the running buckets, in order, through one pipeline
    BwdLane L;
    bwd_empty(L);
    v2f acc_S = {0.f, 0.f}, acc_cxy = {0.f, 0.f}, acc_rg = {0.f, 0.f};
    float acc_cw = 0, acc_op = 0, acc_b = 0;
    v2f st = {0.f, 0.f};  // {T, A} travelling through the lanes
    int32_t tag = 0;
    v2f nst = {0.f, 0.f};  // the next step's injection, fetched one step ahead
    int32_t ntag = 0;
    // wave-uniform bookkeeping (steps are counted; the "until" values are step numbers)
    uint32_t step = 0;
    uint32_t drain_until = 0;                 // the pipeline is empty (every item past its last lane) once step >= drain_until
    uint32_t ring_free[CH_RING] = {};         // staging slot r holds parked results of every lane once step >= ring_free[r]
    bool ring_full[CH_RING] = {};             // ... and they have not been written out yet
    uint32_t nmark = 0;                       // markers injected so far
    uint32_t cur_tile = 0xffffffffu;
    bool live = false;                        // the lanes hold a bucket whose sums have not been stored yet

    uint64_t todo = runmask;
    int j = __builtin_ctzll(todo);
    todo &= todo - 1;
    // this lane's Gaussian of a bucket, ready for the pipeline
    auto load_gaussian = [&](BwdLane& G, uint32_t tile_, uint32_t rx_v, uint32_t n_v, uint32_t bs_v) {
        bwd_empty(G);
        const uint32_t kit = bs_v + (uint32_t)lane;  // splat index in tile
        if (kit < n_v) {
            G.slot = a.inst_slot[rx_v + kit];
            const float4* rp = a.rec + 3 * (size_t)a.point_list[rx_v + kit];
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
            G.d0.x = r0.x - (float)((int)(tile_ % (uint32_t)a.gx) * GS_TILE); G.d0.y = r0.y - (float)((int)(tile_ / (uint32_t)a.gx) * GS_TILE);
            G.hAC.x = -0.5f * LOG2E * r0.z; G.nB = -LOG2E * r0.w; G.hAC.y = -0.5f * LOG2E * r1.x;
            G.rop = r1.y > 0.f ? 1.0f / r1.y : 0.f; G.lop = __builtin_amdgcn_logf(r1.y);
            G.col_rg.x = r1.z; G.col_rg.y = r1.w; G.colb = r2.x;
        }
    };
    float4 p_ck, p_pf;  // register prefetch: the next chunk's per-pixel data is loaded one stage ahead
    {
        GS_META(j, t0, rx0, n0, bs0);
        (void)rx0; (void)n0; (void)bs0;
        p_ck = a.ckpt[(size_t)(gb0 + (uint32_t)j) * GS_TILE_PIX + lane];
        p_pf = a.pix_final[(size_t)t0 * GS_TILE_PIX + lane];
    }
    for (;;) {
        GS_META(j, tile, rx, n, bstart);
        const uint32_t gb = gb0 + (uint32_t)j;
        const int j2 = todo ? __builtin_ctzll(todo) : -1;
        const int tx0 = (int)(tile % (uint32_t)a.gx) * GS_TILE, ty0 = (int)(tile / (uint32_t)a.gx) * GS_TILE;
        const uint32_t nvalid = (n - bstart) < (uint32_t)GS_BUCKET ? (n - bstart) : (uint32_t)GS_BUCKET;
        const uint32_t tile2 = j2 >= 0 ? readlane_u(t_, j2) : 0u;

        if (tile != cur_tile) {
            // other tile: its pixels need another grec[].  Let the pipeline run empty, store what the lanes hold, start afresh.
            while (step < drain_until) GS_CH_STEP_IDLE();
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < CH_RING; r++)
                if (ring_full[r]) { bwd_flush_ring(a.partials, &prm[r][0][0], lane); ring_full[r] = false; }
            if (live) bwd_store(a.partials, L, acc_S, acc_cxy, acc_rg, acc_cw, acc_op, acc_b, kx, ky);
            acc_S = acc_cxy = acc_rg = (v2f){0.f, 0.f}; acc_cw = acc_op = acc_b = 0.f;
            live = false;
            cur_tile = tile;
            __builtin_amdgcn_wave_barrier();
#pragma unroll 2
            for (int c = 0; c < 4; c++) {
                const int pidx = c * 64 + lane;
                const int px = tx0 + (pidx & 15), py = ty0 + (pidx >> 4);
                float g0 = 0.f, g1 = 0.f, g2 = 0.f;
                if (px < a.W && py < a.H) {
                    const size_t pid = (size_t)py * a.W + px;
                    g0 = a.dL_dpix[pid]; g1 = a.dL_dpix[plane + pid]; g2 = a.dL_dpix[2 * plane + pid];
                }
                grec[pidx] = make_float4(g0, g1, g2, 0.f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (!live) {
            load_gaussian(L, tile, rx, n, bstart);  // pipeline empty: every lane takes its Gaussian directly
            live = true;
        } else {
            // same tile, pixels of the previous bucket may still be in flight: stage the parameters and send a marker after them
            const uint32_t r = nmark % CH_RING;
            while (step < ring_free[r]) GS_CH_STEP_IDLE();  // the slot's previous marker has not reached lane 63 yet (short buckets only)
            __builtin_amdgcn_wave_barrier();
            if (ring_full[r]) bwd_flush_ring(a.partials, &prm[r][0][0], lane);  // every lane has parked its previous bucket there
            __builtin_amdgcn_wave_barrier();
            {   // (loaded after the waits above: not held in registers across a step loop — registers decide the occupancy here)
                BwdLane N;
                load_gaussian(N, tile, rx, n, bstart);
                bwd_put(N, &prm[r][0][0], lane);
            }
            if (lane == 0) { itags[64] = (int32_t)(0x80000000u | (uint32_t)(r * sizeof(prm[0]))); init[64] = make_float2(0.f, 0.f); }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            GS_CH_PREFETCH(nst, ntag, 64);
            GS_CH_SHIFT_INJ(st, tag, nst, ntag, st, tag); GS_CH_SWITCH(tag); GS_CH_BODY(st, tag);
            ring_free[r] = step + 63;
            ring_full[r] = true;
            drain_until = step + 63;
            nmark++;
        }

        // the tile's pixels that reach this bucket, 64 at a time; the next chunk (of this or of the next bucket) is in flight
#pragma unroll 1
        for (int c = 0; c < 4; c++) {
            const float4 ck = p_ck, pf = p_pf;
            if (c < 3) {
                p_ck = a.ckpt[(size_t)gb * GS_TILE_PIX + (c + 1) * 64 + lane];
                p_pf = a.pix_final[(size_t)tile * GS_TILE_PIX + (c + 1) * 64 + lane];
            } else if (j2 >= 0) {
                p_ck = a.ckpt[(size_t)(gb0 + (uint32_t)j2) * GS_TILE_PIX + lane];
                p_pf = a.pix_final[(size_t)tile2 * GS_TILE_PIX + lane];
            }
            const int pidx = c * 64 + lane;
            const int px = tx0 + (pidx & 15), py = ty0 + (pidx >> 4);
            const bool inside = px < a.W && py < a.H;
            const uint32_t ncp = inside ? __float_as_uint(pf.w) : 0u;
            const uint32_t rel = ncp > bstart ? (ncp - bstart < 64u ? ncp - bstart : 64u) : 0u;
            const float4 gr = grec[pidx];
            float A0 = (ck.y - pf.x) * gr.x;  // ar = checkpoint colour - final colour (backward.cu:522-523), dotted with dL/dpixel
            A0 = __builtin_fmaf(ck.z - pf.y, gr.y, A0);
            A0 = __builtin_fmaf(ck.w - pf.z, gr.z, A0);
            __builtin_amdgcn_wave_barrier();  // the previous chunk's last lane-0 read of init[] / itags[] precedes these writes
            init[lane] = make_float2(ck.x, A0);
            itags[lane] = (int32_t)((rel << 16) | (((uint32_t)pidx >> 4) << 8) | (((uint32_t)pidx & 15u) << 4));
            uint64_t active = __ballot(ncp > bstart);  // pixels that reach this bucket; the others contribute nothing here
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (active) {
                // two steps per trip: half the loop-closing branches, and the state ping-pongs between two register sets (a DPP's
                // destination is the register that held the injected value)
                // register sets: (st, tag) holds the state and (nst, ntag) the fetched injection on entry to a trip; inside, (st2, tag2) is
                // the state after the first step (in the registers of nst / ntag) and (nst2, ntag2) the second injection (in those of st / tag)
                GS_CH_PREFETCH(nst, ntag, __builtin_ctzll(active));
                for (;;) {
                    v2f st2, nst2 = {0.f, 0.f};
                    int32_t tag2, ntag2 = 0;
                    active &= active - 1;
                    GS_CH_SHIFT_INJ(st2, tag2, nst, ntag, st, tag);
                    if (active) GS_CH_PREFETCH(nst2, ntag2, __builtin_ctzll(active));
                    GS_CH_SWITCH(tag2); GS_CH_BODY(st2, tag2);
                    if (!active) { st = st2; tag = tag2; break; }
                    active &= active - 1;
                    GS_CH_SHIFT_INJ(st, tag, nst2, ntag2, st2, tag2);
                    if (active) GS_CH_PREFETCH(nst, ntag, __builtin_ctzll(active));
                    GS_CH_SWITCH(tag); GS_CH_BODY(st, tag);
                    if (!active) break;
                }
                const uint32_t du = step + nvalid - 1;  // the last pixel still has to pass the bucket's remaining (valid) lanes
                drain_until = du > drain_until ? du : drain_until;
            }
        }
        if (j2 < 0) break;
        j = j2;
        todo &= todo - 1;
    }
    while (step < drain_until) GS_CH_STEP_IDLE();
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < CH_RING; r++)
        if (ring_full[r]) bwd_flush_ring(a.partials, &prm[r][0][0], lane);
    if (live) bwd_store(a.partials, L, acc_S, acc_cxy, acc_rg, acc_cw, acc_op, acc_b, kx, ky);
#undef GS_META
}

static int clamp_chain(int v) { return v < 1 ? 1 : (v > 64 ? 64 : v); }
static int g_bwd_chain = clamp_chain(getenv("GSLIC_BWD_CHAIN") ? atoi(getenv("GSLIC_BWD_CHAIN")) : 8);
int set_bwd_chain(int k)
{
    const int old = g_bwd_chain;
    if (k > 0) g_bwd_chain = clamp_chain(k);
    return old;
}
static int bwd_chain_length() { return g_bwd_chain; }

int launch_render_fwd(const RenderFwdArgs& a, hipStream_t s)
{
    if (g_strict_math) GS_LAUNCH(K_RENDER_FWD, render_fwd_kernel<true>, dim3(a.gx * a.gy), dim3(64), 0, s, a);
    else GS_LAUNCH(K_RENDER_FWD, render_fwd_kernel<false>, dim3(a.gx * a.gy), dim3(64), 0, s, a);
    return GSLIC_OK;
}
int launch_render_bwd(const RenderBwdArgs& a, hipStream_t s)
{
    if (a.B <= 0) return GSLIC_OK;
    if (g_strict_math) {
        GS_LAUNCH(K_RENDER_BWD, render_bwd_strict_kernel, dim3(a.B), dim3(64), 0, s, a);
    } else {
        const int chain = bwd_chain_length();
        GS_LAUNCH(K_RENDER_BWD, render_bwd_chain_kernel, dim3((a.B + chain - 1) / chain), dim3(64), 0, s, a, chain);
    }
    return GSLIC_OK;
}

}  // namespace gslic
