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
        float fdx = 0, fdy = 0, fhA = 0, fhC = 0, fnB = 0, fop = 0, fr = 0, fg = 0, fb = 0;
        uint32_t fmask = 0;
        if (lane < m) {
            const uint32_t g = a.point_list[range.x + (uint32_t)(base + lane)];
            const float4* rp = a.rec + 3 * (size_t)g;
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
            fdx = r0.x - (float)tx0; fdy = r0.y - (float)ty0;
            fhA = -0.5f * LOG2E * r0.z; fnB = -LOG2E * r0.w; fhC = -0.5f * LOG2E * r1.x;
            fop = r1.y; fr = r1.z; fg = r1.w; fb = r2.x;
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
        s_rec[3 * lane + 1] = make_float4(fhC, fop, fr, fg);
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
            const float dx = gdx - lx;
            const float pA = (hA * dx) * dx;   // log2(e) * (-1/2 A dx^2)
            const float pB = nB * dx;          // log2(e) * (-B dx)
            const float dy0 = gdy - ly;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!(smask & (1u << q))) continue;  // wave-uniform
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

// Whole-wave shift by one lane: lane l >= 1 receives v[l-1], lane 0 receives 0 (bound_ctrl), so source and destination may be the same
// register (v_mov_b32 v, v wave_shr:1): one VALU op per value, no copies.
__device__ __forceinline__ float shift_zero_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ uint32_t shift_zero_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define GS_PK_FMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))
#define GS_SPLAT(x) ((v2f){(x), (x)})

// One pipeline step.  The evolving part of a pixel's state {T, ar[3]} and its tag (rel << 16 | py << 8 | 16 px, see kcmp) move
// lane -> lane+1 with one DPP each; the injected values enter through the DPP's `old` operand (lane 0 has no source lane).
// The per-pixel CONSTANTS (dL/dpixel) do not travel: the wave parks every 64-pixel chunk in LDS once (coalesced
// ds_write_b128) and a lane fetches its current pixel's record with one ds_read_b128.  VALU is the bound of this kernel
// (profiles/r01i_sq_counters.txt), so every value taken off the conveyor is three VALU ops saved per step, and the
// gradient arithmetic is written on float2 pairs: gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 at the rate of the scalar forms
// (the fp32 peak of the part assumes them), with op_sel covering the splats and the one swizzle for free.  Each component
// still sees exactly the scalar operation sequence, so the results do not change.
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
#define GS_BWD_BODY()                                                                                                \
    do {                                                                                                             \
        if (kcmp < tag) { /* lane < n_contrib - bucket start: this Gaussian precedes the pixel's last one (backward.cu:538) */ \
            const v2f pxy16 = {(float)(tag & 0xffu), (float)((tag >> 8) & 0xffu)}; /* v_cvt_f32_ubyte0 / ubyte1: {16 px, py} */ \
            const v2f d = GS_PK_FMA(pxy16, ((v2f){-0.0625f, -1.0f}), d0); /* exact: d0 - {px, py} */                 \
            float p2 = (hA * d.x) * d.x; /* same operation order as render_fwd: identical alpha on both sides */    \
            p2 = __builtin_fmaf(hC * d.y, d.y, p2);                                                                  \
            p2 = __builtin_fmaf(nB * d.x, d.y, p2); /* = log2(e) * power */                                          \
            const float G = __builtin_amdgcn_exp2f(p2);                                                              \
            const float alpha = fminf(0.99f, op * G);                                                                \
            if (!(p2 > 0.0f) && !(alpha < (1.0f / 255.0f))) {                                                        \
                const float4 gr = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(grec) + (tag & 0xffffu)); \
                const v2f grxy = {gr.x, gr.y};                                                                       \
                const float om = 1.0f - alpha;                                                                       \
                const float rinv = __builtin_amdgcn_rcpf(om);                                                        \
                const float Ta = st.z * alpha;                                                                          \
                st.xy = GS_PK_FMA(GS_SPLAT(Ta), col_rg, st.xy); st.w = __builtin_fmaf(Ta, colb, st.w);                   \
                acc_rg = GS_PK_FMA(GS_SPLAT(Ta), grxy, acc_rg); acc_b = __builtin_fmaf(Ta, gr.z, acc_b);             \
                const v2f t = GS_PK_FMA(GS_SPLAT(rinv), st.xy, col_rg * GS_SPLAT(st.z));                                 \
                const float tb = __builtin_fmaf(rinv, st.w, colb * st.z);                                                \
                float dLda = t.x * gr.x;                                                                             \
                dLda = __builtin_fmaf(t.y, gr.y, dLda);                                                              \
                dLda = __builtin_fmaf(tb, gr.z, dLda);                                                               \
                st.z *= om;                                                                                          \
                const float q = op * dLda; /* dL/dG */                                                               \
                const v2f gd = GS_SPLAT(G) * d;                                                                      \
                const v2f gdyx = {gd.y, gd.x};                                                                       \
                const v2f inner = GS_PK_FMA(gd, cAC, gdyx * GS_SPLAT(cB)); /* sign and 0.5*W applied at the end */  \
                acc_m = GS_PK_FMA(GS_SPLAT(q), inner, acc_m);                                                        \
                acc_cxy = GS_PK_FMA(GS_SPLAT(gd.x) * d, GS_SPLAT(q), acc_cxy); /* -0.5 applied at the end */         \
                acc_cw = __builtin_fmaf(gd.y * d.y, q, acc_cw);                                                      \
                acc_op = __builtin_fmaf(G, dLda, acc_op);                                                            \
            }                                                                                                        \
        }                                                                                                            \
    } while (0)

// The same step with the reference's arithmetic operation for operation (backward.cu:538-581, contraction off, exp() and the
// IEEE divide of the device library, absolute pixel coordinates): the per-instance sums are then the reference's Register_*
// values up to the order in which a lane meets its pixels.
#define GS_BWD_BODY_STRICT()                                                                                         \
    do {                                                                                                             \
        if (kcmp < tag) {                                                                                            \
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
#define GS_BWD_STEP_BODY()                                                                                           \
    do {                                                                                                             \
        if constexpr (STRICT) GS_BWD_BODY_STRICT(); else GS_BWD_BODY();                                              \
    } while (0)

static constexpr int BWD_WAVES = 1;  // buckets (waves) per workgroup
template <bool STRICT>
__global__ __launch_bounds__(64 * BWD_WAVES) void render_bwd_kernel(RenderBwdArgs a)
{
    // per-wave LDS: the pixel records of the whole tile (dL/dpixel, by pixel index) and the current chunk's start states
    __shared__ float4 s_grec[BWD_WAVES][GS_TILE_PIX];
    __shared__ float4 s_init[BWD_WAVES][64];
    __shared__ uint32_t s_itag[BWD_WAVES][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float4* const grec = s_grec[wave];
    float4* const init = s_init[wave];
    uint32_t* const itags = s_itag[wave];
    const uint32_t bucket = blockIdx.x * (uint32_t)BWD_WAVES + (uint32_t)wave;
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
    v2f d0 = {0.f, 0.f}, col_rg = {0.f, 0.f}, mabs = {0.f, 0.f};
    if (valid) {
        const uint32_t g = a.point_list[range.x + kit];
        const float4* rp = a.rec + 3 * (size_t)g;
        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
        mabs.x = r0.x; mabs.y = r0.y;  // STRICT works on absolute coordinates like the reference
        d0.x = r0.x - (float)tx0; d0.y = r0.y - (float)ty0;
        cA = r0.z; cB = r0.w; cC = r1.x; op = r1.y; col_rg.x = r1.z; col_rg.y = r1.w; colb = r2.x;
    }
    const float LOG2E = 1.4426950408889634f;
    const float hA = -0.5f * LOG2E * cA, hC = -0.5f * LOG2E * cC, nB = -LOG2E * cB;
    const v2f cAC = {cA, cC};
    const float ddelx_dx = (float)(0.5 * a.W), ddely_dy = (float)(0.5 * a.H);  // backward.cu:464-465
    // pixel tag = rel << 16 | py << 8 | 16 px, rel = min(n_contrib - bucket start, 64): the low half is both the byte offset of the pixel's
    // float4 in grec[] (row stride 256 B) and two bytes v_cvt_f32_ubyte0/1 turn into coordinates; kcmp < tag  <=>  lane < rel
    const uint32_t kcmp = ((uint32_t)lane << 16) | 0xffffu;
    v2f acc_m = {0.f, 0.f}, acc_cxy = {0.f, 0.f}, acc_rg = {0.f, 0.f};
    float acc_cw = 0, acc_op = 0, acc_b = 0;
    const size_t plane = (size_t)a.H * a.W;

    // evolving pixel state travelling through the lanes
    v4f st = {0.f, 0.f, 0.f, 0.f};  // {ar0, ar1, T, ar2}: one register quad, ar0/ar1 an aligned pair for the packed ops
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
        const uint32_t pidx = (uint32_t)(c * 64 + lane);
        const uint32_t rel = ncp > bstart ? (ncp - bstart < 64u ? ncp - bstart : 64u) : 0u;
        const uint32_t ftag = (rel << 16) | ((pidx >> 4) << 8) | ((pidx & 15u) << 4);
        grec[c * 64 + lane] = make_float4(fg0, fg1, fg2, 0.f);
        init[lane] = make_float4(ck.y - pf.x, ck.z - pf.y, ck.x, ck.w - pf.z);  // ar0, ar1, T, ar2 (ar = checkpoint colour - final colour)
        itags[lane] = ftag;
        uint64_t active = __ballot(ncp > bstart);  // pixels that reach this bucket; the others contribute nothing here
        if (c < 3) load_chunk(c + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // two steps per trip: a DPP's destination is the register that held the injected value, so the state ping-pongs
        // between two register sets; a single-step loop pays five v_mov per step to bring it back
        while (active) {
            {
                const int sl = __builtin_ctzll(active);
                active &= active - 1;
                GS_BWD_SHIFT(sl);
                GS_BWD_STEP_BODY();
            }
            if (!active) break;
            {
                const int sl = __builtin_ctzll(active);
                active &= active - 1;
                GS_BWD_SHIFT(sl);
                GS_BWD_STEP_BODY();
            }
        }
        __builtin_amdgcn_wave_barrier();  // init[] is rewritten by the next chunk only after its last read above
    }
    // drain: the last injected pixel still has to pass the bucket's remaining (valid) lanes
    const int nvalid = (n - bstart) < (uint32_t)GS_BUCKET ? (int)(n - bstart) : GS_BUCKET;
#pragma unroll 1
    for (int dr = 1; dr < nvalid; dr++) {
        GS_BWD_SHIFT_ZERO();
        GS_BWD_STEP_BODY();
    }

    if (valid) {
        float4* o = a.partials + 3 * (size_t)slot;
        if constexpr (STRICT) {  // every factor was applied term by term, as the reference does
            o[0] = make_float4(acc_m.x, acc_m.y, acc_cxy.x, acc_cxy.y);
            o[1] = make_float4(acc_cw, acc_op, acc_rg.x, acc_rg.y);
        } else {
            const float sx = -0.5f * (float)a.W, sy = -0.5f * (float)a.H;  // -(...) * ddelx_dx, ddelx_dx = 0.5 W (backward.cu:464-465)
            o[0] = make_float4(acc_m.x * sx, acc_m.y * sy, -0.5f * acc_cxy.x, -0.5f * acc_cxy.y);
            o[1] = make_float4(-0.5f * acc_cw, acc_op, acc_rg.x, acc_rg.y);
        }
        o[2] = make_float4(acc_b, 0.f, 0.f, 0.f);
    }
}

int launch_render_fwd(const RenderFwdArgs& a, hipStream_t s)
{
    if (g_strict_math) GS_LAUNCH(K_RENDER_FWD, render_fwd_kernel<true>, dim3(a.gx * a.gy), dim3(64), 0, s, a);
    else GS_LAUNCH(K_RENDER_FWD, render_fwd_kernel<false>, dim3(a.gx * a.gy), dim3(64), 0, s, a);
    return GSLIC_OK;
}
int launch_render_bwd(const RenderBwdArgs& a, hipStream_t s)
{
    if (a.B <= 0) return GSLIC_OK;
    const dim3 grid((a.B + BWD_WAVES - 1) / BWD_WAVES), block(64 * BWD_WAVES);
    if (g_strict_math) GS_LAUNCH(K_RENDER_BWD, render_bwd_kernel<true>, grid, block, 0, s, a);
    else GS_LAUNCH(K_RENDER_BWD, render_bwd_kernel<false>, grid, block, 0, s, a);
    return GSLIC_OK;
}

}  // namespace gslic
