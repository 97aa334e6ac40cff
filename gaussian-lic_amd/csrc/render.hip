// render.hip — the two alpha-blend kernels, designed for 64-lane waves.
//
// render_fwd_kernel<STRICT, SPLIT>  replaces renderCUDA<3> (forward.cu:321-481)
//   SPLIT WAVES PER 16x16 TILE (default 2), each blending 4 / SPLIT of the tile's four 8x8 pixel quadrants per lane (lane l owns
//   pixel (l & 7, l >> 3) of every quadrant of its wave: quadrants cull better than 16x4 strips, -9 % pairs evaluated).  A wave fetches 64 list entries at a time (one 48-byte record per lane, three dwordx4 loads),
//   pre-scales them, parks them in LDS and fetches entry j with three ds_read_b128 at a wave-uniform address (LDS broadcast, next
//   entry prefetched): the kernel is VALU-issue bound and LDS reads cost no issue slot.  A 4-bit mask per entry says which quadrants
//   it can reach at all; a checkpoint {T, C} per pixel is stored at every 64th entry (bucket = one wave of entries) as one coalesced
//   1-KiB dwordx4 store per quadrant.  One wave per workgroup: no barrier anywhere.
//
// render_bwd_kernel<BITS>  replaces PerGaussianRenderCUDA<3> (backward.cu:379-597)
//   ONE WAVE PER BUCKET of 64 list entries: lane = Gaussian, the tile's pixels stream through the lanes as a 64-deep systolic
//   pipeline.  Two values per pixel, {T, A = ar . dL/dpixel}, move lane -> lane+1 with one v_mov_b32 DPP wave_shr:1 each; the
//   schedule is static, so every lane fetches the record {dL/dpixel, tag} of the pixel it holds from LDS at a per-lane address one
//   step ahead.  Only pixels whose n_contrib reaches this bucket are injected, deepest first.  Each lane accumulates its Gaussian's
//   nine 2D gradients in registers and writes them ONCE to its emission slot (plain 48-byte store): no atomics — the sum over a
//   Gaussian's tiles is a contiguous segmented reduction in preprocess_bwd_kernel, deterministic run to run.  (Details above the kernel.)
#include "gslic_common.h"
#include "kernels.h"
#include <stdlib.h>
#include <mutex>

namespace gslic {

// Upper bound of p2(x, y) = hA dx^2 + hC dy^2 + nB dx dy (dx = gx - x, dy = gy - y; a negative-definite form scaled by log2 e)
// over the pixel rectangle [x0, x1] x [y0, y1], plus a rounding margin: the maximum of a concave quadratic over a rectangle that
// does not contain its centre lies on an edge facing the centre, where it is a 1-D parabola.  Used to skip whole 16x4 pixel
// quadrants that an entry cannot reach (alpha < 1/255 everywhere): conservative, so the image is unchanged bit for bit.
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
// SPLIT = waves per tile (1, 2 or 4): each takes 4 / SPLIT of the tile's four 8x8 quadrants.  The kernel's duration is set by its
// longest tiles (one wave alone on a SIMD issues a VALU instruction every 4.3 cycles, half the rate of a busy SIMD), so the default
// splits every tile over two waves: +9 % instructions (the per-entry setup is repeated), half the latency of a long tile.
// Timing by elimination (results are WRONG with any bit set; the counterpart of GS_SCAN_SKIP in render_bwd_scan.hip): 1 = no entry is blended,
// 2 = no checkpoints stored, 4 = no decision masks stored, 8 = no records gathered.  (The conditions also test a kernel argument that is never
// negative, so the code stays in the binary.)
#ifndef GS_FWD_SKIP
#define GS_FWD_SKIP 0
#endif
#ifndef GS_FWD_TAIL4_DEFAULT
#define GS_FWD_TAIL4_DEFAULT 0.10f   // measured at 2M / 1080p: 0 -> 0.3496 ms, 0.05 -> 0.339, 0.08 -> 0.332, 0.10 -> 0.330, 0.125 -> 0.330-0.333, 0.15 -> 0.333, 0.25 -> 0.347, 1.0 -> 0.392 (profiles/r06u_fwd_tail4_*.log)
#endif
#ifndef GS_FWD_WAVES
#define GS_FWD_WAVES 7   // (8: 64 VGPRs, 12 spilled — 0.349 ms either way, profiles/r06n_occupancy_others_ab.log)
#endif
template <bool STRICT, int SPLIT>
__global__ __launch_bounds__(64, SPLIT >= 2 ? GS_FWD_WAVES : 5) void render_fwd_kernel(RenderFwdArgs a)
{
    __shared__ float4 s_rec[3 * GS_BUCKET];
    // 1-D grid in groups of 8 * SPLIT workgroups: workgroup b of a group works on tile 8 * group + b % 8, quadrants (b / 8) * QN...:
    // consecutive workgroups go to consecutive XCDs (eight L2s), so the SPLIT waves of one tile land on the SAME XCD, a few dispatches
    // apart — the second wave's fetch of the tile's records hits the L2 the first one filled (render_fwd 0.30 -> 0.265 ms).
    const uint32_t grp = blockIdx.x / (8u * SPLIT), rem = blockIdx.x % (8u * SPLIT);
    const int tile = (int)(grp * 8u + (rem & 7u));
    if (tile >= a.gx * a.gy) return;
    const int q0 = (int)(rem >> 3) * (4 / SPLIT);   // first quadrant of this wave
#define GS_BODY_SPLIT SPLIT
#include "render_fwd_body.inc"
#undef GS_BODY_SPLIT
}

// TAIL: the tiles in front of a.tail4_from are blended by two waves (two quadrants each), the tiles from tail4_from on by four, one quadrant each.
// A wave of the two-per-tile launch lives for 0.44 of the kernel's duration (16 320 waves over 7168 slots at 2M / 1080p), so the launch ends in a
// long drain at falling occupancy — and the kernel loses 18 % from seven waves per SIMD to five (profiles/r06r_blend_occupancy_sensitivity.log);
// the tiles dispatched last ARE the drain, and with half the work per wave it is half as long.  What a pixel computes does not depend on which
// wave owns it: the image is unchanged bit for bit.  Both bodies in one kernel behind a wave-uniform branch, so that neither pays for the other's
// registers.  Workgroups [0, 2 tail4_from): the two-per-tile mapping of render_fwd_kernel; behind them four per tile (tail4_from is a multiple of
// eight, so a workgroup's XCD — blockIdx % 8 — is its tile's in both parts).  No workgroup of the grid is empty: a first version that launched four
// per tile everywhere and let two of them leave ran the two-wave tiles on HALF the machine (the dispatcher's workgroup -> CU pattern is periodic:
// 0.35 -> 0.52 ms).
template <bool STRICT>
__global__ __launch_bounds__(64, GS_FWD_WAVES) void render_fwd_tail_kernel(RenderFwdArgs a)
{
    __shared__ float4 s_rec[3 * GS_BUCKET];
    const uint32_t n2 = 2u * (uint32_t)a.tail4_from;
    if (blockIdx.x < n2) {
        const uint32_t grp = blockIdx.x / 16u, rem = blockIdx.x % 16u;
        const int tile = (int)(grp * 8u + (rem & 7u));
        const int q0 = (int)(rem >> 3) * 2;
#define GS_BODY_SPLIT 2
#include "render_fwd_body.inc"
#undef GS_BODY_SPLIT
    } else {
        const uint32_t bb = blockIdx.x - n2, grp = bb / 32u, rem = bb % 32u;
        const int tile = a.tail4_from + (int)(grp * 8u + (rem & 7u));
        if (tile >= a.gx * a.gy) return;
        const int q0 = (int)(rem >> 3);
#define GS_BODY_SPLIT 4
#include "render_fwd_body.inc"
#undef GS_BODY_SPLIT
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define GS_PK_FMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))
#define GS_SPLAT(x) ((v2f){(x), (x)})

// =========================================================================================================
// Backward: ONE WAVE PER BUCKET, lane = Gaussian, the tile's pixels stream through the lanes.  The two modes (gslic_set_math_mode) differ in
// where the blend / skip decision of a pair comes from (BITS, below), not in the pipeline or in the arithmetic of a contribution.
//
// What travels lane -> lane+1 is {T, A}: A = sum_ch ar[ch] * dL/dpixel[ch] replaces the colour vector ar[3] of the reference's
// formulation (dL/dalpha only ever needs that dot product: dL/dalpha = A' / (1 - alpha) + T (c . g), A' = A + T alpha (c . g)): two
// DPP moves per step instead of five.  Which pixel a lane works on does not travel at all: the schedule is static — lane L at step s
// holds the pixel injected at step s - L — so the pixel's data {dL/dpixel, tag} and start state {T, A} are stored in LDS in INJECTION
// order (three float2 arrays) and every lane fetches entry s - L with three ds_read_b64 at ONE per-lane byte offset that advances by
// 8 per step, one step ahead of its use (64 consecutive entries per read: no bank conflicts, no dependence on anything computed in the
// step); lane 0's {T, A} is the injection.  alpha = min(0.99, exp2(p2 + log2 opacity)) with the SAME operation sequence as render_fwd
// (identical bits on both sides, so both take the same alpha < 1/255 decisions); the products with G = exp(power) that the reference
// forms are written on a = opacity * G, dL/dopacity is divided by the opacity once per instance, and the conic factors of dL/dmean2D
// (per-Gaussian constants) are applied once to sum w d instead of in every step.
//
// The step is STRAIGHT-LINE code: no exec-mask branches.  A (pixel, Gaussian) pair that does not blend (lane >= rel, power > 0,
// alpha < 1/255, empty slot) runs the same instructions with alpha and w forced to zero, which leaves every sum and the travelling
// state unchanged bit for bit.  Measured on gfx950 (tools/ubench/issue_rate): a wave64 VALU instruction issues every 2.1-2.4 cycles
// when all its operands are VGPRs, every 4.2 with an SGPR / literal operand, as a DPP move or as a packed op, every 8 for v_exp / v_rcp,
// and one wave alone issues at most every 4.3 cycles — so the per-step cost is set by the dependent chain of each wave (SALU, taken
// branches, LDS round trips) as much as by the VALU count: branch-free, the pixel's constants are fetched from LDS at the top of the
// step, the next injection is fetched one step ahead and enters through the DPP's `old` operand, and the loop constants live in VGPRs.
//
// tag = rel << 16 | py << 8 | 16 px (rel = min(n_contrib - bucket start, 64) >= 1 for an injected pixel; the two low bytes are what
// v_cvt_f32_ubyte0/1 turn into coordinates), 0 for an empty slot; kcmp < tag <=> lane < rel.
// The offset is clamped to [0, ninj] (one v_med3_i32): before the first pixel reaches lane L (s < L) the lane re-reads entry 0 while its
// travelling state is still T = A = 0, which makes every product of the step an exact zero whatever the entry says (alpha <= 0.99 keeps
// 1 / (1 - alpha) finite); behind the last pixel every lane reads entry ninj, all zeros (tag 0: nothing blends; T = A = 0 injected).
// No slack entries, so the three arrays take 3 x 257 x 8 = 6168 bytes per wave (26 waves per CU; the unclamped layout with 64 spare
// entries at either end took 7680: 21).
// BITS (the strict mode, gslic_set_math_mode(1), the default): which (pixel, Gaussian) pairs blend is not re-derived at all — the strict forward
// recorded, per bucket and pixel, the 64-bit mask of the bucket's list entries the pixel blended (SampleState::hit, pixel-major like the
// checkpoints), and those bits ARE the reference's decisions (lane < n_contrib - bucket start, power <= 0, alpha >= 1/255:
// backward.cu:538-546), since the strict forward is bit-identical to the reference's.  The mask travels with the pixel's other data (a fourth
// float2-sized array at the same clamped offset) and lane L turns bit L of it into an AND mask: v_bfi (the dword of its half), v_bfe_i32, v_and.  What a pair
// CONTRIBUTES is computed with the arithmetic described above in both modes (within a few ulp of the reference's; gradients are sums of
// 1e2..1e4 such terms in an order that differs from the reference's atomics anyway).  A forward that recorded no bits (fast mode, or the mode
// was switched in between) leaves status[GS_FLAG_HITBITS] = 0 and the kernel re-derives the decisions like the fast variant.
struct BwdLane {
    v2f d0, hAC, col_rg;          // centre relative to the tile origin; log2(e)-scaled conic diagonal {-1/2 A, -1/2 C}; colour r, g
    float nB, lop, colb;          // log2(e)-scaled -B; log2(opacity); colour b
};

// DST <- shift(SRC) with lane 0 <- INJ.  The DPP's destination is the register that held the injected value, so two steps per trip
// with the roles of the two register sets swapped need no copies at all.
// (T, A, tag) <- shift of the current state with lane 0 <- the fetched injection.  Written as asm so that the destination IS the register
// that held the injection (with the builtin the compiler copied it first); bound_ctrl off: lane 0 has no source lane and keeps the
// destination's value.  s_nop 1: a DPP source written by the preceding VALU instruction needs two wait states.
#define GS_BW_SHIFT_INJ(IT, IA, ST, SA)                                                                              \
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_mov_b32_dpp %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf"                                           \
                 : "+v"(IT), "+v"(IA) : "v"(ST), "v"(SA))
#define GS_BW_PREFETCH(USE_BITS, NT, NA, NR, NH)                                                                     \
    do { /* entry (step + 1) - lane of the arrays, clamped to [0, ninj]: entry 0 until the lane's first pixel arrives (its state is still */ \
         /* T = A = 0, see above), the all-zero entry ninj once the last pixel has passed; lane 0's {T, A} is the next injection */          \
        uint32_t oc_;                                                                                                \
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(oc_) : "v"(off), "v"(kzero_i), "v"(khi)); /* (the compiler emits min + cmp + select) */ \
        const float2 rg_ = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(smem) + oc_);              \
        const float2 bt_ = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(smem) + BW_OFF_BT + oc_);   \
        const float2 ta_ = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(smem) + BW_OFF_TA + oc_);   \
        if (USE_BITS) NH = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(smem) + BW_OFF_HM + oc_);    \
        NR = make_float4(rg_.x, rg_.y, bt_.x, bt_.y);                                                                \
        NT = ta_.x; NA = ta_.y;                                                                                      \
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(off) : "v"(keight)); /* (left to the compiler: three adds and a copy per two steps) */ \
        __builtin_amdgcn_sched_barrier(0); /* keep the LDS reads at the top of the step: a whole step passes before they are used */ \
    } while (0)
#define GS_BW_BODY(USE_BITS, T_, A_, GR, GH)                                                                         \
    do {                                                                                                             \
        const float4 gr = GR;                                                                                        \
        const uint32_t TAG = __float_as_uint(gr.w);                                                                  \
        const v2f pxy16 = {(float)(TAG & 0xffu), (float)((TAG >> 8) & 0xffu)}; /* v_cvt_f32_ubyte0 / ubyte1: {16 px, py} */ \
        const v2f d = GS_PK_FMA(pxy16, kneg, L.d0); /* exact: d0 - {px, py} */                                       \
        float p2 = __builtin_fmaf(L.hAC.x * d.x, d.x, L.lop); /* the operation sequence of render_fwd<false> */      \
        p2 = __builtin_fmaf(L.hAC.y * d.y, d.y, p2);                                                                 \
        p2 = __builtin_fmaf(L.nB * d.x, d.y, p2); /* = log2(e) * power + log2(opacity) */                            \
        const float araw = __builtin_amdgcn_exp2f(p2); /* opacity * G */                                             \
        float ah; /* alpha before the 0.99 cap if the pair blends, else 0: masks both alpha and the gradient weight */ \
        if (USE_BITS) { /* bit `lane` of the pixel's mask (the strict forward blended this pair) as an all-ones / all-zeros word: the dword   */ \
            uint32_t sel_, m_; /* of this lane's half (v_bfi), the bit sign-extended (v_bfe_i32), one AND - no compare, no lane-mask operand */ \
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sel_) : "v"(klo), "v"(GH.x), "v"(GH.y));                            \
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m_) : "v"(sel_), "v"(kbit));                                         \
            ah = __uint_as_float(__float_as_uint(araw) & m_);                                                        \
        } else { /* lane < n_contrib - bucket start (backward.cu:538), power <= 0, alpha >= 1/255 (:543-546; min(0.99, a) < 1/255 iff a < 1/255) */ \
            const bool hit = (kcmp < TAG) & !(p2 > L.lop) & !(araw < c255);                                          \
            ah = hit ? araw : 0.0f;                                                                                  \
        }                                                                                                            \
        const float alpha = __builtin_amdgcn_fmed3f(ah, ninf, c099); /* min(0.99, .) without the canonicalising v_max fminf costs */ \
        const float om = 1.0f - alpha;                                                                               \
        const float rinv = __builtin_amdgcn_rcpf(om); /* 1 / (1 - alpha) scales a gradient term, it decides nothing (<= 1 ulp) */   \
        const float Ta = T_ * alpha;                                                                                 \
        float cg = L.col_rg.x * gr.x;                                                                                \
        cg = __builtin_fmaf(L.col_rg.y, gr.y, cg);                                                                   \
        cg = __builtin_fmaf(L.colb, gr.z, cg); /* c . dL/dpixel */                                                   \
        acc_rg = GS_PK_FMA(GS_SPLAT(Ta), ((v2f){gr.x, gr.y}), acc_rg); acc_b = __builtin_fmaf(Ta, gr.z, acc_b);      \
        A_ = __builtin_fmaf(Ta, cg, A_);                                                                             \
        const float dLda = __builtin_fmaf(rinv, A_, T_ * cg);                                                        \
        T_ *= om;                                                                                                    \
        const float w = ah * dLda; /* opacity * G * dL/dalpha = G * dL/dG (dLda is finite: alpha <= 0.99) */         \
        const v2f wd = GS_SPLAT(w) * d;                                                                              \
        acc_S += wd;                                                                                                 \
        acc_cxy = GS_PK_FMA(GS_SPLAT(wd.x), d, acc_cxy); /* -0.5 applied at the end */                               \
        acc_cw = __builtin_fmaf(wd.y, d.y, acc_cw);                                                                  \
        acc_op += w; /* divided by the opacity at the end */                                                         \
    } while (0)

template <bool BITS>
__global__ __launch_bounds__(64) void render_bwd_kernel(RenderBwdArgs a)
{
    // Three float2 arrays of 256 + 1 entries in injection order — {dL/dpixel.r, .g}, {dL/dpixel.b, tag}, {T, A} at the start of this bucket
    // (BITS: a fourth with the pixels' decision masks) — read with ONE per-lane byte offset clamped to [0, ninj]: entry ninj is all zeros
    // (what the pipeline takes in while the last pixels drain); no slack entries in front or behind.  6168 bytes per wave: 26 workgroups per CU
    // (the 7.5 KB of the unclamped layout allowed 21).
    constexpr int NENT = GS_TILE_PIX + 1;
    constexpr uint32_t BW_OFF_BT = NENT * 8u, BW_OFF_TA = 2u * NENT * 8u, BW_OFF_HM = 3u * NENT * 8u;
    __shared__ float2 smem[(BITS ? 4 : 3) * NENT];   // BITS: + the pixels' recorded 64-bit decision masks (8224 bytes per wave: 19 per CU)
    float2* const s_rg = smem;
    float2* const s_bt = smem + NENT;
    float2* const s_ta = smem + 2 * NENT;
    uint2* const s_hm = reinterpret_cast<uint2*>(smem + 3 * NENT);   // (BITS only)
    const int lane = threadIdx.x;
    // Workgroup i runs on XCD i % 8 (eight private L2s).  The buckets of a tile are consecutive and all read the tile's pix_final, dL_dpixel
    // and the forward's per-tile data: runs of 2^xcd_lg consecutive buckets go to the same XCD, back to back, so that only the first
    // bucket of a tile misses the L2 (a bijection on [0, gridDim.x), gridDim.x a multiple of 8 << xcd_lg).
    uint32_t bucket = blockIdx.x;
    if (a.xcd_lg >= 0) {
        const uint32_t x = bucket & 7u, j = bucket >> 3;
        bucket = ((((j >> a.xcd_lg) << 3) + x) << a.xcd_lg) + (j & ((1u << a.xcd_lg) - 1u));
    }
    if (a.status[2] != 0u || bucket >= a.bucket_offsets[a.T - 1]) return;  // capacity overflow in the forward / B was a capacity
    const bool use_bits = BITS && a.status[GS_FLAG_HITBITS] != 0u;         // (wave-uniform) the forward recorded its blend decisions
    if (a.skip_if_bits && use_bits) return;                                // ... and render_bwd_scan_kernel, launched in front of this one, used them
    const uint32_t tile = a.bucket_to_tile[bucket];
    const uint2 range = a.ranges[tile];
    const uint32_t n = range.y - range.x;
    const uint32_t bbm = (tile == 0) ? 0u : a.bucket_offsets[tile - 1];
    const uint32_t bstart = (bucket - bbm) * GS_BUCKET;
    const uint32_t kit = bstart + (uint32_t)lane;  // splat index in tile
    const bool valid = kit < n;
    const uint32_t slot = valid ? a.inst_slot[range.x + kit] : 0u;

    // bucket entirely behind every pixel's last contributor (backward.cu:428): its instances' gradients are exactly zero.  One flag byte
    // per instance says so to preprocess_bwd (which then skips the row) instead of 36 bytes of zeros written here and read back there:
    // on the 2M / 1080p scene 58 % of the buckets end here
    if (bstart >= a.max_contrib[tile]) {
        if (valid) a.dead[slot] = 1;
        return;
    }

    const int tx0 = (int)(tile % (uint32_t)a.gx) * GS_TILE, ty0 = (int)(tile / (uint32_t)a.gx) * GS_TILE;
    const size_t plane = (size_t)a.H * a.W;
    // ---- the tile's 256 pixels, four per lane, fetched at once
    float4 ck[4], pf[4];
    float fg[4][3];
    bool inside[4];
    uint2 hm[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int pidx = c * 64 + lane;
        ck[c] = a.ckpt[(size_t)bucket * GS_TILE_PIX + pidx];
        hm[c] = make_uint2(0u, 0u);
        if constexpr (BITS) {
            if (use_bits) { const uint64_t m = a.hit[(size_t)bucket * GS_TILE_PIX + pidx]; hm[c] = make_uint2((uint32_t)m, (uint32_t)(m >> 32)); }
        }
        pf[c] = a.pix_final[(size_t)tile * GS_TILE_PIX + pidx];
        const int px = tx0 + tile_pix_x(pidx), py = ty0 + tile_pix_y(pidx);
        inside[c] = px < a.W && py < a.H;
        fg[c][0] = fg[c][1] = fg[c][2] = 0.f;
        if (inside[c]) {
            const size_t pid = (size_t)py * a.W + px;
            fg[c][0] = a.dL_dpix[pid]; fg[c][1] = a.dL_dpix[plane + pid]; fg[c][2] = a.dL_dpix[2 * plane + pid];
        }
    }
    // ---- this lane's Gaussian
    const float LOG2E = 1.4426950408889634f;
    BwdLane L;
    L.d0 = L.hAC = L.col_rg = (v2f){0.f, 0.f};
    L.nB = L.colb = 0.f;
    L.lop = -__builtin_inff();  // a lane without a Gaussian: alpha = exp2(-inf) = 0, contributes nothing whatever the decision bits say
    float rop = 0.f;            // 1 / opacity
    if (valid) {
        const uint32_t g = a.point_list[range.x + kit];
        const float4* rp = a.rec + GS_REC_F4 * (size_t)g;
        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
        L.d0.x = r0.x - (float)tx0; L.d0.y = r0.y - (float)ty0;
        L.hAC.x = -0.5f * LOG2E * r0.z; L.nB = -LOG2E * r0.w; L.hAC.y = -0.5f * LOG2E * r1.x;
        L.lop = __builtin_amdgcn_logf(r1.y);
        rop = r1.y > 0.f ? 1.0f / r1.y : 0.f;
        L.col_rg.x = r1.z; L.col_rg.y = r1.w; L.colb = r2.x;
    }

    // ---- injection order.  A pixel injected at position i with rel_i Gaussians of this bucket still in front of its last contributor
    // occupies the pipeline until step i + rel_i: injecting in DESCENDING order of rel makes max(i + rel_i) — the number of steps — minimal
    // (the pixels that leave the pipeline early go last, so the drain is short: -12 % steps on the 2M / 1080p scene,
    // profiles/r02_bwd_pipeline_model.txt).  Four classes of sixteen rel values, pixel order inside a class, get all but 2 % of that:
    // sixteen ballots kept in scalar registers, popcounts and v_mbcnt — no sort.
    uint32_t rel[4], key[4], pos[4];
    uint64_t cm[4][4];   // cm[c][k]: lanes whose pixel of chunk c falls into class k
    uint32_t ninj = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t ncp = inside[c] ? __float_as_uint(pf[c].w) : 0u;
        rel[c] = ncp > bstart ? (ncp - bstart < 64u ? ncp - bstart : 64u) : 0u;  // 0: the pixel does not reach this bucket
        key[c] = rel[c] ? (rel[c] - 1u) >> 4 : 4u;                               // 0..3, 4 = not injected
#pragma unroll
        for (int k = 0; k < 4; k++) cm[c][k] = __ballot(key[c] == (uint32_t)k);
    }
    {
        uint32_t run = 0;  // classes in descending order, chunks in ascending order inside a class
#pragma unroll
        for (int k = 3; k >= 0; k--)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint64_t m = cm[c][k];
                if (key[c] == (uint32_t)k) pos[c] = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                run += (uint32_t)__popcll(m);
            }
        ninj = run;
    }
    uint32_t last = 0;  // number of steps = max over the injected pixels of position + rel
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t e = rel[c] ? pos[c] + rel[c] : 0u;
        last = e > last ? e : last;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)last, d, 64);
        last = o > last ? o : last;
    }
    const uint32_t nsteps = readlane_u(last, 0);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint32_t pidx = (uint32_t)(c * 64 + lane);
        if (rel[c]) {
            float A0 = (ck[c].y - pf[c].x) * fg[c][0];  // ar = checkpoint colour - final colour (backward.cu:522-523), dotted with dL/dpixel
            A0 = __builtin_fmaf(ck[c].z - pf[c].y, fg[c][1], A0);
            A0 = __builtin_fmaf(ck[c].w - pf[c].z, fg[c][2], A0);
            s_rg[pos[c]] = make_float2(fg[c][0], fg[c][1]);
            s_bt[pos[c]] = make_float2(fg[c][2], __uint_as_float((rel[c] << 16) | ((uint32_t)tile_pix_y((int)pidx) << 8) | ((uint32_t)tile_pix_x((int)pidx) << 4)));
            if constexpr (BITS) s_hm[pos[c]] = hm[c];
            s_ta[pos[c]] = make_float2(ck[c].x, A0);
        }
    }
    if (lane == 0) {   // the drain entry (tag 0, mask 0: no pair blends; T = A = 0 injected)
        s_rg[ninj] = s_bt[ninj] = s_ta[ninj] = make_float2(0.f, 0.f);
        if constexpr (BITS) s_hm[ninj] = make_uint2(0u, 0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // loop constants in VGPRs: a literal or SGPR operand doubles the issue cost of the instruction that reads it
    float c099 = 0.99f, c255 = 1.0f / 255.0f, ninf = -__builtin_inff();
    v2f kneg = {-0.0625f, -1.0f};
    uint32_t kcmp = ((uint32_t)lane << 16) | 0xffffu;
    uint32_t kbit = (uint32_t)lane & 31u;           // BITS: this lane's bit inside ...
    uint32_t klo = lane >= 32 ? 0u : 0xffffffffu;   // ... the low (all ones) or the high (all zeros) dword of the pixel's mask
    asm volatile("" : "+v"(c099), "+v"(c255), "+v"(ninf), "+v"(kneg), "+v"(kcmp), "+v"(kbit), "+v"(klo));
    v2f acc_S = {0.f, 0.f}, acc_cxy = {0.f, 0.f}, acc_rg = {0.f, 0.f};
    float acc_cw = 0, acc_op = 0, acc_b = 0;
    // {T, A}: the state travelling through the lanes (set 1) and the injection fetched one step ahead (set 2); a step shifts set 1
    // into set 2 (whose lane 0 holds the injection), so the sets swap roles every step: two steps per trip, no register copies.  The
    // records alternate between Ra and Rb the same way.
    float T1 = 0.f, A1 = 0.f, T2, A2;
    float4 Ra, Rb;
    uint2 Ha = make_uint2(0u, 0u), Hb = make_uint2(0u, 0u);
    int off = -8 * lane, kzero_i = 0, khi = 8 * (int)ninj, keight = 8;   // byte offset of entry (0 - lane); the clamp bounds; all in VGPRs
    asm volatile("" : "+v"(off), "+v"(kzero_i), "+v"(khi), "+v"(keight));
    // Two steps per trip, an even number of steps: one step more than needed is harmless — every lane then holds a pixel at or behind its
    // last contributor in this bucket (or the all-zero drain entry), which blends nothing — and the trip needs no exit test in its middle.
#define GS_BW_LOOP(USE_BITS)                                                                                         \
    GS_BW_PREFETCH(USE_BITS, T2, A2, Ra, Ha);                                                                        \
    for (uint32_t sidx = 0; sidx < nsteps; sidx += 2) {                                                              \
        GS_BW_SHIFT_INJ(T2, A2, T1, A1); /* set 2 = state */                                                         \
        GS_BW_PREFETCH(USE_BITS, T1, A1, Rb, Hb);                                                                    \
        GS_BW_BODY(USE_BITS, T2, A2, Ra, Ha);                                                                        \
        GS_BW_SHIFT_INJ(T1, A1, T2, A2); /* set 1 = state */                                                         \
        GS_BW_PREFETCH(USE_BITS, T2, A2, Ra, Ha);                                                                    \
        GS_BW_BODY(USE_BITS, T1, A1, Rb, Hb);                                                                        \
    }
    if (use_bits) { GS_BW_LOOP(true) } else { GS_BW_LOOP(false) }
#undef GS_BW_LOOP

    if (valid) {
        // acc_S = sum of w d (w = G dL/dG): dL/dmean2D = -(0.5 W, 0.5 H) o (A S.x + B S.y, C S.y + B S.x) (backward.cu:566-573), written on the
        // log2(e)-scaled conic the lane holds: A = -2 hA / log2 e, B = -nB / log2 e
        const float kx = 0.5f * (float)a.W / LOG2E, ky = 0.5f * (float)a.H / LOG2E;
        const float gx = __builtin_fmaf(2.0f * L.hAC.x, acc_S.x, L.nB * acc_S.y) * kx;
        const float gy = __builtin_fmaf(2.0f * L.hAC.y, acc_S.y, L.nB * acc_S.x) * ky;
        // nine floats, one 36-byte row per instance (dword-aligned wide stores; plain, not non-temporal: rows at scattered slots need the
        // L2 to merge them — non-temporal: 0.59 -> 0.94 ms).  acc_op = sum of opacity * G * dL/dalpha (backward.cu:580 sums G * dL/dalpha)
        float* o = a.partials + GS_PROW * (size_t)slot;
        *reinterpret_cast<gs_v4f_u*>(o) = (gs_v4f_u){gx, gy, -0.5f * acc_cxy.x, -0.5f * acc_cxy.y};
        *reinterpret_cast<gs_v4f_u*>(o + 4) = (gs_v4f_u){-0.5f * acc_cw, acc_op * rop, acc_rg.x, acc_rg.y};
        o[8] = acc_b;
    }
}

// Host-side note of which arithmetic the last forward into a sample buffer ran in (key: the address of its decision masks).  The device flag
// status[GS_FLAG_HITBITS] stays the kernels' source of truth; the note only lets the strict backward skip the launch of its fallback kernel
// when it is known that the masks exist (unknown buffer: both kernels are launched and one of them leaves at once).
static std::mutex g_fwd_note_mu;
static struct { const void* hit; bool strict; } g_fwd_note[64];
static unsigned g_fwd_note_next = 0;
static void note_forward(const void* hit, bool strict)
{
    std::lock_guard<std::mutex> lk(g_fwd_note_mu);
    for (auto& e : g_fwd_note)
        if (e.hit == hit) { e.strict = strict; return; }
    g_fwd_note[g_fwd_note_next++ & 63u] = {hit, strict};
}
static bool forward_known_strict(const void* hit)
{
    std::lock_guard<std::mutex> lk(g_fwd_note_mu);
    for (auto& e : g_fwd_note)
        if (e.hit == hit) return e.strict;
    return false;
}

int launch_render_fwd(const RenderFwdArgs& a, hipStream_t s)
{
    if (!a.no_color && a.hit) note_forward(a.hit, g_strict_math != 0);
    // waves per tile: two halve the serial chain of a tile's wave, at the price of both fetching the tile's records — which the XCD-aware
    // grid turns into an L2 hit.  Measured (profiles/r03h_fwd_split_ab.log): 1080p strict 0.364 (2 waves) vs 0.394 (4), fast 0.259 vs 0.291; 4K
    // (32 400 tiles) strict 1.22 (2) vs 1.29 (1), fast 0.865 vs 0.875.  GSLIC_FWD_SPLIT pins it.
    static const int forced = [] { const char* e = getenv("GSLIC_FWD_SPLIT"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
    const unsigned T = (unsigned)(a.gx * a.gy);
    const int split = forced ? forced : 2;
    const unsigned groups = (T + 7u) / 8u;   // groups of 8 tiles x split waves (the kernel's blockIdx -> (tile, quadrants) mapping)
    // GSLIC_FWD_TAIL4 = the fraction of the tiles (the last ones of the grid) that four waves blend instead of two (0 = none)
    static const float tail4 = [] { const char* e = getenv("GSLIC_FWD_TAIL4"); const float v = e ? (float)atof(e) : GS_FWD_TAIL4_DEFAULT; return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }();
    if (split == 2 && tail4 > 0.f) {
        RenderFwdArgs b = a;
        // the drain is about half a round of the chip's wave slots (256 CUs x 28) whatever the number of tiles: the default fraction is capped at
        // 896 tiles = 3584 four-wave workgroups (at 4K, 32 400 tiles, a tenth of the tiles costs 0.9 % instead of gaining: profiles/r06v_*)
        static const bool tail4_env = getenv("GSLIC_FWD_TAIL4") != nullptr;
        unsigned tail_tiles = (unsigned)(tail4 * (float)T);
        if (!tail4_env && tail_tiles > 896u) tail_tiles = 896u;
        b.tail4_from = (int)(T - tail_tiles) & ~7;   // whole groups of eight tiles
        const unsigned grid = 2u * (unsigned)b.tail4_from + 32u * ((T - (unsigned)b.tail4_from + 7u) / 8u);
        if (g_strict_math) GS_LAUNCH(K_RENDER_FWD, (render_fwd_tail_kernel<true>), dim3(grid), dim3(64), 0, s, b);
        else GS_LAUNCH(K_RENDER_FWD, (render_fwd_tail_kernel<false>), dim3(grid), dim3(64), 0, s, b);
        return GSLIC_OK;
    }
    if (g_strict_math && split == 1) GS_LAUNCH(K_RENDER_FWD, (render_fwd_kernel<true, 1>), dim3(groups * 8u), dim3(64), 0, s, a);
    else if (g_strict_math && split == 4) GS_LAUNCH(K_RENDER_FWD, (render_fwd_kernel<true, 4>), dim3(groups * 32u), dim3(64), 0, s, a);
    else if (g_strict_math) GS_LAUNCH(K_RENDER_FWD, (render_fwd_kernel<true, 2>), dim3(groups * 16u), dim3(64), 0, s, a);   // per-pixel arithmetic does not depend on the split
    else if (split == 1) GS_LAUNCH(K_RENDER_FWD, (render_fwd_kernel<false, 1>), dim3(groups * 8u), dim3(64), 0, s, a);
    else if (split == 4) GS_LAUNCH(K_RENDER_FWD, (render_fwd_kernel<false, 4>), dim3(groups * 32u), dim3(64), 0, s, a);
    else GS_LAUNCH(K_RENDER_FWD, (render_fwd_kernel<false, 2>), dim3(groups * 16u), dim3(64), 0, s, a);
    return GSLIC_OK;
}
int launch_render_bwd(const RenderBwdArgs& a, hipStream_t s)
{
    if (a.B <= 0) return GSLIC_OK;
    // GSLIC_BWD_XCD_RUN = run length of consecutive buckets per XCD (a power of two; 0 = plain blockIdx order)
    static const int xcd_lg = [] {
        const char* e = getenv("GSLIC_BWD_XCD_RUN");
        const int v = e ? atoi(e) : 16;
        int lg = -1;
        for (int k = 0; k < 12; k++) if (v == (1 << k)) lg = k;
        return lg;
    }();
    // GSLIC_BWD_SCAN = 0: the pipeline kernel in the strict mode too (A/B runs)
    static const bool use_scan = [] { const char* e = getenv("GSLIC_BWD_SCAN"); return !(e && atoi(e) == 0); }();
    RenderBwdArgs b = a;
    b.xcd_lg = xcd_lg;
    b.skip_if_bits = 0;
    unsigned grid = (unsigned)a.B;
    if (xcd_lg >= 0) { const unsigned unit = 8u << xcd_lg; grid = (grid + unit - 1u) / unit * unit; }
    if (g_strict_math && use_scan) {
        // the row-scan kernel works from the decision masks of a strict forward; when the forward recorded none (the mode was switched in
        // between) every one of its workgroups leaves at once and the pipeline kernel behind it re-derives the decisions instead
        launch_render_bwd_scan(b, grid, s);
        if (forward_known_strict(a.hit)) return GSLIC_OK;
        b.skip_if_bits = 1;
    }
    if (g_strict_math) GS_LAUNCH(K_RENDER_BWD, render_bwd_kernel<true>, dim3(grid), dim3(64), 0, s, b);
    else GS_LAUNCH(K_RENDER_BWD, render_bwd_kernel<false>, dim3(grid), dim3(64), 0, s, b);
    return GSLIC_OK;
}

}  // namespace gslic
