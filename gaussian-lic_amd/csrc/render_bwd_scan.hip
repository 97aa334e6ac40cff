// render_bwd_scan.hip — the blend backward of the strict (default) arithmetic as ROW SCANS instead of a 64-deep pipeline
// (replaces PerGaussianRenderCUDA<3>, backward.cu:379-597; the pipeline of render.hip stays the kernel of the fast arithmetic and the fallback
// for a forward that recorded no decision masks).
//
// Why a second decomposition.  The pipeline (lane = list entry, pixels stream lane -> lane + 1) pays 64 fill / drain steps per bucket and
// walks every injected pixel past all 64 entries, although a pixel blends 18 of them on average (profiles/r04b_bwd_hitmask_model.txt: 23 % of
// its (pixel, entry) slots do any work, and deriving injection from the recorded masks instead of n_contrib saves 0.5 % of its steps).  The
// strict forward recorded, per bucket and pixel, WHICH entries the pixel blended (SampleState::hit) — the reference's own decisions
// (backward.cu:538-546), since that forward is bit-identical to the reference's.  Those masks make the work list explicit:
//
//   * per 8x8 pixel quadrant only the entries SOME pixel of the quadrant blended are visited (28 of 64 on the 2M / 1080p scene): the
//     OR of the quadrant's masks, compacted into groups of 16;
//   * only pixels with a non-empty mask are visited;
//   * a wave works on 16 entries x 4 pixels per step: lane = (entry slot i = lane & 15, pixel row r = lane >> 4).  The transmittance
//     T_i = T_in prod_{j<i} (1 - alpha_j) and the colour-behind term A_i = A_in + sum_{j<=i} T_j alpha_j (c_j . dL/dpixel) are LOG-STEP
//     SCANS over the 16 lanes of a DPP row (row_shr:1/2/4/8) — no fill, no drain; the state {T, A} of a pixel is handed from one entry
//     group to the next through lane 15 of its row (row_newbcast:15) in registers.
//
// 57.7 % of the pipeline's steps on the default scene, 58.4 % on the faint one (model: tools/bwd_hitmask_model.py).  What a pair CONTRIBUTES
// is computed with the pipeline's formulas (GS_BW_BODY of render.hip): dL/dalpha = T (c . g) + A' / (1 - alpha), the nine sums of
// backward.cu:548-590; only the association of the products inside T and of the sums inside A differs (a scan tree instead of a chain).
// Deterministic run to run: no atomics, every sum has a fixed order.
#include "gslic_common.h"
#include "kernels.h"
#include <stdlib.h>

// Timing by elimination (results are WRONG with any bit set; profiles/r04y_bwd_scan_by_elimination.log): 1 = no block passes, 2 = no quadrant loop at
// all, 4 = block passes without their chunk loops, 8 = dead buckets do not flag their instances, 16 = no partial rows stored, 32 = no records gathered.
// (the conditions also test a kernel argument that is never negative, so the code stays in the binary)
#ifndef GS_SCAN_SKIP
#define GS_SCAN_SKIP 0
#endif

namespace gslic {

typedef float v2f_s __attribute__((ext_vector_type(2)));

// DPP controls (row = 16 lanes): row_shr:n = 0x110 + n, row_newbcast:n = 0x150 + n, row_bcast15 = 0x142, row_bcast31 = 0x143
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t src)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
#ifdef GS_SCAN_SAFE   // A/B and debugging: the same primitives through ds_bpermute (no DPP idiom to get wrong)
__device__ __forceinline__ float row_scan_mul(float x)
{
    const int i = lane_id() & 15;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const float o = __shfl_up(x, d, 16); if (i >= d) x *= o; }
    return x;
}
__device__ __forceinline__ float row_scan_add(float x)
{
    const int i = lane_id() & 15;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const float o = __shfl_up(x, d, 16); if (i >= d) x += o; }
    return x;
}
__device__ __forceinline__ float row_last(float x) { return __shfl(x, 15, 16); }
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d, 64);
    return readlane_u(v, 0);
}
#else
#ifndef GS_DPP_WAIT
#define GS_DPP_WAIT "s_nop 1\n\t"   // (timing experiment: with "" the results are WRONG and the kernel is no faster — the wait states hide behind the other waves)
#endif
// inclusive product / sum over the 16 lanes of a row (lanes without a source keep the identity)
// (the product scan is written out: with the builtin the compiler materialises the identity 1.0 for every step — v_mov + v_mov_dpp + v_mul —
// where the add scan folds into v_add_f32_dpp with bound_ctrl; in place, a lane without a source lane is not written and keeps its own value.
// s_nop 1: a DPP source written by the preceding VALU instruction needs two wait states)
__device__ __forceinline__ float row_scan_mul(float x)
{
    asm(GS_DPP_WAIT "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        GS_DPP_WAIT "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        GS_DPP_WAIT "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        GS_DPP_WAIT "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
        : "+v"(x));
    return x;
}
__device__ __forceinline__ float row_scan_add(float x)
{
    asm(GS_DPP_WAIT "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        GS_DPP_WAIT "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        GS_DPP_WAIT "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        GS_DPP_WAIT "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
        : "+v"(x));
    return x;
}
__device__ __forceinline__ float row_last(float x) { return dpp_f<0x15f>(x, x); }   // lane 15 of the row, to all 16
// OR over the 64 lanes, returned as a wave-uniform value
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v |= dpp_u<0x111>(0u, v);
    v |= dpp_u<0x112>(0u, v);
    v |= dpp_u<0x114>(0u, v);
    v |= dpp_u<0x118>(0u, v);            // lane 15 of each row: the row's OR
    v |= dpp_u<0x142, 0xa>(0u, v);       // row_bcast15 into rows 1 and 3
    v |= dpp_u<0x143, 0xc>(0u, v);       // row_bcast31 into rows 2 and 3: lane 63 holds the wave's OR
    return readlane_u(v, 63);
}
#endif

// OR over lanes 0..31 and over lanes 32..63 (wave-uniform results)
__device__ __forceinline__ void half_or(uint32_t v, uint32_t& lo_half, uint32_t& hi_half)
{
#ifdef GS_SCAN_SAFE
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d, 64);
    lo_half = readlane_u(v, 0); hi_half = readlane_u(v, 32);
#else
    v |= dpp_u<0x111>(0u, v);
    v |= dpp_u<0x112>(0u, v);
    v |= dpp_u<0x114>(0u, v);
    v |= dpp_u<0x118>(0u, v);            // lane 15 of each row: the row's OR
    v |= dpp_u<0x142, 0xa>(0u, v);       // row_bcast15 into rows 1 and 3: lanes 31 and 63 hold their halves' ORs
    lo_half = readlane_u(v, 31); hi_half = readlane_u(v, 63);
#endif
}

struct ScanEntry {     // one list entry of the bucket, as the pipeline's BwdLane holds it
    float d0x, d0y;    // centre relative to the tile origin
    float hA, hC, nB;  // log2(e)-scaled conic: -1/2 A, -1/2 C, -B
    float lop;         // log2(opacity); -inf for an empty slot (alpha = exp2(-inf) = 0)
    float cr, cg, cb;  // colour
};

constexpr int SC_HALF = 32 + 1;            // pixel records of a half quadrant (8x4 pixels) + its all-zero record
constexpr int SC_NENT = 2 * SC_HALF;       // both halves of the quadrant being worked on
constexpr int SC_ENT_F4 = 3;               // float4 per staged entry

struct ScanLds {
    float4 ent[64 * SC_ENT_F4];       // the bucket's 64 entries (stride 12 floats)
    float acc[64 * 9];                // their nine sums, accumulated over the four quadrants
    uint32_t list[2 * 64];            // compacted entry indices of the two half quadrants being worked on
    float2 ta[SC_NENT], rg[SC_NENT], bx[SC_NENT], py[SC_NENT];   // pixel records in compacted order: {T, A} {g.r, g.g} {g.b, px} {py, -}
    uint2 hm[SC_NENT];                // ... and the pixel's decision mask
};

// One block's pixels (npx records in LDS from index rb) against its ne compacted entries (S.list from lb) in NG = ceil(ne / 16) groups:
// lane = (entry slot, pixel row).  A specialisation per group count: the chunk loop carries no per-group branch, and every group's entry and
// its nine sums stay in registers.  The block is a HALF quadrant (8x4 pixels): its entry set is smaller than the quadrant's (23 against 28 of
// 64 on the 2M / 1080p scene) and fills its groups of sixteen better — 16 % fewer steps than whole quadrants (tools/bwd_hitmask_model.py).
template <int NG>
__device__ __forceinline__ void block_pass(ScanLds& S, const int lane, const int rb, const int lb, const int npx_, const int ne, float c099)
{
    const int npx = rb + npx_;
    const int slot_i = lane & 15, row = lane >> 4;
    ScanEntry E[NG];
    uint32_t ent[NG], kbit[NG], klo[NG];
    v2f_s acc_S[NG], acc_cxy[NG], acc_rg[NG];
    float acc_cw[NG], acc_op[NG], acc_b[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) {
        acc_S[g] = acc_cxy[g] = acc_rg[g] = (v2f_s){0.f, 0.f};
        acc_cw[g] = acc_op[g] = acc_b[g] = 0.f;
        E[g] = {0.f, 0.f, 0.f, 0.f, 0.f, -__builtin_inff(), 0.f, 0.f, 0.f};   // an empty slot: alpha = exp2(-inf) = 0 whatever the masks say
        ent[g] = 0u; kbit[g] = 0u; klo[g] = 0u;
        const int k = 16 * g + slot_i;
        if (k < ne) {
            const uint32_t e = S.list[lb + k];
            const float4 e0 = S.ent[SC_ENT_F4 * e], e1 = S.ent[SC_ENT_F4 * e + 1], e2 = S.ent[SC_ENT_F4 * e + 2];
            E[g] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x};
            ent[g] = e; kbit[g] = e & 31u; klo[g] = e < 32u ? 0xffffffffu : 0u;
        }
    }
    const int nchunk = ((GS_SCAN_SKIP & 4) && c099 > 0.f) ? 0 : (npx_ + 3) >> 2;
    int off = rb + row;   // record index of this row's pixel; clamped to the block's all-zero record
    for (int c = 0; c < nchunk; c++, off += 4) {
        const int idx = off < npx ? off : npx;
        const float2 ta = S.ta[idx], rg = S.rg[idx], bx = S.bx[idx], pyv = S.py[idx];
        const uint2 hm = S.hm[idx];
        float Tin = ta.x, Ain = ta.y;
        const float pxf = bx.y, pyf = pyv.x;
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const float dx = E[g].d0x - pxf, dy = E[g].d0y - pyf;
            // log2(e) * power + log2(opacity) = dx (hA dx + nB dy) + hC dy^2 + lop: five instructions (the decisions come from the masks, so
            // the order of the rounding is free here; the pipeline's six repeat the fast forward's sequence bit for bit)
            float p2 = __builtin_fmaf(__builtin_fmaf(E[g].hA, dx, E[g].nB * dy), dx, E[g].lop);
            p2 = __builtin_fmaf(E[g].hC * dy, dy, p2);
            const float araw = __builtin_amdgcn_exp2f(p2);         // opacity * G
            // bit `entry` of the pixel's mask (the strict forward blended this pair) as an all-ones / all-zeros word
            uint32_t sel_, m_;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sel_) : "v"(klo[g]), "v"(hm.x), "v"(hm.y));
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m_) : "v"(sel_), "v"(kbit[g]));
            const float ah = __uint_as_float(__float_as_uint(araw) & m_);
            float alpha;                                           // min(0.99, .) (plain v_min_f32: VOP2, no canonicalising v_max in front of it)
            asm("v_min_f32 %0, %1, %2" : "=v"(alpha) : "v"(ah), "v"(c099));
            const float om = 1.0f - alpha;
            const float rinv = __builtin_amdgcn_rcpf(om);          // scales a gradient term, decides nothing
            const float Tincl = Tin * row_scan_mul(om);            // T behind this entry
            const float Ti = Tincl * rinv;                         // T in front of it
            const float Ta = Ti * alpha;
            float cgd = E[g].cr * rg.x;
            cgd = __builtin_fmaf(E[g].cg, rg.y, cgd);
            cgd = __builtin_fmaf(E[g].cb, bx.x, cgd);              // c . dL/dpixel
            const float Ap = Ain + row_scan_add(Ta * cgd);         // A behind this entry (the pipeline's A_ after its own contribution)
            const float dLda = __builtin_fmaf(rinv, Ap, Ti * cgd);
            const float w = ah * dLda;                             // opacity * G * dL/dalpha = G * dL/dG
            acc_rg[g] = __builtin_elementwise_fma((v2f_s){Ta, Ta}, (v2f_s){rg.x, rg.y}, acc_rg[g]);
            acc_b[g] = __builtin_fmaf(Ta, bx.x, acc_b[g]);
            const v2f_s d = {dx, dy};
            const v2f_s wd = (v2f_s){w, w} * d;
            acc_S[g] += wd;
            acc_cxy[g] = __builtin_elementwise_fma((v2f_s){wd.x, wd.x}, d, acc_cxy[g]);
            acc_cw[g] = __builtin_fmaf(wd.y, dy, acc_cw[g]);
            acc_op[g] += w;
            if (g + 1 < NG) { Tin = row_last(Tincl); Ain = row_last(Ap); }   // the pixel's state behind the group's sixteenth entry
        }
    }
    // the four rows hold sums over different pixels of the same 16 entries: add them (fixed order); row 0 adds the result to the entry's running sums
#pragma unroll
    for (int g = 0; g < NG; g++) {
        float v[9] = {acc_S[g].x, acc_S[g].y, acc_cxy[g].x, acc_cxy[g].y, acc_cw[g], acc_op[g], acc_rg[g].x, acc_rg[g].y, acc_b[g]};
#pragma unroll
        for (int k = 0; k < 9; k++) {
            v[k] += __shfl_xor(v[k], 16, 64);
            v[k] += __shfl_xor(v[k], 32, 64);
        }
        if (row == 0 && 16 * g + slot_i < ne) {
#pragma unroll
            for (int k = 0; k < 9; k++) S.acc[9 * ent[g] + k] += v[k];
        }
    }
}

#ifndef GS_SCAN_WAVES
#define GS_SCAN_WAVES 4   // waves per SIMD the register allocation aims at (124 VGPRs, no spills).  5 (96 VGPRs, 34 spilled, all but three reloads outside the
                          // chunk loops) was measured: 0.449 -> 0.592 ms (profiles/r04w_bwd_scan_five_waves_ab.log)
#endif
__global__ __launch_bounds__(64, GS_SCAN_WAVES) void render_bwd_scan_kernel(RenderBwdArgs a)
{
    __shared__ ScanLds S;
    const int lane = threadIdx.x;
    uint32_t bucket = blockIdx.x;
    if (a.xcd_lg >= 0) {   // runs of 2^xcd_lg consecutive buckets per XCD (launch_render_bwd)
        const uint32_t x = bucket & 7u, j = bucket >> 3;
        bucket = ((((j >> a.xcd_lg) << 3) + x) << a.xcd_lg) + (j & ((1u << a.xcd_lg) - 1u));
    }
    if (a.status[2] != 0u || bucket >= a.bucket_offsets[a.T - 1]) return;
    if (a.status[GS_FLAG_HITBITS] == 0u) return;   // no recorded decisions: the pipeline kernel (launched behind this one) does the work
    const uint32_t tile = a.bucket_to_tile[bucket];
    const uint2 range = a.ranges[tile];
    const uint32_t n = range.y - range.x;
    const uint32_t bbm = (tile == 0) ? 0u : a.bucket_offsets[tile - 1];
    const uint32_t bstart = (bucket - bbm) * GS_BUCKET;
    const uint32_t kit = bstart + (uint32_t)lane;
    const bool valid = kit < n;
    const uint32_t slot = valid ? a.inst_slot[range.x + kit] : 0u;
    if (bstart >= a.max_contrib[tile]) {   // bucket behind every pixel's last contributor (backward.cu:428)
        if (valid && !((GS_SCAN_SKIP & 8) && a.T > -1)) a.dead[slot] = 1;
        return;
    }
    const int tx0 = (int)(tile % (uint32_t)a.gx) * GS_TILE, ty0 = (int)(tile / (uint32_t)a.gx) * GS_TILE;
    const size_t plane = (size_t)a.H * a.W;

    // ---- lane = entry: stage the bucket's entries
    const float LOG2E = 1.4426950408889634f;
    {
    ScanEntry L = {0.f, 0.f, 0.f, 0.f, 0.f, -__builtin_inff(), 0.f, 0.f, 0.f};
    float rop = 0.f;
    if (valid && !((GS_SCAN_SKIP & 32) && a.T > -1)) {
        const uint32_t g = a.point_list[range.x + kit];
        const float4* rp = a.rec + GS_REC_F4 * (size_t)g;
        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
        L.d0x = r0.x - (float)tx0; L.d0y = r0.y - (float)ty0;
        L.hA = -0.5f * LOG2E * r0.z; L.nB = -LOG2E * r0.w; L.hC = -0.5f * LOG2E * r1.x;
        L.lop = __builtin_amdgcn_logf(r1.y);
        rop = r1.y > 0.f ? 1.0f / r1.y : 0.f;
        L.cr = r1.z; L.cg = r1.w; L.cb = r2.x;
    }
    S.ent[SC_ENT_F4 * lane] = make_float4(L.d0x, L.d0y, L.hA, L.hC);
    S.ent[SC_ENT_F4 * lane + 1] = make_float4(L.nB, L.lop, L.cr, L.cg);
    S.ent[SC_ENT_F4 * lane + 2] = make_float4(L.cb, rop, __uint_as_float(slot), 0.f);   // (.y, .z: what the lane needs again at the very end)
#pragma unroll
    for (int k = 0; k < 9; k++) S.acc[9 * lane + k] = 0.f;
    }

    float c099 = 0.99f;
    asm volatile("" : "+v"(c099));

    // One quadrant's per-pixel inputs.  All four loads of a lane are issued together (the checkpoint of a pixel that turns out inactive is
    // stale memory and is never used), and quadrant q + 1's are in flight while quadrant q is worked on: a bucket otherwise spends eight
    // dependent round trips to memory here with four waves per SIMD to hide them.
    struct PixIn { uint64_t hm; float4 pf, ck; float g0, g1, g2; };
    auto load_pix = [&](int q) {
        PixIn r;
        const int pidx = q * 64 + lane;
        r.hm = a.hit[(size_t)bucket * GS_TILE_PIX + pidx];
        r.pf = a.pix_final[(size_t)tile * GS_TILE_PIX + pidx];
        r.ck = a.ckpt[(size_t)bucket * GS_TILE_PIX + pidx];
        const int px = tx0 + tile_pix_x(pidx), py = ty0 + tile_pix_y(pidx);
        r.g0 = r.g1 = r.g2 = 0.f;
        if (px < a.W && py < a.H) {
            const size_t pid = (size_t)py * a.W + px;
            r.g0 = a.dL_dpix[pid]; r.g1 = a.dL_dpix[plane + pid]; r.g2 = a.dL_dpix[2 * plane + pid];
        }
        return r;
    };
#ifndef GS_SCAN_PREFETCH
#define GS_SCAN_PREFETCH 0   // 1: quadrant q + 1's inputs are fetched while quadrant q is worked on — measured slower (profiles/r04g_bwd_scan_ab.log:
#endif                       // the thirteen parked registers spill at four waves per SIMD)
#if GS_SCAN_PREFETCH
    PixIn nxt = load_pix(0);
#endif
    uint32_t any_lo = 0u, any_hi = 0u;   // entries SOME pixel of the tile blended (wave-uniform): the others' nine sums are exact zeros
    for (int q = 0; q < (((GS_SCAN_SKIP & 2) && a.T > -1) ? 0 : 4); q++) {
        // ---- lane = pixel of quadrant q
#if GS_SCAN_PREFETCH
        const PixIn cur = nxt;
        if (q < 3) nxt = load_pix(q + 1);   // (wave-uniform)
#else
        const PixIn cur = load_pix(q);
#endif
        const int pidx = q * 64 + lane;
        const uint64_t hm64 = cur.hm;
        const int lx = tile_pix_x(pidx), ly = tile_pix_y(pidx);
        const int px = tx0 + lx, py = ty0 + ly;
        // (a forward wave stops writing masks and checkpoints once all ITS pixels are finished: beyond a pixel's last contributor both are stale)
        const float4 pf = cur.pf;
        const bool active = px < a.W && py < a.H && __float_as_uint(pf.w) > bstart && hm64 != 0ull;
        const uint64_t bal = __ballot(active);
        if (bal == 0ull) continue;   // (wave-uniform)
        const uint32_t mlo = active ? (uint32_t)hm64 : 0u, mhi = active ? (uint32_t)(hm64 >> 32) : 0u;
        // lanes 0..31 are the quadrant's upper 8x4 pixels, lanes 32..63 its lower: one entry set per half
        uint32_t S_lo[2], S_hi[2];
        half_or(mlo, S_lo[0], S_lo[1]);
        half_or(mhi, S_hi[0], S_hi[1]);
        any_lo |= S_lo[0] | S_lo[1]; any_hi |= S_hi[0] | S_hi[1];
        const uint32_t balh[2] = {(uint32_t)bal, (uint32_t)(bal >> 32)};
        const int half = lane >> 5;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous quadrant's LDS traffic is done before its records are overwritten
        __builtin_amdgcn_wave_barrier();
        if (active) {
            const float4 ck = cur.ck;
            const float g0 = cur.g0, g1 = cur.g1, g2 = cur.g2;
            float A0 = (ck.y - pf.x) * g0;   // ar = checkpoint colour - final colour (backward.cu:522-523), dotted with dL/dpixel
            A0 = __builtin_fmaf(ck.z - pf.y, g1, A0);
            A0 = __builtin_fmaf(ck.w - pf.z, g2, A0);
            // position among the active pixels of this lane's half (mbcnt_lo counts the set bits below a lane of the lower half, mbcnt_hi of the upper)
            const uint32_t pos = half ? (uint32_t)SC_HALF + __builtin_amdgcn_mbcnt_hi(balh[1], 0u) : __builtin_amdgcn_mbcnt_lo(balh[0], 0u);
            S.ta[pos] = make_float2(ck.x, A0);
            S.rg[pos] = make_float2(g0, g1);
            S.bx[pos] = make_float2(g2, (float)lx);
            S.py[pos] = make_float2((float)ly, 0.f);
            S.hm[pos] = make_uint2(mlo, mhi);
        }
        if ((lane & 31) == 0) {   // the all-zero record of each half: what a row without a pixel (last chunk) works on — nothing blends, every product is an exact zero
            const int z = half * SC_HALF + __popc(half ? balh[1] : balh[0]);
            S.ta[z] = S.rg[z] = S.bx[z] = S.py[z] = make_float2(0.f, 0.f);
            S.hm[z] = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {   // the half's entries in list order: lane j with bit j of its set is entry number popcount(set below j)
            const bool mine = lane < 32 ? ((S_lo[h] >> lane) & 1u) : ((S_hi[h] >> (lane - 32)) & 1u);
            const uint32_t k = __builtin_amdgcn_mbcnt_hi(S_hi[h], __builtin_amdgcn_mbcnt_lo(S_lo[h], 0u));
            if (mine) S.list[64 * h + k] = (uint32_t)lane;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int h = 0; h < 2; h++) {
            const int npx = __popc(h ? balh[1] : balh[0]);
            if (npx == 0 || ((GS_SCAN_SKIP & 1) && a.T > -1)) continue;
            const int ne = h ? __popc(S_lo[1]) + __popc(S_hi[1]) : __popc(S_lo[0]) + __popc(S_hi[0]);
            switch ((ne + 15) >> 4) {
            case 1: block_pass<1>(S, lane, h * SC_HALF, 64 * h, npx, ne, c099); break;
            case 2: block_pass<2>(S, lane, h * SC_HALF, 64 * h, npx, ne, c099); break;
            case 3: block_pass<3>(S, lane, h * SC_HALF, 64 * h, npx, ne, c099); break;
            default: block_pass<4>(S, lane, h * SC_HALF, 64 * h, npx, ne, c099); break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // An entry no pixel of the tile blended (21 % of the instances of live buckets on the 2M / 1080p scene: the conservative tile test let them in,
    // or every pixel they reach was finished) has nine exact zeros: a flag byte — what the dead buckets write — stands for its 36-byte row, and
    // preprocess_bwd does not fetch it.  (GS_SCAN_ZERO_ROWS=1 writes the rows of zeros as before: A/B.)
#ifndef GS_SCAN_ZERO_ROWS
#define GS_SCAN_ZERO_ROWS 0
#endif
    const bool blended = GS_SCAN_ZERO_ROWS || (lane < 32 ? ((any_lo >> lane) & 1u) : ((any_hi >> (lane - 32)) & 1u));
    if (valid && !blended && !((GS_SCAN_SKIP & 8) && a.T > -1)) a.dead[slot] = 1;
    if (valid && blended && !((GS_SCAN_SKIP & 16) && a.T > -1)) {   // lane = entry again: the instance's 36-byte row (as render_bwd_kernel writes it)
        const float4 e0 = S.ent[SC_ENT_F4 * lane], e1 = S.ent[SC_ENT_F4 * lane + 1], e2 = S.ent[SC_ENT_F4 * lane + 2];
        const ScanEntry L = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x};
        const float rop = e2.y;
        const uint32_t slot = __float_as_uint(e2.z);
        const float* s = S.acc + 9 * lane;
        const float Sx = s[0], Sy = s[1], cxx = s[2], cxy = s[3], cyy = s[4], op = s[5], cr = s[6], cg = s[7], cb = s[8];
        const float kx = 0.5f * (float)a.W / LOG2E, ky = 0.5f * (float)a.H / LOG2E;
        const float gx = __builtin_fmaf(2.0f * L.hA, Sx, L.nB * Sy) * kx;
        const float gy = __builtin_fmaf(2.0f * L.hC, Sy, L.nB * Sx) * ky;
        float* o = a.partials + GS_PROW * (size_t)slot;
        *reinterpret_cast<gs_v4f_u*>(o) = (gs_v4f_u){gx, gy, -0.5f * cxx, -0.5f * cxy};
        *reinterpret_cast<gs_v4f_u*>(o + 4) = (gs_v4f_u){-0.5f * cyy, op * rop, cr, cg};
        o[8] = cb;
    }
}

int launch_render_bwd_scan(const RenderBwdArgs& b, unsigned grid, hipStream_t s)
{
    GS_LAUNCH(K_RENDER_BWD, render_bwd_scan_kernel, dim3(grid), dim3(64), 0, s, b);
    return GSLIC_OK;
}

}  // namespace gslic
