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
//   64-deep systolic pipeline; the per-pixel state {T, ar[3], n_contrib, dL/dpixel[3]} moves lane -> lane+1 with
//   one v_mov_b32 DPP wave_shr:1 per value (no ds_bpermute, no LDS), lane 0 is fed from a 64-pixel register
//   chunk via v_readlane.  Each lane accumulates its Gaussian's nine 2D gradients in registers and writes them
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
    const float pxf = (float)px;
    float pyf[4], T[4], Cr[4], Cg[4], Cb[4];
    uint32_t last[4];
    bool done[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int py = pyb + 4 * q;
        pyf[q] = (float)py;
        T[q] = 1.0f; Cr[q] = Cg[q] = Cb[q] = 0.0f;
        last[q] = 0;
        done[q] = !(px < a.W && py < a.H);
    }

    for (int base = 0; base < n; base += GS_BUCKET) {
        if (__all(done[0] && done[1] && done[2] && done[3])) break;
        if (color) {
            float4* ck = a.ckpt + ((size_t)(bbm + (uint32_t)(base / GS_BUCKET)) * GS_TILE_PIX) + lane;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (!done[q]) ck[q * 64] = make_float4(T[q], Cr[q], Cg[q], Cb[q]);
        }
        const int m = (n - base) < GS_BUCKET ? (n - base) : GS_BUCKET;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
        if (lane < m) {
            const uint32_t g = a.point_list[range.x + (uint32_t)(base + lane)];
            const float4* rp = a.rec + 3 * (size_t)g;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
        }
        for (int j = 0; j < m; j++) {
            const float gmx = readlane_f(r0.x, j), gmy = readlane_f(r0.y, j);
            const float cA = readlane_f(r0.z, j), cB = readlane_f(r0.w, j), cC = readlane_f(r1.x, j);
            const float op = readlane_f(r1.y, j);
            const float colr = readlane_f(r1.z, j), colg = readlane_f(r1.w, j), colb = readlane_f(r2.x, j);
            const uint32_t contributor = (uint32_t)(base + j + 1);
            const float dx = gmx - pxf;
            const float adx2 = cA * dx * dx;
            const float bdx = cB * dx;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float dy = gmy - pyf[q];
                const float power = -0.5f * (adx2 + cC * dy * dy) - bdx * dy;
                const float alpha = fminf(0.99f, op * __expf(power));
                bool ok = !done[q] && !(power > 0.0f) && !(alpha < (1.0f / 255.0f));
                const float test_T = T[q] * (1.0f - alpha);
                const bool stop = ok && (test_T < 0.0001f);
                done[q] = done[q] || stop;
                ok = ok && !stop;
                if (ok) {
                    const float w = alpha * T[q];
                    Cr[q] += colr * w; Cg[q] += colg * w; Cb[q] += colb * w;
                    T[q] = test_T;
                    last[q] = contributor;
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
            a.out_final_T[pid] = T[q];
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

__global__ __launch_bounds__(256) void render_bwd_kernel(RenderBwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t bucket = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (bucket >= (uint32_t)a.B) return;
    const uint32_t tile = a.bucket_to_tile[bucket];
    const uint2 range = a.ranges[tile];
    const uint32_t n = range.y - range.x;
    const uint32_t bbm = (tile == 0) ? 0u : a.bucket_offsets[tile - 1];
    const uint32_t bit = bucket - bbm;
    const uint32_t kit = bit * GS_BUCKET + (uint32_t)lane;  // splat index in tile
    const bool valid = kit < n;
    const uint32_t slot = valid ? a.inst_slot[range.x + kit] : 0u;

    // bucket entirely behind every pixel's last contributor (backward.cu:428): gradients are exactly zero
    if (bit * GS_BUCKET >= a.max_contrib[tile]) {
        if (valid) {
            float4* o = a.partials + 3 * (size_t)slot;
            o[0] = o[1] = o[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

    float gmx = 0, gmy = 0, cA = 0, cB = 0, cC = 0, op = 0, colr = 0, colg = 0, colb = 0;
    if (valid) {
        const uint32_t g = a.point_list[range.x + kit];
        const float4* rp = a.rec + 3 * (size_t)g;
        const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
        gmx = r0.x; gmy = r0.y; cA = r0.z; cB = r0.w; cC = r1.x; op = r1.y; colr = r1.z; colg = r1.w; colb = r2.x;
    }
    float acc_mx = 0, acc_my = 0, acc_cx = 0, acc_cy = 0, acc_cw = 0, acc_op = 0, acc_r = 0, acc_g = 0, acc_b = 0;

    const int tx0 = (int)(tile % (uint32_t)a.gx) * GS_TILE, ty0 = (int)(tile / (uint32_t)a.gx) * GS_TILE;
    const float ddelx_dx = 0.5f * (float)a.W, ddely_dy = 0.5f * (float)a.H;
    const size_t plane = (size_t)a.H * a.W;

    // pipeline state (pixel currently at this lane) and the 64-pixel feed chunk
    float T = 0, ar0 = 0, ar1 = 0, ar2 = 0, g0 = 0, g1 = 0, g2 = 0;
    uint32_t nc = 0;
    float fT = 0, fa0 = 0, fa1 = 0, fa2 = 0, fg0 = 0, fg1 = 0, fg2 = 0;
    uint32_t fnc = 0;

    for (int i = 0; i < GS_TILE_PIX + 63; i++) {
        if ((i & 63) == 0 && i < GS_TILE_PIX) {
            const int pidx = i + lane;
            const float4 ck = a.ckpt[(size_t)bucket * GS_TILE_PIX + pidx];
            const float4 pf = a.pix_final[(size_t)tile * GS_TILE_PIX + pidx];
            const int px = tx0 + (pidx & 15), py = ty0 + (pidx >> 4);
            const bool inside = px < a.W && py < a.H;
            fnc = inside ? __float_as_uint(pf.w) : 0u;
            fT = ck.x;
            fa0 = ck.y - pf.x; fa1 = ck.z - pf.y; fa2 = ck.w - pf.z;
            fg0 = fg1 = fg2 = 0.f;
            if (inside) {
                const size_t pid = (size_t)py * a.W + px;
                fg0 = a.dL_dpix[pid]; fg1 = a.dL_dpix[plane + pid]; fg2 = a.dL_dpix[2 * plane + pid];
            }
        }
        // hand the pixel state to the next lane
        T = wave_shr1_f(T); ar0 = wave_shr1_f(ar0); ar1 = wave_shr1_f(ar1); ar2 = wave_shr1_f(ar2);
        g0 = wave_shr1_f(g0); g1 = wave_shr1_f(g1); g2 = wave_shr1_f(g2);
        nc = wave_shr1_u(nc);
        // lane 0 takes pixel i from the feed chunk (or an empty slot once the tile is exhausted)
        {
            const int sl = i & 63;
            const float iT = readlane_f(fT, sl), ia0 = readlane_f(fa0, sl), ia1 = readlane_f(fa1, sl), ia2 = readlane_f(fa2, sl);
            const float ig0 = readlane_f(fg0, sl), ig1 = readlane_f(fg1, sl), ig2 = readlane_f(fg2, sl);
            const uint32_t inc = (i < GS_TILE_PIX) ? readlane_u(fnc, sl) : 0u;
            if (lane == 0) { T = iT; ar0 = ia0; ar1 = ia1; ar2 = ia2; g0 = ig0; g1 = ig1; g2 = ig2; nc = inc; }
        }
        if (kit < nc) {  // this Gaussian was (possibly) blended into this pixel (backward.cu:538)
            const int idx = i - lane;
            const float dx = gmx - (float)(tx0 + (idx & 15));
            const float dy = gmy - (float)(ty0 + (idx >> 4));
            const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, op * G);
            if (!(power > 0.0f) && !(alpha < (1.0f / 255.0f))) {
                const float dchannel_dcolor = alpha * T;
                const float alpha_inv = 1.0f / (1.0f - alpha);
                const float Ta = T * alpha;
                ar0 += Ta * colr; ar1 += Ta * colg; ar2 += Ta * colb;
                acc_r += dchannel_dcolor * g0; acc_g += dchannel_dcolor * g1; acc_b += dchannel_dcolor * g2;
                float dL_dalpha = ((colr * T) + alpha_inv * ar0) * g0;
                dL_dalpha += ((colg * T) + alpha_inv * ar1) * g1;
                dL_dalpha += ((colb * T) + alpha_inv * ar2) * g2;
                T *= (1.0f - alpha);
                const float dL_dG = op * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * cA - gdy * cB;
                const float dG_ddely = -gdy * cC - gdx * cB;
                acc_mx += dL_dG * dG_ddelx * ddelx_dx;
                acc_my += dL_dG * dG_ddely * ddely_dy;
                acc_cx += -0.5f * gdx * dx * dL_dG;
                acc_cy += -0.5f * gdx * dy * dL_dG;
                acc_cw += -0.5f * gdy * dy * dL_dG;
                acc_op += G * dL_dalpha;
            }
        }
    }
    if (valid) {
        float4* o = a.partials + 3 * (size_t)slot;
        o[0] = make_float4(acc_mx, acc_my, acc_cx, acc_cy);
        o[1] = make_float4(acc_cw, acc_op, acc_r, acc_g);
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
