// preprocess.hip — per-Gaussian forward stage and the binning kernels.  COMPILE WITH -ffp-contract=off:
// everything here decides integers (radii, tiles_touched, sort keys) and follows the canonical operation
// order of oracle/gs_oracle.c bit for bit.
//
//   preprocess_kernel   replaces preprocessCUDA<3> fwd (forward.cu:232-319)
//   keybuild_kernel     replaces duplicateWithKeys (rasterizer_impl.cu:59-193)
//   finalize_lists_kernel replaces identifyTileRanges (rasterizer_impl.cu:195-218) + resolves payload -> Gaussian id
//   bucket_count_kernel replaces perTileBucketCount (rasterizer_impl.cu:221-231), bucket = 64 entries
//
// Wave64 load balancing: every lane tests the first SEQ_TILES tiles of its own rectangle; rectangles with
// more tiles are then swept one Gaussian at a time by the whole wave (64 tiles per step, ballot + mbcnt
// compaction), selected by a 64-bit ballot.  The per-tile test is a pure function of the Gaussian, so the
// schedule cannot change the result.
#include "gslic_common.h"
#include "kernels.h"

namespace gslic {

static constexpr int SEQ_TILES = 16;

__constant__ float c_SH_C0 = 0.28209479177387814f;
__constant__ float c_SH_C1 = 0.4886025119029199f;
__constant__ float c_SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                 0.5462742152960396f};
__constant__ float c_SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                 -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

__device__ __forceinline__ float fmin_c(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float fmax_c(float a, float b) { return (b > a) ? b : a; }

// SH -> RGB of one Gaussian (forward.cu:29-77), the reference's operation order; sh = the Gaussian's features_rest row.
__device__ __forceinline__ void sh_to_rgb(const PreprocessArgs& a, const int idx, const float px, const float py, const float pz,
                                          const float* __restrict__ sh, float (&rgb)[3], uint32_t& clamp_bits)
{
    float dx = px - a.campos[0], dy = py - a.campos[1], dz = pz - a.campos[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    const float* __restrict__ d0 = a.dc + 3 * (size_t)idx;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float res = c_SH_C0 * d0[ch];
#define S(k) sh[3 * (k) + ch]
        if (a.D > 0) {
            res = res - c_SH_C1 * y * S(0) + c_SH_C1 * z * S(1) - c_SH_C1 * x * S(2);
            if (a.D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + c_SH_C2[0] * xy * S(3) + c_SH_C2[1] * yz * S(4) + c_SH_C2[2] * (2.0f * zz - xx - yy) * S(5) +
                      c_SH_C2[3] * xz * S(6) + c_SH_C2[4] * (xx - yy) * S(7);
                if (a.D > 2) {
                    res = res + c_SH_C3[0] * y * (3.0f * xx - yy) * S(8) + c_SH_C3[1] * xy * z * S(9) +
                          c_SH_C3[2] * y * (4.0f * zz - xx - yy) * S(10) +
                          c_SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(11) +
                          c_SH_C3[4] * x * (4.0f * zz - xx - yy) * S(12) + c_SH_C3[5] * z * (xx - yy) * S(13) +
                          c_SH_C3[6] * x * (xx - 3.0f * yy) * S(14);
                }
            }
        }
#undef S
        res += 0.5f;
        if (res < 0.0f) clamp_bits |= (1u << ch);
        rgb[ch] = fmax_c(res, 0.0f);
    }
}

// LDS_SH (M == 15, colour mode, one wave per workgroup): the block's SH rows reach their threads through LDS (see "SH -> RGB" below).
// Latency-bound, and purely so: with LDS padding taking it from twenty waves per CU to fifteen it runs 22 % longer (profiles/r06s_small_kernels_occupancy_sensitivity.log).
// 97 VGPRs kept it at four waves per SIMD; capped at 96 it ran five (0.1330 -> 0.1304 ms); with the SH staging's float4 columns fetched three at a time
// instead of six (GS_PRE_NTB: 24 registers held across the staging -> 12) it fits 80 with two spilled and runs SIX: 0.1344 -> 0.1274 ms, same box
// (profiles/r06s_preprocess_staging_batches_ab.log; seven waves / 72 VGPRs / two columns at a time: 0.136).  -DGS_PRE_WPE=n -DGS_PRE_NTB=n for A/B runs.
#ifndef GS_PRE_EARLY_REC
#define GS_PRE_EARLY_REC 1   // 78 VGPRs, no spills, no scratch at six waves: 0.1249 -> 0.1227 ms (seven waves with it: 0.126; profiles/r06ad_preprocess_early_record_ab.log)
#endif
#ifndef GS_PRE_NTB
#define GS_PRE_NTB 3
#endif
#ifndef GS_PRE_WPE
#define GS_PRE_WPE 6
#endif
#define GS_PRE_WPE_ATTR __attribute__((amdgpu_waves_per_eu(GS_PRE_WPE, GS_PRE_WPE)))
template <bool LDS_SH, int BS>
__global__ __launch_bounds__(BS) GS_PRE_WPE_ATTR void preprocess_kernel(PreprocessArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds_sh[LDS_SH ? (BS / 2) * 45 : 4];
    __shared__ uint8_t lds_v[LDS_SH ? BS : 4];
    const int idx = blockIdx.x * BS + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (int t = idx; t < a.gx * a.gy; t += gridDim.x * BS) a.ranges[t] = make_uint2(0u, 0u);
    bool active = idx < a.P;
    const float* __restrict__ V = a.view;
    const float* __restrict__ Pm = a.proj;

    float px = 0, py = 0, pz = 0, depth = 0;
    float mx = 0, my = 0, cA = 0, cB = 0, cC = 0, op = 0, thr = 0;
    int radius = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0;

    if (active) {
        px = a.means[3 * idx]; py = a.means[3 * idx + 1]; pz = a.means[3 * idx + 2];
        // in_frustum (auxiliary.h:149-171): near cull only
        const float vz = V[2] * px + V[6] * py + V[10] * pz + V[14];
        depth = vz;
        if (vz <= 0.2f) {
            active = false;
            if (a.prefiltered) a.flags[0] = 1u;
        }
    }
    if (active) {
        const float hx = Pm[0] * px + Pm[4] * py + Pm[8] * pz + Pm[12];
        const float hy = Pm[1] * px + Pm[5] * py + Pm[9] * pz + Pm[13];
        const float hw = Pm[3] * px + Pm[7] * py + Pm[11] * pz + Pm[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        const float projx = hx * pw, projy = hy * pw;

        // computeCov3D (forward.cu:120-149)
        float qr = a.rots[4 * idx], qx = a.rots[4 * idx + 1], qy = a.rots[4 * idx + 2], qz = a.rots[4 * idx + 3];
        float sc0 = a.scales[3 * idx], sc1 = a.scales[3 * idx + 1], sc2 = a.scales[3 * idx + 2];
        if (a.raw) {  // activations of gaussian.cpp:147-175: exp(scaling_), normalize(rotation_) (eps 1e-12)
            sc0 = expf(sc0); sc1 = expf(sc1); sc2 = expf(sc2);
            const float nrm = fmax_c(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f);
            qr = qr / nrm; qx = qx / nrm; qy = qy / nrm; qz = qz / nrm;
        }
        float Rm[3][3];
        Rm[0][0] = 1.f - 2.f * (qy * qy + qz * qz); Rm[0][1] = 2.f * (qx * qy - qr * qz); Rm[0][2] = 2.f * (qx * qz + qr * qy);
        Rm[1][0] = 2.f * (qx * qy + qr * qz); Rm[1][1] = 1.f - 2.f * (qx * qx + qz * qz); Rm[1][2] = 2.f * (qy * qz - qr * qx);
        Rm[2][0] = 2.f * (qx * qz - qr * qy); Rm[2][1] = 2.f * (qy * qz + qr * qx); Rm[2][2] = 1.f - 2.f * (qx * qx + qy * qy);
        const float s[3] = {a.scale_modifier * sc0, a.scale_modifier * sc1, a.scale_modifier * sc2};
        float Mk[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) Mk[k][c] = s[k] * Rm[c][k];
#define SIG(i, j) (Mk[0][i] * Mk[0][j] + Mk[1][i] * Mk[1][j] + Mk[2][i] * Mk[2][j])
        const float c6[6] = {SIG(0, 0), SIG(0, 1), SIG(0, 2), SIG(1, 1), SIG(1, 2), SIG(2, 2)};
#undef SIG
        // computeCov2D (forward.cu:79-118)
        float t0 = V[0] * px + V[4] * py + V[8] * pz + V[12];
        float t1 = V[1] * px + V[5] * py + V[9] * pz + V[13];
        const float t2 = V[2] * px + V[6] * py + V[10] * pz + V[14];
        const float txtz = t0 / t2, tytz = t1 / t2;
        t0 = fmin_c(a.limx_pos, fmax_c(a.limx_neg, txtz)) * t2;
        t1 = fmin_c(a.limy_pos, fmax_c(a.limy_neg, tytz)) * t2;
        const float J00 = a.focal_x / t2;
        const float J02 = -(a.focal_x * t0) / (t2 * t2);
        const float J11 = a.focal_y / t2;
        const float J12 = -(a.focal_y * t1) / (t2 * t2);
        float T0[3], T1[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float w0 = V[4 * i + 0], w1 = V[4 * i + 1], w2 = V[4 * i + 2];
            T0[i] = w0 * J00 + w2 * J02;
            T1[i] = w1 * J11 + w2 * J12;
        }
        const float Vr[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float A0[3], A1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            A0[k] = T0[0] * Vr[k][0] + T0[1] * Vr[k][1] + T0[2] * Vr[k][2];
            A1[k] = T1[0] * Vr[k][0] + T1[1] * Vr[k][1] + T1[2] * Vr[k][2];
        }
        const float cov0 = (A0[0] * T0[0] + A0[1] * T0[1] + A0[2] * T0[2]) + 0.3f;
        const float cov1 = A1[0] * T0[0] + A1[1] * T0[1] + A1[2] * T0[2];
        const float cov2 = (A1[0] * T1[0] + A1[1] * T1[1] + A1[2] * T1[2]) + 0.3f;

        const float det = cov0 * cov2 - cov1 * cov1;
        op = a.opac[idx];
        if (a.raw) op = 1.0f / (1.0f + expf(-op));  // sigmoid(opacity_)
        if (det == 0.0f || op < (1.0f / 255.0f)) {
            active = false;
        } else {
            const float det_inv = 1.f / det;
            cA = cov2 * det_inv; cB = -cov1 * det_inv; cC = cov0 * det_inv;
            const float mid = 0.5f * (cov0 + cov2);
            const float lambda1 = mid + sqrtf(fmax_c(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(lambda1));
            mx = (float)((((double)projx + 1.0) * (double)a.W - 1.0) * 0.5);  // ndc2Pix, double (auxiliary.h:41-44)
            my = (float)((((double)projy + 1.0) * (double)a.H - 1.0) * 0.5);
            radius = (my_radius >= 2147483520.f) ? 2147483520 : (int)my_radius;
            get_rect(mx, my, radius, a.gx, a.gy, x0, y0, x1, y1);
            thr = cull_threshold(op);
        }
    }

    // ---- exact tile count (forward.cu:151-230) ----
    const int rw = x1 - x0;
    const int ntiles = active ? rw * (y1 - y0) : 0;
    uint32_t cnt = 0;
    uint32_t seqmask = 0;   // bit t = tile t of the rectangle (row-major) passed the test, t < SEQ_TILES: keybuild_kernel emits from it instead of testing again
    float rcpx = 0.f, rcpy = 0.f;
    if (active) tile_power_prep(cA, cC, rcpx, rcpy);
    {
        int tx = x0, ty = y0;
        const int nseq = ntiles < SEQ_TILES ? ntiles : SEQ_TILES;
        for (int t = 0; t < nseq; t++) {
            const uint32_t hit = (tile_min_power_p(cA, cB, cC, mx, my, rcpx, rcpy, tx, ty) <= thr) ? 1u : 0u;
            cnt += hit;
            seqmask |= hit << t;
            if (++tx == x1) { tx = x0; ++ty; }
        }
    }
    uint64_t todo = __ballot(ntiles > SEQ_TILES);
    while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const float sA = readlane_f(cA, src), sB = readlane_f(cB, src), sC = readlane_f(cC, src);
        const float smx = readlane_f(mx, src), smy = readlane_f(my, src), sthr = readlane_f(thr, src);
        const float srx = readlane_f(rcpx, src), sry = readlane_f(rcpy, src);
        const int sx0 = __builtin_amdgcn_readlane(x0, src), sy0 = __builtin_amdgcn_readlane(y0, src);
        const int srw = __builtin_amdgcn_readlane(rw, src), sn = __builtin_amdgcn_readlane(ntiles, src);
        uint32_t c = 0;
        for (int base = SEQ_TILES; base < sn; base += 64) {
            const int t = base + lane;
            const bool in = t < sn;
            const int ty = sy0 + t / srw, tx = sx0 + t % srw;
            const bool ok = in && (tile_min_power_p(sA, sB, sC, smx, smy, srx, sry, tx, ty) <= sthr);
            c += (uint32_t)__popcll(__ballot(ok));
        }
        if (lane == src) cnt += c;
    }
    const bool visible = active && cnt > 0;

    if (idx < a.P) {
        a.radii[idx] = visible ? radius : 0;
        a.tiles_touched[idx] = visible ? cnt : 0u;
        if (a.depth_keys) a.depth_keys[idx] = visible ? __float_as_uint(depth) : 0xffffffffu;  // low half of the reference's sort key (forward.cu:254)
    }
    // ---- SH -> RGB (forward.cu:29-77) ----
    float rgb[3] = {0.f, 0.f, 0.f};
    uint32_t clamp_bits = 0;
#if GS_PRE_EARLY_REC
    // the geometry half of the record leaves before the SH phase instead of living in eight registers across it (the colour half follows)
    if (visible) {
        float4* rec_e = a.rec + GS_REC_F4 * (size_t)idx;
        rec_e[0] = make_float4(mx, my, cA, cB);
        *reinterpret_cast<float2*>(rec_e + 1) = make_float2(cC, op);
        *reinterpret_cast<float2*>(reinterpret_cast<float*>(rec_e + 2) + 1) = make_float2(depth, 0.f);   // (.z rewritten below with the clamp bits)
        reinterpret_cast<float*>(rec_e + 2)[3] = __int_as_float(radius);
    }
#endif
    if constexpr (LDS_SH) {
        // The block's 64 x 45 SH floats are contiguous in memory: they pass through LDS in TWO rounds of 32 rows (5.6 KB instead of 11.3:
        // the kernel is latency-bound and loses 18 % when LDS padding takes it from 13 to 10 waves per CU, profiles/r03t_occupancy_sweep.log),
        // as coalesced float4 columns — only those that touch a visible Gaussian's row — and are read row-wise by the owning thread (row
        // stride 45 floats: odd, bank-conflict free) instead of 45 strided 4-byte loads per visible Gaussian.
        typedef float v4f __attribute__((ext_vector_type(4)));
        constexpr int HR = BS / 2, NV = HR * 45 / 4, NT = (NV + BS - 1) / BS;
        const int row0 = blockIdx.x * BS;
        const int rows = (a.P - row0) < BS ? (a.P - row0) : BS;
        lds_v[threadIdx.x] = visible ? 1 : 0;
        __syncthreads();
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const float* src = a.shs + (size_t)(row0 + h * HR) * 45;
            const int hrows = (rows - h * HR) < HR ? (rows - h * HR) : HR;
            if (hrows == HR) {
                const v4f* s4 = reinterpret_cast<const v4f*>(src);
                v4f* d4 = reinterpret_cast<v4f*>(lds_sh);
                // (GS_PRE_NTB float4 columns of a thread in flight at a time: all NT = 6 at once held 24 registers across the staging)
                constexpr int NTB = GS_PRE_NTB < NT ? GS_PRE_NTB : NT;
#pragma unroll
                for (int k0 = 0; k0 < NT; k0 += NTB) {
                    v4f pre[NTB];
                    bool ld[NTB];
#pragma unroll
                    for (int kk = 0; kk < NTB; kk++) {
                        const int i = threadIdx.x + (k0 + kk) * BS;
                        ld[kk] = false;
                        if (k0 + kk < NT && i < NV) {
                            const int e = 4 * i;
                            ld[kk] = (lds_v[h * HR + e / 45] | lds_v[h * HR + (e + 3) / 45]) != 0;
                            if (ld[kk]) pre[kk] = __builtin_nontemporal_load(s4 + i);   // 360 MB read once per forward: past the L2's retention
                        }
                    }
#pragma unroll
                    for (int kk = 0; kk < NTB; kk++) {
                        const int i = threadIdx.x + (k0 + kk) * BS;
                        if (ld[kk]) d4[i] = pre[kk];
                    }
                }
            } else {
                for (int i = threadIdx.x; i < hrows * 45; i += BS) lds_sh[i] = src[i];
            }
            __syncthreads();
            if (visible && (int)(threadIdx.x / HR) == h) sh_to_rgb(a, idx, px, py, pz, lds_sh + (threadIdx.x % HR) * 45, rgb, clamp_bits);
            __syncthreads();
        }
        if (!visible) return;
    } else {
        if (!visible) return;
        if (!a.no_color) sh_to_rgb(a, idx, px, py, pz, a.shs ? a.shs + (size_t)3 * a.M * idx : nullptr, rgb, clamp_bits);
    }
    float4* rec = a.rec + GS_REC_F4 * (size_t)idx;
#if GS_PRE_EARLY_REC
    *reinterpret_cast<float2*>(reinterpret_cast<float*>(rec + 1) + 2) = make_float2(rgb[0], rgb[1]);
    reinterpret_cast<float*>(rec + 2)[0] = rgb[2];
    reinterpret_cast<float*>(rec + 2)[2] = __uint_as_float(clamp_bits | (seqmask << 16));
#else
    rec[0] = make_float4(mx, my, cA, cB);
    rec[1] = make_float4(cC, op, rgb[0], rgb[1]);
    rec[2] = make_float4(rgb[2], depth, __uint_as_float(clamp_bits | (seqmask << 16)), __int_as_float(radius));   // (.z: clamp bits 0-2, tile mask 16-31)
#endif
}

// Thread i emits the instances of Gaussian g = i (index order, as duplicateWithKeys does: rasterizer_impl.cu:59-193) into the emission
// slots u in [offsets[i-1], offsets[i]), tiles in row-major order: tile_keys[u] = tile, gauss[u] = g, depth[u] = the Gaussian's depth bits.
// The instances are then grouped by tile — block-aggregated atomics (tile_bin.hip) or a STABLE sort on the tile id (radix_sort.hip) — and the
// per-tile sort (tile_depth_sort_wave_kernel) puts every tile's segment into (depth, original index) order: the order the reference's 64-bit
// (tile << 32 | depth) sort of the same index-ordered emission produces (rasterizer_impl.cu:86-128, 419-424).  The slot u is also where the backward writes the
// instance's partial gradients: contiguous per Gaussian, starting at gauss_start[g].
// (Rounds 2-4 sorted the GAUSSIANS by depth first — four passes over P, a gather scan, a depth-ordered gather of the records here — and
// emitted in that order; the per-tile sort replaces all of it: every read of this kernel is coalesced now.)
static constexpr int KB_CAP = 1024;  // instances a wave stages in LDS (64 Gaussians x 4.3 tiles on average; larger waves store directly)
__global__ __launch_bounds__(256) void keybuild_kernel(KeybuildArgs a)
{
    // A wave's 64 Gaussians own one contiguous range of emission slots.  Their (tile, id) pairs are staged in LDS and leave as
    // contiguous 256-byte stores: per-lane runs of ~4 four-byte stores at a stride of ~17 B were measured at 4.6x write amplification.
    __shared__ uint32_t s_tile[4][KB_CAP];
    __shared__ uint8_t s_gid[4][KB_CAP];   // owner lane of every staged instance
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t off = 0, end = 0;
    if (i < a.P) {
        off = (i == 0) ? 0u : a.offsets[i - 1];
        end = a.offsets[i];
    }
    const bool active = end > off;  // tiles_touched > 0  <=>  visible
    const int i0 = blockIdx.x * 256 + wave * 64;
    const int last = (a.P - 1 - i0) < 63 ? (a.P - 1 - i0) : 63;  // last lane with a Gaussian (wave-uniform; < 0: empty wave)
    const uint32_t wbase = readlane_u(off, 0);
    const uint32_t wcount = last >= 0 ? readlane_u(end, last) - wbase : 0u;
    const bool staged = wcount <= (uint32_t)KB_CAP;
    uint32_t* const st = s_tile[wave];
    uint8_t* const sg = s_gid[wave];
    const int idx = active ? (a.order ? (int)a.order[i] : i) : 0;
    float mx = 0, my = 0, cA = 0, cB = 0, cC = 0, thr = 0;
    uint32_t dbits = 0, seqmask = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (active) {
        const float4 r0 = a.rec[GS_REC_F4 * (size_t)idx], r1 = a.rec[GS_REC_F4 * (size_t)idx + 1], r2 = a.rec[GS_REC_F4 * (size_t)idx + 2];
        mx = r0.x; my = r0.y; cA = r0.z; cB = r0.w; cC = r1.x;
        dbits = __float_as_uint(r2.y);   // depth (forward.cu:254: the low half of the reference's key)
        seqmask = __float_as_uint(r2.z) >> 16;   // the preprocess kernel's test results of the rectangle's first SEQ_TILES tiles
        a.gauss_start[idx] = off;
        get_rect(mx, my, __float_as_int(r2.w) /* radius, parked in the record's spare word */, a.gx, a.gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) > SEQ_TILES) thr = cull_threshold(r1.y);   // (only rectangles of more than SEQ_TILES tiles are tested here)
    }
    const int rw = x1 - x0;
    const int ntiles = active ? rw * (y1 - y0) : 0;
    float rcpx = 0.f, rcpy = 0.f;
    if (active && ntiles > SEQ_TILES) tile_power_prep(cA, cC, rcpx, rcpy);
    {
        int tx = x0, ty = y0;
        const int nseq = ntiles < SEQ_TILES ? ntiles : SEQ_TILES;
        for (int t = 0; t < nseq; t++) {
            if ((seqmask >> t) & 1u) {   // (the exact test of forward.cu:151-230 ran in preprocess_kernel: same rectangle, same order)
                const uint32_t key = (uint32_t)(ty * a.gx + tx);
                if (staged) { st[off - wbase] = key; sg[off - wbase] = (uint8_t)lane; }
                else if (off < a.cap) { a.tile_keys[off] = key; a.gauss[off] = (uint32_t)idx; a.depth[off] = dbits; }
                off++;
            }
            if (++tx == x1) { tx = x0; ++ty; }
        }
    }
    uint64_t todo = __ballot(ntiles > SEQ_TILES);
    while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const float sA = readlane_f(cA, src), sB = readlane_f(cB, src), sC = readlane_f(cC, src);
        const float smx = readlane_f(mx, src), smy = readlane_f(my, src), sthr = readlane_f(thr, src);
        const float srx = readlane_f(rcpx, src), sry = readlane_f(rcpy, src);
        const int sx0 = __builtin_amdgcn_readlane(x0, src), sy0 = __builtin_amdgcn_readlane(y0, src);
        const int srw = __builtin_amdgcn_readlane(rw, src), sn = __builtin_amdgcn_readlane(ntiles, src);
        const uint32_t sidx = (uint32_t)__builtin_amdgcn_readlane(idx, src), sdep = readlane_u(dbits, src);
        uint32_t soff = readlane_u(off, src);
        for (int base = SEQ_TILES; base < sn; base += 64) {
            const int t = base + lane;
            const bool in = t < sn;
            const int ty = sy0 + t / srw, tx = sx0 + t % srw;
            const bool ok = in && (tile_min_power_p(sA, sB, sC, smx, smy, srx, sry, tx, ty) <= sthr);
            const uint64_t m = __ballot(ok);
            if (ok) {
                const uint32_t o = soff + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t key = (uint32_t)(ty * a.gx + tx);
                if (staged) { st[o - wbase] = key; sg[o - wbase] = (uint8_t)src; }
                else if (o < a.cap) { a.tile_keys[o] = key; a.gauss[o] = sidx; a.depth[o] = sdep; }
            }
            soff += (uint32_t)__popcll(m);
        }
    }
    if (staged) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the staged pairs are (tile, OWNER LANE): the owner's Gaussian id and depth bits come by a lane permute (no third staging array: LDS
        // is this kernel's occupancy)
        for (uint32_t k0 = 0; k0 < wcount; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            const bool in = k < wcount;
            const int owner = in ? (int)sg[k] : 0;
            const uint32_t gid = (uint32_t)__shfl(idx, owner, 64), dep = (uint32_t)__shfl((int)dbits, owner, 64);
            if (in && wbase + k < a.cap) {
                a.tile_keys[wbase + k] = st[k];
                a.gauss[wbase + k] = gid;
                a.depth[wbase + k] = dep;
            }
        }
    }
    // capacity mode: the instances of this wave do not all fit — report it; every later stage of the step then stands down
    if (lane == 0 && wcount != 0 && wbase + wcount > a.cap) atomicOr(a.status + 2, 1u);
}

// tile ranges of the sorted instance list (identifyTileRanges, rasterizer_impl.cu:131-156)
__global__ __launch_bounds__(256) void finalize_ranges_kernel(uint32_t R_cap, const uint32_t* __restrict__ R_dev, const uint32_t* __restrict__ tiles,
                                                              uint2* __restrict__ ranges, uint8_t* __restrict__ dead)
{
    const uint32_t R = R_dev ? (*R_dev < R_cap ? *R_dev : R_cap) : R_cap;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= R) return;
    if (dead) dead[k] = 0;   // the backward's dead-instance flags (one byte per emission slot) start clear: a coalesced store riding on this pass
    const uint32_t tile = tiles[k];
    if (k == 0) {
        ranges[tile].x = 0;
    } else {
        const uint32_t prev = tiles[k - 1];
        if (tile != prev) {
            ranges[prev].y = k;
            ranges[tile].x = k;
        }
    }
    if (k == R - 1) ranges[tile].y = R;
}

int launch_preprocess(const PreprocessArgs& a, hipStream_t s)
{
    if (a.M == 15 && a.D > 0 && a.shs && !a.no_color) {
        // one wave per workgroup (as preprocess_bwd): the SH staging barrier is wave-level; 0.197 ms against 0.216 at 256 threads.
        // Staging the small inputs / the record through LDS as well was measured and is SLOWER here (0.30 ms): the kernel is
        // latency-bound and the extra barrier ahead of the geometry serialises load -> compute -> store inside the wave.
        GS_LAUNCH(K_PREPROCESS, (preprocess_kernel<true, 64>), dim3(div_up(a.P, 64)), dim3(64), 0, s, a);
    } else {
        GS_LAUNCH(K_PREPROCESS, (preprocess_kernel<false, 256>), dim3(div_up(a.P, 256)), dim3(256), 0, s, a);
    }
    return GSLIC_OK;
}
int launch_keybuild(const KeybuildArgs& a, hipStream_t s)
{
    GS_LAUNCH(K_KEYBUILD, keybuild_kernel, dim3(div_up(a.P, 256)), dim3(256), 0, s, a);
    return GSLIC_OK;
}
int launch_finalize_ranges(uint32_t R, const uint32_t* R_dev, const uint32_t* sorted_tiles, uint2* ranges, uint8_t* dead, hipStream_t s)
{
    if (R == 0) return GSLIC_OK;
    GS_LAUNCH(K_FINALIZE_LISTS, finalize_ranges_kernel, dim3((R + 255u) / 256u), dim3(256), 0, s, R, R_dev, sorted_tiles, ranges, dead);
    return GSLIC_OK;
}
}  // namespace gslic
