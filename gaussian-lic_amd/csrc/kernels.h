// kernels.h — argument blocks and host-side launchers of the rasterizer kernels (internal to libgslic_hip.so).
#pragma once
#include "gslic_common.h"

namespace gslic {

struct PreprocessArgs {
    int P, D, M, W, H, gx, gy;
    float focal_x, focal_y;
    float limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier;
    int prefiltered, no_color, raw;
    const float *means, *scales, *rots, *opac, *dc, *shs, *view, *proj, *campos;
    int32_t* radii;
    float4* rec;
    uint32_t* tiles_touched;
    uint32_t* depth_keys;  // optional [P] depth bits, 0xffffffff when culled (round 5: the forward no longer sorts the Gaussians by depth; NULL)
    uint2* ranges;         // [gx*gy] zeroed here (rasterizer_impl.cu:426) to save a memset launch
    uint32_t* flags;
};
int launch_preprocess(const PreprocessArgs& a, hipStream_t s);

struct KeybuildArgs {
    int P, gx, gy;
    const float4* rec;
    const uint32_t* order;     // NULL (round 5): thread i emits Gaussian i; else [P] Gaussian ids in emission order
    const uint32_t* offsets;   // [P] inclusive scan of tiles_touched in emission order
    uint32_t* tile_keys;       // [R] out: tile id per emission slot
    uint32_t* gauss;           // [R] out: Gaussian id per emission slot
    uint32_t* depth;           // [R] out: depth bits per emission slot (sort key of the per-tile depth sort)
    uint32_t* gauss_start;     // [P] out: first emission slot of each visible Gaussian
    uint32_t cap;              // slots available in tile_keys / gauss (capacity mode; the exact R otherwise)
    uint32_t* status;          // device status words: [2] |= 1 when an instance did not fit (nothing is written past cap)
};
int launch_keybuild(const KeybuildArgs& a, hipStream_t s);
// R_dev != NULL: the real instance count is read on the device and R is the capacity the launch is sized for
// dead != NULL: also clears the per-slot dead flags of the backward (BinningState::dead)
int launch_finalize_ranges(uint32_t R, const uint32_t* R_dev, const uint32_t* sorted_tiles, uint2* ranges, uint8_t* dead, hipStream_t s);
int launch_bucket_scan(int T, const uint2* ranges, uint32_t* bucket_offsets, uint32_t* max_contrib /* zeroed */, hipStream_t s);  // scan.hip

struct RenderFwdArgs {
    int W, H, gx, gy, no_color;
    const uint2* ranges;
    const uint32_t* point_list;
    const float4* rec;
    const uint32_t* bucket_offsets;
    uint32_t* bucket_to_tile;
    float4* ckpt;
    uint64_t* hit;             // SampleState::hit (written by the strict variant only, which then sets status[GS_FLAG_HITBITS])
    float4* pix_final;
    uint32_t* max_contrib;
    float* out_color;
    float* out_final_T;
    uint32_t capB;             // checkpoint buckets available in bucket_to_tile / ckpt
    uint32_t* status;          // device status words: [2] |= 2 when the buckets did not fit; a non-zero [2] on entry aborts the kernel
    int tail4_from;            // set by launch_render_fwd: tiles from this index on are blended by FOUR waves (one quadrant each); >= gx * gy: none
};
int launch_render_fwd(const RenderFwdArgs& a, hipStream_t s);

struct RenderBwdArgs {
    int W, H, gx, B;
    const uint2* ranges;
    const uint32_t* point_list;
    const uint32_t* inst_slot;
    const float4* rec;
    const uint32_t* bucket_offsets;
    const uint32_t* bucket_to_tile;
    const float4* ckpt;
    const uint64_t* hit;       // SampleState::hit (valid when status[GS_FLAG_HITBITS] != 0)
    const float4* pix_final;
    const uint32_t* max_contrib;
    const float* dL_dpix;
    float* partials;           // [9R] 36-byte rows, written for the instances of live buckets only
    uint8_t* dead;             // [R] set to 1 for the instances of dead buckets (zeroed by the forward)
    const uint32_t* status;    // device status words: a non-zero [2] (capacity overflow in the forward) aborts the kernel
    int T;                     // tiles: bucket_offsets[T - 1] is the real bucket count (B may be a capacity)
    int xcd_lg;                // log2 of the run of consecutive buckets one XCD takes (launch_render_bwd); < 0: bucket = blockIdx.x
    int skip_if_bits;          // pipeline kernel only: leave when the forward recorded decision masks (the row-scan kernel has done the work)
};
int launch_render_bwd(const RenderBwdArgs& a, hipStream_t s);
int launch_render_bwd_scan(const RenderBwdArgs& b, unsigned grid, hipStream_t s);   // render_bwd_scan.hip: strict arithmetic, needs SampleState::hit

struct AdamFusedArgs {
    float *p[6], *m[6], *v[6];  // xyz, features_dc, features_rest, opacity, scaling, rotation
    float lr[6];
    float b1, b2, eps;
    int on;
};

struct PreprocessBwdArgs {
    int P, D, M, W, H, raw;
    int row_begin, row_end;   // Gaussians [row_begin, row_end) of the P are processed (the whole range by default; row_begin % 64 == 0): the N > 1
                              // exchange runs the per-Gaussian backward in chunks and ships a chunk while the next one is computed
    float focal_x, focal_y;
    float limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier, lambda_erank;
    const float *means, *scales, *rots, *dc, *shs, *view, *proj, *campos;
    const int32_t* radii;
    const float4* rec;
    const uint32_t* tiles_touched;
    const uint32_t* gauss_start;
    const float* partials;
    const uint8_t* dead;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_ddc, *dL_dsh, *dL_dscale, *dL_drot;
    float* dL_drgb;          // optional [P,3]: dL/d(SH colour) AFTER the clamp mask (backward.cu:41-44) — what the SH backward is linear in.
                             // When set it is written INSTEAD of dL_ddc (which must be NULL): the N > 1 exchange ships these 12 bytes per
                             // Gaussian and every rank rebuilds dL_ddc / dL_dsh of all views from them (launch_sh_grad_from_rgb)
    AdamFusedArgs adam;
    const uint32_t* status;  // device status words: a non-zero [2] (capacity overflow in the forward) aborts the kernel (no gradients, no Adam)
    uint8_t* vis_out;     // optional [P]: 1 = radii > 0 — the visibility byte of the N > 1 exchange's payload, written here instead of by a host compare + copy
    float* campos_out;    // optional [3]: a copy of campos next to it (the payload's camera centre)
    float* cam_partials;  // optional [ceil(P/64)][32]: per-wave sums of the 27 camera-gradient terms (NULL: not computed)
    float* cam_out;       // [35] = dL_dviewmatrix[16] | dL_dprojmatrix[16] | dL_dcampos[3]
};
int launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s);

// dL_ddc [P,3] and dL_dsh [P,M,3] summed over n_views views from the views' masked colour gradients (rgb_all [n_views][P][3], or the
// views' dL_ddc when input_is_ddc) and camera centres (campos_all [n_views][3]): the SH backward (backward.cu:27-136) is the outer
// product of a direction-only coefficient vector with that colour gradient.  View order 0..n_views-1, plain multiply-then-add.
struct ShGradFromRgbArgs {
    int P, D, M, n_views, input_is_ddc;
    size_t rgb_stride, campos_stride;   // floats between consecutive views in rgb_all / campos_all (3 P and 3 for the dense layouts)
    const float *means3D, *campos_all, *rgb_all;
    float *dL_ddc, *dL_dsh;     // outputs; both may be NULL when the Adam update below consumes the rows
    const uint8_t* visible;     // [P] the exchanged visibility mask: rows the Adam update applies to.  vis_stride > 0: the views' masks sit vis_stride
    size_t vis_stride;          // bytes apart (an all-gathered payload) and are OR-ed here; vis_out (optional [P]) receives the OR
    uint8_t* vis_out;
    const float* g_small[4];    // optional: the summed (all-reduced) dL_dxyz [P,3], dL_dopacity [P,1], dL_dscaling [P,3], dL_drotation [P,4]: when set, the
                                // masked Adam of groups 0, 3, 4, 5 runs here too — one launch for the whole optimiser step of an N > 1 rank
    AdamFusedArgs adam;         // on: groups 1 (features_dc) and 2 (features_rest) are updated in place from the rebuilt rows
};
int launch_sh_grad_from_rgb(const ShGradFromRgbArgs& a, hipStream_t s);

}  // namespace gslic
