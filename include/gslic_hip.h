/*
 * gslic_hip.h — C-ABI of libgslic_hip.so: the MI355X (gfx950) differentiable 3D-Gaussian-splatting
 * hot path of Gaussian-LIC (render forward / backward, sparse Adam, fused SSIM, simple-knn).
 *
 * Drop-in boundary.  The reference has no FFI; its seam is the L2 layer of six free functions on
 * torch::Tensor that only unwrap pointers (src/rasterizer/rasterize_points.h:25-96,
 * src/fused-ssim/ssim.h:7-26, src/simple-knn/spatial.h:14) and the raw-pointer L1 layer underneath
 * (src/rasterizer/cuda_rasterizer/rasterizer.h:29-98, adam.h:12-23, src/simple-knn/simple_knn.h:18).
 * Every entry point below names the L1/L2 function it replaces.  The LibTorch shim that re-exports the
 * reference's exact L2 signatures on top of this header lives in gaussian-lic_amd/shim/.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.
 *  - every `const float*` / `float*` / `int*` tensor argument is a DEVICE pointer (HBM), fp32 / int32,
 *    contiguous row-major, exactly the layouts of the reference (SURVEY.md §8b "Layouts").
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, as the reference uses).
 *  - every function returns GSLIC_OK (0) or a negative gslic_status; nothing throws across the boundary.
 *    gslic_last_error() returns a thread-local, NUL-terminated description of the last failure.
 *  - the library never frees or owns caller memory; scratch is obtained through the caller's allocator
 *    callbacks in the same order as the reference (geom -> img -> binning -> sample), each at most once.
 *  - the four scratch buffers are opaque: their internal layout is private to this library (it is NOT
 *    the reference's GeometryState/ImageState/BinningState/SampleState layout) and the buffers written by
 *    gslic_rasterize_forward must be handed unchanged to gslic_rasterize_backward.
 */
#ifndef GSLIC_HIP_H_INCLUDED
#define GSLIC_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSLIC_ABI_VERSION 8

typedef enum gslic_status {
    GSLIC_OK = 0,
    GSLIC_ERR_INVALID_ARG = -1,  /* NULL where a pointer is required, negative sizes, bad degree ...       */
    GSLIC_ERR_UNSUPPORTED = -2,  /* colors_precomp / cov3D_precomp non-NULL (dead in the reference's host) */
    GSLIC_ERR_ALLOC = -3,        /* an allocator callback returned NULL for a non-zero request             */
    GSLIC_ERR_HIP = -4,          /* a HIP runtime call or kernel launch failed                            */
    GSLIC_ERR_PREFILTERED = -5   /* prefiltered=1 and a Gaussian was culled (reference: device __trap)    */
} gslic_status;

/* Replaces std::function<char*(size_t)> (cuda_rasterizer/rasterizer.h:30-33; built by resizeFunctional,
 * rasterize_points.cu:40-48).  Must return a device pointer to at least `bytes` bytes (any alignment:
 * the library re-aligns to 256 B and has already added the slack), valid until the matching backward. */
typedef char* (*gslic_alloc_fn)(void* ctx, size_t bytes);

/* Scalar arguments of CudaRasterizer::Rasterizer::forward/backward (rasterizer.h:29-98), i.e. the fields
 * of GaussianRasterizationSettings (src/rasterizer/rasterizer.h:27-73) that reach the kernels. */
typedef struct gslic_raster_params {
    int32_t P;              /* number of Gaussians (means3D.size(0))                                   */
    int32_t D;              /* active SH degree 0..3 (sh_degree_)                                      */
    int32_t M;              /* number of "rest" SH coefficients per Gaussian = sh.size(1) (15), 0 if sh empty */
    int32_t width;          /* image_width                                                            */
    int32_t height;         /* image_height                                                           */
    float tan_fovx, tan_fovy;
    float limx_neg, limx_pos, limy_neg, limy_pos; /* asymmetric Jacobian clamp (src/camera.h:63-66)    */
    float scale_modifier;   /* always 1 in the reference                                               */
    int32_t prefiltered;    /* bool                                                                    */
    int32_t debug;          /* bool: synchronise + check after every stage (CHECK_CUDA, auxiliary.h:173-180) */
    int32_t no_color;       /* bool: transmittance-only render (forward.cu:338,362,412,446,470)         */
    int32_t raw_params;     /* bool, 0 for the drop-in shim.  1: `opacities`, `scales`, `rotations` are the
                               RAW parameters (logit, log, unnormalised quaternion); sigmoid / exp / normalize of
                               renderer.cpp:57-63 + gaussian.cpp:147-175 run inside the kernels, and the backward returns
                               dL/d(raw) in dL_dopacity / dL_dscale / dL_drot (the chain LibTorch autograd would apply). */
    const uint32_t* tie_rank; /* NULL for the drop-in shim (ABI 7).  DEVICE uint32[P], optional: for a host that keeps the map's rows in a
                               PERMUTED order (trainer.GaussianModel(order="morton"): rows sorted along a space-filling curve so that what a
                               view sees is contiguous in memory), tie_rank[i] = index of row i in the ORIGINAL (insertion) order.  The
                               reference lists Gaussians of equal (tile, depth bits) by ascending index (stable 64-bit sort of the
                               index-ordered emission, rasterizer_impl.cu:395-424); with tie_rank set, equal depths are listed by ascending
                               tie_rank instead of by row index, so a permuted map renders and trains bit-identically to the unpermuted
                               one.  Read by the forward only. */
} gslic_raster_params;

/* ------------------------------------------------------------------------------------------------
 * gslic_rasterize_forward — replaces CudaRasterizer::Rasterizer::forward (rasterizer_impl.cu:312-474),
 * reached from RasterizeGaussiansCUDA (rasterize_points.cu:50-149).
 *
 *  background      [3]      accepted and ignored, like the reference (forward.cu:335,460-469)
 *  means3D         [P,3]    dc [P,1,3]    shs [P,M,3] (NULL when M==0)
 *  colors_precomp  must be NULL   cov3D_precomp must be NULL   (rasterizer.cpp:200-201 always passes empty)
 *  opacities       [P,1] post-sigmoid   scales [P,3] post-exp   rotations [P,4] post-normalise (r,x,y,z)
 *  viewmatrix, projmatrix   [16] device, element (row r, col c) at [4c+r] (src/camera.h:86,109,60)
 *  cam_pos         [3] device
 *  out_color       [3,H,W] (untouched when no_color)   out_final_T [H,W]   radii [P] int32
 *  num_rendered    host int: R = number of (Gaussian, tile) instances      } the two ints the reference
 *  num_buckets     host int: B = number of checkpoint buckets (0 if no_color) } returns (rasterizer_impl.cu:473)
 *
 * Allocator call order geom(P) -> img(N,T) -> binning(R) -> sample(B); sample is called only when
 * !no_color (rasterizer_impl.cu:355,359,401,437-447).  The host waits for the device twice (to learn R and
 * B), exactly where the reference blocks (rasterizer_impl.cu:398,442); the wait is a spin on a pinned-memory
 * mailbox a one-thread kernel writes (GSLIC_NO_MAILBOX=1: hipMemcpyAsync + hipStreamSynchronize instead).
 * P == 0 returns immediately with R = B = 0 and calls no allocator (rasterize_points.cu:110).
 */
int gslic_rasterize_forward(
    const gslic_raster_params* prm,
    gslic_alloc_fn geom_alloc, void* geom_ctx,
    gslic_alloc_fn binning_alloc, void* binning_ctx,
    gslic_alloc_fn img_alloc, void* img_ctx,
    gslic_alloc_fn sample_alloc, void* sample_ctx,
    const float* background,
    const float* means3D,
    const float* dc,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    float* out_color,
    float* out_final_T,
    int32_t* radii,
    int32_t* num_rendered,
    int32_t* num_buckets,
    void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_rasterize_forward_capacity — the same forward with NO host round trip (the reference blocks the host twice to learn R and B,
 * rasterizer_impl.cu:398,442; gslic_rasterize_forward keeps that contract for the allocator callbacks).  The caller owns the four
 * scratch buffers and passes their sizes; geom / img must hold gslic_geom_bytes(P) / gslic_img_bytes(W, H), binning and sample are
 * CAPACITIES: the number of instances / checkpoint buckets they hold is derived from their size (gslic_binning_bytes /
 * gslic_sample_bytes are monotonic), every launch is sized for that capacity and stops at the real count, which stays on the device.
 * Nothing in the call allocates, synchronises or copies to the host, so a whole training step can be captured in a hipGraph.
 *
 *  capacity_R, capacity_B   host ints: the capacities in use.  Pass THEM as R and B to gslic_rasterize_backward* (the buffer
 *                           layout depends on them); the backward stops at the real counts on the device.
 *  status                   DEVICE uint32[8], updated by the last kernel of the call (zero the words once; they accumulate):
 *                           [0] = R, [1] = B (real counts of THIS forward), [2] = its bits — 1: the instances did not fit into `binning`,
 *                           2: the buckets did not fit into `sample`, 4: prefiltered violation, 8: a scan / sort look-back wait timed out
 *                           (device preempted), 16: more than 2^31 instances — [3] += 1 when none of 1 | 2 | 8 | 16 is set (forwards
 *                           that completed), [4] += 1 always (forwards issued), [5] |= 1 << (issue index mod 32) for a forward that
 *                           did not complete, [6] / [7] = the largest R / B seen (what to size a retry from).
 *  The timeout / overflow bits of a forward live in its own geometry buffer (ABI 7): capacity-mode forwards on different buffers may run
 *  concurrently on several streams of one device, and nothing is allocated by the first call.
 *  On overflow or timeout (bit 1, 2, 8 or 16) nothing is written out of bounds, out_color / out_final_T are unspecified, and a following
 *  gslic_rasterize_backward* on these buffers does NOTHING (no gradients, no Adam update): the host re-runs the step with larger
 *  buffers once it has seen the bits.  All other arguments as gslic_rasterize_forward; results are bit-identical to it.
 */
int gslic_rasterize_forward_capacity(
    const gslic_raster_params* prm,
    char* geom_buffer, size_t geom_bytes,
    char* binning_buffer, size_t binning_bytes,
    char* img_buffer, size_t img_bytes,
    char* sample_buffer, size_t sample_bytes,
    const float* background, const float* means3D, const float* dc, const float* shs, const float* colors_precomp,
    const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos,
    float* out_color, float* out_final_T, int32_t* radii,
    int32_t* capacity_R, int32_t* capacity_B, uint32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_rasterize_backward — replaces CudaRasterizer::Rasterizer::backward (rasterizer_impl.cu:476-581),
 * reached from RasterizeGaussiansBackwardCUDA (rasterize_points.cu:151-246).
 *
 *  R, B                       the two ints gslic_rasterize_forward returned
 *  geom/binning/img/sample    the SAME memory the forward filled (rasterizer.cpp:83-98 keeps it alive)
 *  dL_dpix        [3,H,W]     gradient of the loss w.r.t. out_color
 *  outputs (each [P,*], EVERY row is written — rows of invisible Gaussians get exact zeros — so the caller
 *  may pass uninitialised memory; the reference instead requires ten zero-filled tensors,
 *  rasterize_points.cu:192-201):
 *    dL_dmean2D [P,3] (NDC-scaled, z = 0)   dL_dconic [P,2,2] (x,y,-,w slots; slot 2 = 0)   dL_dopacity [P,1]
 *    dL_dcolor [P,3]   dL_dmean3D [P,3]   dL_dcov3D [P,6]   dL_ddc [P,1,3]   dL_dsh [P,M,3]
 *    dL_dscale [P,3]   dL_drot [P,4]
 *  dL_dmean2D, dL_dconic, dL_dcolor and dL_dcov3D may be NULL (the host discards them); the others are required
 *  (dL_dsh may be NULL when M == 0).
 *  shs == NULL skips the whole SH backward like the reference's `if (shs)` (backward.cu:352): dL_ddc = 0.
 */
int gslic_rasterize_backward(
    const gslic_raster_params* prm,
    int32_t R, int32_t B,
    const float* background,
    const float* means3D,
    const float* dc,
    const float* shs,
    const float* colors_precomp,
    const float* scales,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    const int32_t* radii,
    char* geom_buffer,
    char* binning_buffer,
    char* img_buffer,
    char* sample_buffer,
    const float* dL_dpix,
    float* dL_dmean2D,
    float* dL_dconic,
    float* dL_dopacity,
    float* dL_dcolor,
    float* dL_dmean3D,
    float* dL_dcov3D,
    float* dL_ddc,
    float* dL_dsh,
    float* dL_dscale,
    float* dL_drot,
    float lambda_erank,
    void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_rasterize_backward_adam — single-GPU fast path (no reference counterpart; SURVEY.md §8f row 2 taken one step further):
 * exactly gslic_rasterize_backward with raw_params = 1 followed by gslic_adam_update_groups(visible = radii > 0) over the six
 * parameter groups, but the Adam update is applied inside the per-Gaussian backward kernel while the gradients are still in
 * registers / LDS, so the 236 B/Gaussian of gradients are neither written nor re-read (and no second launch).  Results are
 * bit-identical to the two-call sequence.  The six gradient output pointers may be NULL (nothing is written) or non-NULL.
 * Group order everywhere: xyz, features_dc, features_rest, opacity, scaling, rotation (src/gaussian.cpp:399-418);
 * param[0] / [1] / [2] / [4] / [5] must be the tensors passed as means3D / dc / shs / scales / rotations.
 * Not usable when gradients must be exchanged between GPUs before the update.
 */
typedef struct gslic_adam_fused {
    float* param[6];
    float* exp_avg[6];
    float* exp_avg_sq[6];
    float lr[6];
    float b1, b2, eps;
    uint8_t* visible_out;   /* optional (ABI 7) DEVICE [P]: gslic_rasterize_backward_adam also writes 1 = radii > 0 per Gaussian — the `visible` mask
                               of renderer.cpp:85 that the host reads for its statistics — so that no compare kernel has to be launched for it;
                               ignored by the other entry points that take this descriptor */
} gslic_adam_fused;
int gslic_rasterize_backward_adam(
    const gslic_raster_params* prm, int32_t R, int32_t B,
    const float* background, const float* means3D, const float* dc, const float* shs, const float* colors_precomp,
    const float* scales, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, const int32_t* radii,
    char* geom_buffer, char* binning_buffer, char* img_buffer, char* sample_buffer, const float* dL_dpix,
    float* dL_dopacity, float* dL_dmean3D, float* dL_ddc, float* dL_dsh, float* dL_dscale, float* dL_drot,
    float lambda_erank, const gslic_adam_fused* adam, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_rasterize_backward_rgb + gslic_sh_grad_from_rgb — the backward split for the N-GPU exchange (no reference counterpart: the
 * reference is single-GPU).  The SH backward (computeColorFromSH, backward.cu:27-136) is linear in the clamp-masked colour gradient
 * dRGB: dL_ddc = SH_C0 dRGB and dL_dsh[k] = c_k(dir) dRGB, with c_k a function of the view direction normalize(p - campos) only.
 * So a rank does not have to all-reduce the 48 floats of dL_ddc / dL_dsh per Gaussian (81 % of the gradient bytes): it ships dRGB
 * (12 bytes), all-gathers the other views' dRGB and camera centres, and rebuilds the SUMMED rows locally.  xGMI is point-to-point —
 * bytes on the links are what bounds the N > 1 step.
 *
 * gslic_rasterize_backward_rgb  = gslic_rasterize_backward that writes dL_drgb [P,3] (zeros for invisible Gaussians) instead of
 *                                 dL_ddc / dL_dsh; the gradients the host discards are not materialised.
 * gslic_sh_grad_from_rgb        dL_ddc [P,3], dL_dsh [P,M,3] = sum over v < n_views of view v's rows, from
 *                                 rgb_all [n_views][P][3] and campos_all [n_views][3] (device), in view order with the backward's own
 *                                 arithmetic (bit-identical to summing the per-view gslic_rasterize_backward outputs in that order).
 *                                 input_is_ddc = 1: rgb_all holds the views' dL_ddc instead (hosts that only see the reference's
 *                                 gradient tensors); dRGB is then recovered as dL_ddc * (1 / SH_C0), exact to 1 ulp.
 *                                 view_stride = 0: the dense layouts above.  view_stride > 0 (floats, >= 3 P): view v's colour gradients
 *                                 start at rgb_all + v * view_stride and its camera centre at campos_all + v * view_stride — one
 *                                 all-gather of a per-rank payload {dRGB [P,3], camera centre [3], ...} needs no unpacking
 *                                 (pass campos_all = rgb_all + 3 P).
 */
int gslic_rasterize_backward_rgb(
    const gslic_raster_params* prm, int32_t R, int32_t B,
    const float* background, const float* means3D, const float* dc, const float* shs, const float* colors_precomp,
    const float* scales, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, const int32_t* radii,
    char* geom_buffer, char* binning_buffer, char* img_buffer, char* sample_buffer, const float* dL_dpix,
    float* dL_dopacity, float* dL_dmean3D, float* dL_drgb, float* dL_dscale, float* dL_drot,
    float lambda_erank, void* stream);
/* gslic_rasterize_backward_rgb that also fills the rest of a rank's all-gather payload {dRGB [P,3], camera centre [3], visibility [P] bytes}:
 * vis_out[g] = radii[g] > 0 and campos_out[0..2] = cam_pos are written by the per-Gaussian kernel, so the host issues no compare / copy
 * launches between the backward and the collective.  Either may be NULL. */
int gslic_rasterize_backward_rgb_payload(
    const gslic_raster_params* prm, int32_t R, int32_t B,
    const float* background, const float* means3D, const float* dc, const float* shs, const float* colors_precomp,
    const float* scales, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, const int32_t* radii,
    char* geom_buffer, char* binning_buffer, char* img_buffer, char* sample_buffer, const float* dL_dpix,
    float* dL_dopacity, float* dL_dmean3D, float* dL_drgb, float* dL_dscale, float* dL_drot,
    float lambda_erank, uint8_t* vis_out, float* campos_out, void* stream);
/* gslic_rasterize_backward_rgb in row chunks: the per-Gaussian half of the backward for Gaussians [row_begin, row_end) only (row_begin a
 * multiple of 64; the gradient pointers are still indexed by the ABSOLUTE Gaussian index), the blend half (one pass over the whole image) unless
 * skip_blend.  A host that exchanges gradients calls it once per chunk — the first call with skip_blend = 0 — and puts chunk c on the wire
 * while chunk c + 1 is computed. */
int gslic_rasterize_backward_rgb_rows(
    const gslic_raster_params* prm, int32_t R, int32_t B,
    const float* background, const float* means3D, const float* dc, const float* shs, const float* colors_precomp,
    const float* scales, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, const int32_t* radii,
    char* geom_buffer, char* binning_buffer, char* img_buffer, char* sample_buffer, const float* dL_dpix,
    float* dL_dopacity, float* dL_dmean3D, float* dL_drgb, float* dL_dscale, float* dL_drot,
    float lambda_erank, int32_t row_begin, int32_t row_end, int32_t skip_blend, void* stream);
int gslic_sh_grad_from_rgb(
    int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D, const float* campos_all /*[n_views,3]*/,
    const float* rgb_all /*[n_views,P,3]*/, int32_t input_is_ddc, float* dL_ddc, float* dL_dsh, int64_t view_stride, void* stream);
/* The same rebuild with the masked Adam update of features_dc / features_rest (groups 1 and 2 of `adam`; the other groups are ignored)
 * applied straight from the rebuilt rows: the 192 B/Gaussian of dL_ddc / dL_dsh are neither written nor re-read.  `visible` = the
 * exchanged (OR-ed) mask, one byte per Gaussian.  dL_ddc / dL_dsh may be NULL (not materialised) or non-NULL (also written).
 * Bit-identical to gslic_sh_grad_from_rgb followed by gslic_adam_update_groups on the two groups. */
int gslic_sh_grad_from_rgb_adam(
    int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D, const float* campos_all, const float* rgb_all,
    int32_t input_is_ddc, const uint8_t* visible, const gslic_adam_fused* adam, float* dL_ddc, float* dL_dsh, int64_t view_stride, void* stream);
/* The whole optimiser step of an N > 1 rank in ONE launch, straight from what the exchange delivered: the views' visibility masks are OR-ed
 * here (view v's mask at vis_all + v * vis_stride bytes; vis_stride = 0: vis_all is one already OR-ed mask; the OR is also written to vis_out
 * when that is non-NULL), dL_ddc / dL_dsh are rebuilt from the views' colour gradients and consumed by the masked Adam of groups 1 and 2,
 * and — when the four summed (all-reduced) small gradients dL_dmean3D [P,3], dL_dopacity [P,1], dL_dscale [P,3], dL_drot [P,4] are given —
 * the masked Adam of groups 0, 3, 4, 5 reads them in place.  Bit-identical to gslic_sh_grad_from_rgb_adam + gslic_adam_update_groups on the
 * OR-ed mask.  (src/gaussian.cpp:697-707 is the step this replaces on every rank: set_visibility_and_N + SparseGaussianAdam::step.) */
int gslic_sh_grad_from_rgb_adam_all(
    int32_t P, int32_t D, int32_t M, int32_t n_views, const float* means3D, const float* campos_all, const float* rgb_all,
    int32_t input_is_ddc, const uint8_t* vis_all, int64_t vis_stride, uint8_t* vis_out, const gslic_adam_fused* adam,
    const float* dL_dmean3D, const float* dL_dopacity, const float* dL_dscale, const float* dL_drot, int64_t view_stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_rasterize_backward_camera — gslic_rasterize_backward plus the gradient w.r.t. the CAMERA inputs (the "cam" of the
 * north-star; no reference counterpart: the reference's autograd node returns an undefined tensor for the raster settings,
 * src/rasterizer/rasterizer.cpp:171-182).  viewmatrix, projmatrix and cam_pos are treated as the three independent inputs they
 * are at this boundary; a host that derives them from a pose chains through its own (tiny) LibTorch graph:
 *   dL_dviewmatrix [16]  same element order as viewmatrix ([4c+r]); rows 0..2 carry gradient — through t = V [p,1] (cov2D Jacobian; EXACT
 *                        through the frustum clamp: a clamped t.x = lim t.z still moves with t.z, a term the reference's parameter
 *                        gradients drop, backward.cu:225-233, and this one keeps) and through the rotation part W of T = W J (:180-197)
 *   dL_dprojmatrix [16]  rows 0, 1, 3 — through p_hom = P [p,1] -> mean2D (backward.cu:339-350)
 *   dL_dcampos     [3]   through the SH view direction normalize(p - campos) (backward.cu:27-136)
 * All three are sums over the visible Gaussians, reduced in a fixed order (bit-reproducible).  Depth ordering, culling and the
 * tile assignment are piecewise constant in the camera and contribute nothing, as for every other input.  All device pointers.
 */
int gslic_rasterize_backward_camera(
    const gslic_raster_params* prm, int32_t R, int32_t B,
    const float* background, const float* means3D, const float* dc, const float* shs, const float* colors_precomp,
    const float* scales, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, const int32_t* radii,
    char* geom_buffer, char* binning_buffer, char* img_buffer, char* sample_buffer, const float* dL_dpix,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
    float* dL_ddc, float* dL_dsh, float* dL_dscale, float* dL_drot, float lambda_erank,
    float* dL_dviewmatrix, float* dL_dprojmatrix, float* dL_dcampos, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_adam_update — replaces ADAM::adamUpdate / adamUpdateCUDA (cuda_rasterizer/adam.cu:9-66), reached
 * from adamUpdate (rasterize_points.cu:248-273) <- SparseGaussianAdam::custom_step (optim_utils.h:102-137).
 * In place on param / exp_avg / exp_avg_sq, all [N,M]; rows with visible[g]==0 are left untouched.
 * No bias correction, p += -lr*m/(sqrt(v)+eps).  `visible` is one byte per Gaussian (torch bool).
 */
int gslic_adam_update(
    float* param, const float* param_grad, float* exp_avg, float* exp_avg_sq,
    const uint8_t* visible,
    float lr, float b1, float b2, float eps,
    uint32_t N, uint32_t M, void* stream);

/* One launch over several parameter groups (the six groups of gaussian.cpp:399-418) — same arithmetic as
 * n_groups calls of gslic_adam_update; exists to remove five launches and the per-group grad.clone()
 * (optim_utils.h:130).  `groups` is a HOST array. */
typedef struct gslic_adam_group {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    float lr; uint32_t M;
} gslic_adam_group;
int gslic_adam_update_groups(
    const gslic_adam_group* groups, int32_t n_groups,
    const uint8_t* visible, float b1, float b2, float eps, uint32_t N, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_fusedssim_forward / _backward — replace fusedssim / fusedssim_backward and their kernels
 * (src/fused-ssim/ssim.cu:186-441).  Images are [B,CH,H,W]; 11-tap separable Gaussian window, zero padding.
 * Forward writes ssim_map and, when the three dm_* pointers are non-NULL (train=true), the partial
 * derivative maps; all outputs are fully overwritten (no pre-zeroing needed).
 */
int gslic_fusedssim_forward(
    int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2,
    const float* img1, const float* img2,
    float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12,
    void* stream);
int gslic_fusedssim_backward(
    int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2,
    const float* img1, const float* img2, const float* dL_dmap,
    const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
    float* dL_dimg1, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_l1_ssim_loss_forward / _backward — SURVEY.md §8f row 2: the loss of optimize() (src/gaussian.cpp:685-691),
 *   loss = (1 - lambda) * mean|img - gt| + lambda * (1 - mean(ssim_map(img, gt))),
 * with l1_loss (loss_utils.h:30-33) folded into the fused-SSIM kernels.  Forward writes the three derivative maps (as
 * gslic_fusedssim_forward with train = 1) and terms[0] = mean|img - gt|, terms[1] = mean ssim (DEVICE floats, reduced in a fixed
 * order: bit-reproducible); `partials` is DEVICE scratch of gslic_loss_partials_count(B,CH,H,W) floats.  Backward writes
 * dL/dimg for dL/dloss = 1: (1-lambda)/N sign(img-gt) - lambda/N (conv(dm_dmu1) + 2 img conv(dm_dsigma1_sq) + gt conv(dm_dsigma12)).
 */
int64_t gslic_loss_partials_count(int32_t B, int32_t CH, int32_t H, int32_t W);
int gslic_l1_ssim_loss_forward(
    int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, const float* img, const float* gt,
    float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, float* partials, float* terms /*[2] device*/, void* stream);
int gslic_l1_ssim_loss_backward(
    int32_t B, int32_t CH, int32_t H, int32_t W, float lambda_dssim, const float* img, const float* gt,
    const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg, void* stream);
/* Both of them as TWO launches instead of three (ABI 7): the fixed-order reduction of the forward's partial sums into `terms` rides on the
 * first workgroup of the backward kernel.  Same bits in the maps, in terms and in dL_dimg as the two calls above. */
int gslic_l1_ssim_loss_forward_backward(
    int32_t B, int32_t CH, int32_t H, int32_t W, float C1, float C2, float lambda_dssim, const float* img, const float* gt,
    float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, float* partials, float* terms /*[2] device*/, float* dL_dimg, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_knn_mean_dist2 — replaces SimpleKNN::knn (src/simple-knn/simple_knn.cu:185-221) behind distCUDA2
 * (src/simple-knn/spatial.cu:15-26): mean_dists[i] = mean of the 3 smallest squared distances from point i
 * to the other points (FLT_MAX stands in for a missing neighbour when P < 4, as in the reference).
 * Scratch comes from the caller's allocator (the reference cudaMalloc's internally, simple_knn.cu:187-216).
 */
int gslic_knn_mean_dist2(
    int32_t P, const float* points, float* mean_dists,
    gslic_alloc_fn scratch_alloc, void* scratch_ctx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * gslic_extend_select / gslic_extend_emit — SURVEY.md §8f row 1: the device-side replacement of the point selection inside
 * extend() (src/gaussian.cpp:536-603; CPU unordered_map<string,...> dedupe at :557-572) and of the construction of the new
 * Gaussians' parameter rows (:605-626).  The caller renders the latest camera with no_color = 1 first (final_T, :501-507).
 *
 *  select: points [n,3] world, depths_rsp [n] (sensor range, :534), R_cw [9] row-major and t_cw [3] (DEVICE), intrinsics,
 *          final_T [H,W].  A point survives iff it is the nearest (smallest camera z, lowest index on ties) of the points
 *          falling into its pixel, the pixel is inside the image, depths_rsp > 0 and 1 - final_T < 0.99.
 *          Returns *count (host; the call synchronises the stream once) and two device arrays inside the scratch buffer:
 *          keep_flags [n] (0/1) and keep_pos [n] (exclusive rank among survivors, ascending point index — the reference's
 *          order is unordered_map iteration order, i.e. unspecified).
 *  emit:   writes `count` rows at the given row pointers (= the model's tensors offset to row P): xyz = point,
 *          dc = (colour - 0.5)/C0, rest = 0 [M,3], opacity = inverse_sigmoid(0.1), scaling = log(scaling_scale*range/focal) x3,
 *          rotation = (1,0,0,0).  Growing the tensors (densificationPostfix, :426-497) stays with the host.
 */
int gslic_extend_select(
    int32_t n, const float* points, const float* depths_rsp, const float* R_cw, const float* t_cw,
    float fx, float fy, float cx, float cy, int32_t width, int32_t height, const float* final_T,
    gslic_alloc_fn scratch_alloc, void* scratch_ctx,
    uint32_t** keep_flags, uint32_t** keep_pos, int32_t* count, void* stream);
int gslic_extend_emit(
    int32_t n, const uint32_t* keep_flags, const uint32_t* keep_pos, const float* points, const float* colors,
    const float* depths_rsp, float scaling_scale, float focal, int32_t M,
    float* xyz, float* dc, float* rest, float* opacity, float* scaling, float* rotation, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Introspection / measurement (no reference counterpart; used by bench.py and the tests).
 */
int gslic_abi_version(void);
const char* gslic_last_error(void);

/* Arithmetic of the two blend kernels (process-wide; returns the previous mode).
 * 1 (default) = strict: the reference's operations in source order (forward.cu:424-445, backward.cu:538-581) — absolute pixel
 * coordinates, every product of the power rounded on its own (no contraction), exp() as hipcc lowers expf() (the same instruction
 * sequence, inlined without its two range checks), opacity * exp, (colour * alpha) * T.  Image / final_T / n_contrib are bit-identical
 * to the reference's kernels compiled by hipcc -ffp-contract=off for the same GPU, every blend / skip decision of the backward is the
 * reference's, gradients agree to fp32 summation order (tests/test_fullsize_reference_gpu.py).  This is what bench.py times.
 * 0 = fast (opt-in: GSLIC_FAST_MATH=1 in the environment, or this call): conic pre-scaled by log2(e), tile-relative coordinates,
 * alpha = v_exp_f32(p2 + log2 opacity), fused multiply-adds — within fp32 rounding of the reference, which flips the alpha < 1/255
 * and T < 1e-4 decisions of a few (pixel, Gaussian) pairs per million (counted in the same test). */
int gslic_set_math_mode(int32_t strict);

/* How the forward groups the (Gaussian, tile) instances by tile — the tile half of cub::DeviceRadixSort::SortPairs,
 * rasterizer_impl.cu:419-424; the lists are the reference's bit for bit on either path.  Returns the previous mode.
 * 0 = auto (default; GSLIC_BINNING=auto): block-aggregated atomics on the tiles' list cursors while the map's row order keeps consecutive
 *     Gaussians on neighbouring tiles (measured per forward: the global atomics the binning kernel needed per instance), the stable radix sort
 *     otherwise, above 36864 tiles (the block histogram's 144 KB of LDS) and on a device / driver that does not grant a kernel more than
 *     64 KB of dynamic LDS while the image has more than 14336 tiles;  1 = always the radix sort (GSLIC_BINNING=radix);  2 = atomics
 *     whenever the tile count allows (GSLIC_BINNING=atomic: a refused LDS grant is then an error, GSLIC_ERR_HIP, instead of a fallback).
 *     Any other value only reads the mode.  Process-wide: the mode is atomic, and setting it makes EVERY host thread forget what its last
 *     forwards measured (ABI 8; until ABI 7 only the calling thread's state was reset). */
int gslic_set_binning_mode(int32_t mode);

/* Which grouping the calling thread's LAST forward with at least one instance took (ABI 8): GSLIC_BINNING_PATH_NONE before any,
 * _RADIX (sort_hist / sort_scatter / finalize_ranges launches) or _ATOMIC (tile_hist / tile_scan / tile_bin launches).  sampled_atomics /
 * sampled_instances (either may be NULL): what the binning kernel's sampled workgroups reported for that forward — global atomics and
 * instances; `auto` keeps the atomic path while atomics <= 0.4 * instances.  Zero on the radix path and in capacity mode (nothing is read back). */
#define GSLIC_BINNING_PATH_NONE 0
#define GSLIC_BINNING_PATH_RADIX 1
#define GSLIC_BINNING_PATH_ATOMIC 2
int gslic_get_binning_path(uint32_t* sampled_atomics, uint32_t* sampled_instances);

/* The size a host allocator should round a scratch request of `bytes` up to (ABI 8; the allocator callbacks of this repository's hosts —
 * _lib.TensorAllocator, the LibTorch shim's resize callbacks — all call it): requests above 1 MB go to 32 MB granules, above 64 MB to
 * granules of min(half the largest power of two in the request, 256 MB).  A caching allocator then keeps serving the same block while R and B
 * drift from step to step, a GROWING map pays a hipMalloc once per ~1.3x of growth below 512 MB, and the over-allocation stays below 50 % up
 * to 512 MB and below 256 MB beyond (a 2.1 GB request becomes 2.25 GB, not 3). */
size_t gslic_scratch_round_up(size_t bytes);

/* Sizes the four scratch buffers would need, for hosts that prefer to pre-size (bytes incl. slack). */
size_t gslic_geom_bytes(int32_t P);
size_t gslic_img_bytes(int32_t width, int32_t height);
size_t gslic_binning_bytes(int32_t R, int32_t no_color);
size_t gslic_sample_bytes(int32_t B);

/* Per-kernel HIP-event timing.  While enabled, every kernel launch of this library is bracketed by a
 * hipEvent pair recorded on the launch stream; gslic_profile_collect synchronises the device and adds the
 * elapsed times to per-kernel totals.  Kernel ids are stable and named by gslic_profile_kernel_name. */
#define GSLIC_PROFILE_MAX_KERNELS 48
int gslic_profile_enable(int32_t on); /* 0 = off, 1 or -1 = every kernel, otherwise bit i selects kernel id i (ids < 31) */
int gslic_profile_reset(void);
int gslic_profile_collect(void);
int gslic_profile_num_kernels(void);
const char* gslic_profile_kernel_name(int32_t id);
int gslic_profile_get(int32_t id, double* total_ms, int64_t* launches);

/* Test-only view into the opaque scratch buffers: copies stage boundaries to DEVICE arrays the caller owns
 * (any pointer may be NULL).  Used by the parity tests to compare tiles_touched / point_list / ranges with
 * the oracle bit-for-bit; not part of the drop-in surface. */
int gslic_debug_export(
    const gslic_raster_params* prm, int32_t R, int32_t B,
    const char* geom_buffer, const char* binning_buffer, const char* img_buffer, const char* sample_buffer,
    uint32_t* tiles_touched /*[P]*/, float* means2D /*[P,2]*/, float* depths /*[P]*/,
    float* conic_opacity /*[P,4]*/, float* rgb /*[P,3]*/,
    uint64_t* sorted_keys /*[R]*/, uint32_t* point_list /*[R]*/, uint32_t* ranges /*[T,2]*/,
    uint32_t* n_contrib /*[H*W] row-major image order*/, uint32_t* max_contrib /*[T]*/,
    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSLIC_HIP_H_INCLUDED */
