// renderer.cpp — drop-in replacement for the reference's src/rasterizer/renderer.cpp (renderer.cpp:21-88): the SAME render() signature and
// the same five results, so gaussian.cpp / mapping.cpp compile against it unchanged (CMake: list this file instead of the reference's).
//
// What changes underneath: the reference calls pc->getOpacity() / getScaling() / getRotation() (renderer.cpp:57-63 -> gaussian.cpp:147-175:
// sigmoid, exp, normalize as LibTorch ops, each with an autograd node) and allocates + zero-fills screenspace_points [P,3] every call.  Here the
// model's RAW leaf tensors (xyz_, features_dc_, features_rest_, opacity_, scaling_, rotation_: gaussian.h:153-158) go straight into one autograd
// node whose kernels apply the activations (gslic_raster_params::raw_params = 1) and whose backward returns the gradients w.r.t. those raw
// tensors — what autograd would have chained to.  Per view that removes ~16 LibTorch elementwise launches over P and the four gradient
// tensors the reference's node materialises only to drop them (dL_dmeans2D, dL_dcolors_precomp, dL_dconic, dL_dcov3D: rasterizer.cpp:171-182).
//
// Differences a caller could observe, both deliberate:
//   * std::get<2> (screenspace_points) is a zero-stride [P,3] view of one zero row, not a fresh requires_grad tensor: no caller of render()
//     reads it or its gradient (gaussian.cpp:506,683,757,797).  -DGSLIC_RENDER_SCREENSPACE_TENSOR restores the dense zeros.
//   * use_trained_exposure is ignored exactly as in the reference (renderer.cpp never reads it).
// Image, final_T, radii and the six parameter gradients equal the reference path's up to fp32 rounding of the activation chain
// (tests/test_shim_gpu.py::test_dropin_renderer_cpp).
#include "renderer.h"   // the reference's header (src/rasterizer/renderer.h): Camera, GaussianModel, the render() declaration

#include "include/gslic_stream.h"
#include "../../include/gslic_hip.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

inline const float* fp(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }
inline char* bp(const torch::Tensor& t) { return t.numel() ? reinterpret_cast<char*>(t.data_ptr()) : nullptr; }
inline void check(int rc, const char* what) { TORCH_CHECK(rc == GSLIC_OK, what, " failed (", rc, "): ", gslic_last_error()); }
// Large scratch requests are rounded up by the library's rule (gslic_scratch_round_up, include/gslic_hip.h: 32 MB granules, above 64 MB
// min(half the largest power of two in the request, 256 MB)), so that the caching allocator keeps serving the same block while R and B drift
// and a growing map pays a hipMalloc once per ~1.3x of growth; _lib.TensorAllocator calls the same function.
inline size_t scratch_round_up(size_t n) { return gslic_scratch_round_up(n); }
// resizeFunctional of rasterize_points.cu:40-48: the callback grows one byte tensor and returns its storage
char* resize_cb(void* ctx, size_t n)
{
    torch::Tensor& t = *static_cast<torch::Tensor*>(ctx);
    t.resize_({(int64_t)scratch_round_up(n)});
    return reinterpret_cast<char*>(t.data_ptr());
}

struct RawRasterize : public torch::autograd::Function<RawRasterize> {
    static tensor_list forward(AutogradContext* ctx, torch::Tensor xyz, torch::Tensor dc, torch::Tensor rest, torch::Tensor opacity,
                               torch::Tensor scaling, torch::Tensor rotation, torch::Tensor bg, torch::Tensor view, torch::Tensor proj,
                               torch::Tensor campos, int64_t H, int64_t W, double tanfovx, double tanfovy, double limx_neg, double limx_pos,
                               double limy_neg, double limy_pos, double scale_modifier, int64_t sh_degree, bool no_color, double lambda_erank)
    {
        TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "means3D must have dimensions (num_points, 3)");   // rasterize_points.cu:77-80
        const int64_t P = xyz.size(0);
        auto fo = xyz.options().requires_grad(false);
        xyz = xyz.contiguous(); dc = dc.contiguous(); rest = rest.contiguous(); opacity = opacity.contiguous();
        scaling = scaling.contiguous(); rotation = rotation.contiguous();
        bg = bg.contiguous(); view = view.contiguous(); proj = proj.contiguous(); campos = campos.contiguous();
        gslic_raster_params rp{};
        rp.P = (int32_t)P; rp.D = (int32_t)sh_degree; rp.M = rest.numel() ? (int32_t)rest.size(1) : 0; rp.width = (int32_t)W; rp.height = (int32_t)H;
        rp.tan_fovx = (float)tanfovx; rp.tan_fovy = (float)tanfovy;
        rp.limx_neg = (float)limx_neg; rp.limx_pos = (float)limx_pos; rp.limy_neg = (float)limy_neg; rp.limy_pos = (float)limy_pos;
        rp.scale_modifier = (float)scale_modifier; rp.no_color = no_color ? 1 : 0; rp.raw_params = 1;
        torch::Tensor color = (P == 0 || no_color) ? torch::zeros({3, H, W}, fo) : torch::empty({3, H, W}, fo);   // rasterize_points.cu:95-97
        torch::Tensor final_T = P == 0 ? torch::zeros({H, W}, fo) : torch::empty({H, W}, fo);
        torch::Tensor radii = torch::empty({P}, fo.dtype(torch::kInt32));
        auto bytes = fo.dtype(torch::kByte);
        torch::Tensor geom = torch::empty({0}, bytes), binning = torch::empty({0}, bytes), img = torch::empty({0}, bytes), sample = torch::empty({0}, bytes);
        int32_t R = 0, B = 0;
        if (P != 0)
            check(gslic_rasterize_forward(&rp, resize_cb, &geom, resize_cb, &binning, resize_cb, &img, resize_cb, &sample, fp(bg), fp(xyz), fp(dc), fp(rest),
                                          nullptr, fp(opacity), fp(scaling), fp(rotation), nullptr, fp(view), fp(proj), fp(campos),
                                          color.data_ptr<float>(), final_T.data_ptr<float>(), radii.data_ptr<int32_t>(), &R, &B, gslic::current_stream()),
                  "gslic_rasterize_forward");
        ctx->save_for_backward({xyz, dc, rest, opacity, scaling, rotation, bg, view, proj, campos, radii, geom, binning, img, sample});
        ctx->saved_data["R"] = (int64_t)R; ctx->saved_data["B"] = (int64_t)B; ctx->saved_data["H"] = H; ctx->saved_data["W"] = W;
        ctx->saved_data["tanfovx"] = tanfovx; ctx->saved_data["tanfovy"] = tanfovy;
        ctx->saved_data["limx_neg"] = limx_neg; ctx->saved_data["limx_pos"] = limx_pos; ctx->saved_data["limy_neg"] = limy_neg; ctx->saved_data["limy_pos"] = limy_pos;
        ctx->saved_data["scale_modifier"] = scale_modifier; ctx->saved_data["sh_degree"] = sh_degree; ctx->saved_data["lambda_erank"] = lambda_erank;
        ctx->mark_non_differentiable({radii, final_T});
        return {color, radii, final_T};
    }

    static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs)
    {
        auto sv = ctx->get_saved_variables();
        const torch::Tensor &xyz = sv[0], &dc = sv[1], &rest = sv[2], &opacity = sv[3], &scaling = sv[4], &rotation = sv[5], &bg = sv[6], &view = sv[7],
                            &proj = sv[8], &campos = sv[9], &radii = sv[10];
        torch::Tensor geom = sv[11], binning = sv[12], img = sv[13], sample = sv[14];
        const int64_t P = xyz.size(0), H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt();
        torch::Tensor dL = grad_outputs[0].contiguous();   // only d/dcolor flows back (rasterizer.cpp:136-138)
        auto fo = xyz.options().requires_grad(false);
        // every row of the six outputs is written by the kernel (invisible Gaussians get exact zeros): no torch::zeros (rasterize_points.cu:192-201)
        torch::Tensor g_xyz = torch::empty_like(xyz, fo), g_dc = torch::empty_like(dc, fo), g_rest = torch::empty_like(rest, fo);
        torch::Tensor g_op = torch::empty_like(opacity, fo), g_sc = torch::empty_like(scaling, fo), g_rot = torch::empty_like(rotation, fo);
        if (P != 0) {
            gslic_raster_params rp{};
            rp.P = (int32_t)P; rp.D = (int32_t)ctx->saved_data["sh_degree"].toInt(); rp.M = rest.numel() ? (int32_t)rest.size(1) : 0;
            rp.width = (int32_t)W; rp.height = (int32_t)H;
            rp.tan_fovx = (float)ctx->saved_data["tanfovx"].toDouble(); rp.tan_fovy = (float)ctx->saved_data["tanfovy"].toDouble();
            rp.limx_neg = (float)ctx->saved_data["limx_neg"].toDouble(); rp.limx_pos = (float)ctx->saved_data["limx_pos"].toDouble();
            rp.limy_neg = (float)ctx->saved_data["limy_neg"].toDouble(); rp.limy_pos = (float)ctx->saved_data["limy_pos"].toDouble();
            rp.scale_modifier = (float)ctx->saved_data["scale_modifier"].toDouble(); rp.raw_params = 1;
            auto w = [](torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; };
            check(gslic_rasterize_backward(&rp, (int32_t)ctx->saved_data["R"].toInt(), (int32_t)ctx->saved_data["B"].toInt(), fp(bg), fp(xyz), fp(dc), fp(rest),
                                           nullptr, fp(scaling), fp(rotation), nullptr, fp(view), fp(proj), fp(campos), radii.data_ptr<int32_t>(), bp(geom),
                                           bp(binning), bp(img), bp(sample), fp(dL), /*dL_dmean2D*/ nullptr, /*dL_dconic*/ nullptr, w(g_op),
                                           /*dL_dcolor*/ nullptr, w(g_xyz), /*dL_dcov3D*/ nullptr, w(g_dc), w(g_rest), w(g_sc), w(g_rot),
                                           (float)ctx->saved_data["lambda_erank"].toDouble(), gslic::current_stream()),
                  "gslic_rasterize_backward");
        } else {
            for (auto* t : {&g_xyz, &g_dc, &g_rest, &g_op, &g_sc, &g_rot}) t->zero_();
        }
        torch::Tensor none;
        return {g_xyz, g_dc, g_rest, g_op, g_sc, g_rot, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
    }
};

}  // namespace

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
render(const std::shared_ptr<Camera>& viewpoint_camera,
       std::shared_ptr<GaussianModel> pc,
       torch::Tensor& bg_color,
       bool use_trained_exposure,
       bool no_color,
       float scaling_modifier)
{
    (void)use_trained_exposure;
    const float tanfovx = std::tan(viewpoint_camera->FoVx_ * 0.5f);   // renderer.cpp:31-32
    const float tanfovy = std::tan(viewpoint_camera->FoVy_ * 0.5f);
    auto res = RawRasterize::apply(pc->xyz_, pc->features_dc_, pc->features_rest_, pc->opacity_, pc->scaling_, pc->rotation_, bg_color,
                                   viewpoint_camera->world_view_transform_, viewpoint_camera->full_proj_transform_, viewpoint_camera->camera_center_,
                                   (int64_t)viewpoint_camera->image_height_, (int64_t)viewpoint_camera->image_width_, (double)tanfovx, (double)tanfovy,
                                   (double)viewpoint_camera->limx_neg_, (double)viewpoint_camera->limx_pos_, (double)viewpoint_camera->limy_neg_,
                                   (double)viewpoint_camera->limy_pos_, (double)scaling_modifier, (int64_t)pc->sh_degree_, no_color, (double)pc->lambda_erank_);
    torch::Tensor rendered_image = res[0], radii = res[1], rendered_final_T = res[2];
#ifdef GSLIC_RENDER_SCREENSPACE_TENSOR
    torch::Tensor screenspace_points = torch::zeros_like(pc->xyz_, pc->xyz_.options().requires_grad(true));
#else
    torch::Tensor screenspace_points = torch::zeros({1, 3}, pc->xyz_.options().requires_grad(false)).expand({pc->xyz_.size(0), 3});
#endif
    return std::make_tuple(rendered_image, rendered_final_T, screenspace_points, radii > 0, radii);
}
