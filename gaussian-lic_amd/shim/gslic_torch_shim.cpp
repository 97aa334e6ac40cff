// gslic_torch_shim.cpp — LibTorch side of the drop-in boundary.  Re-exports the reference's six L2 functions
// (src/rasterizer/rasterize_points.{h,cu}, src/fused-ssim/ssim.{h,cu}, src/simple-knn/spatial.{h,cu}) with their exact
// signatures; each body only unwraps pointers, provides torch-backed allocator callbacks and calls the C-ABI of
// include/gslic_hip.h.  No HIP headers, no kernels here: plain C++ compiled by g++ against LibTorch.
// Launches go to gslic::current_stream() (include/gslic_stream.h): LibTorch's current HIP stream in this build — the legacy default stream,
// like the reference's bare <<<grid, block>>>, unless the host installed a stream guard.
#include "gslic_stream.h"
#include "rasterizer/rasterize_points.h"
#include "fused-ssim/ssim.h"
#include "simple-knn/spatial.h"

#include "../../include/gslic_hip.h"

#include <c10/util/Exception.h>
#include <torch/library.h>

#include <vector>

namespace {

// Large scratch requests are rounded up by the library's rule (gslic_scratch_round_up, include/gslic_hip.h: 32 MB granules, above 64 MB
// min(half the largest power of two in the request, 256 MB)), so that the caching allocator keeps serving the same block while R and B drift
// and a growing map pays a hipMalloc once per ~1.3x of growth; _lib.TensorAllocator calls the same function.
inline size_t scratch_round_up(size_t n) { return gslic_scratch_round_up(n); }
// role of resizeFunctional (rasterize_points.cu:40-48): a byte tensor the library may (re)size once
char* resize_cb(void* ctx, size_t n)
{
    at::Tensor& t = *static_cast<at::Tensor*>(ctx);
    t.resize_({static_cast<long long>(scratch_round_up(n))});
    return reinterpret_cast<char*>(t.contiguous().data_ptr());
}

const float* fptr(const at::Tensor& t) { return t.numel() == 0 ? nullptr : t.data_ptr<float>(); }
float* fptr_mut(at::Tensor& t) { return t.numel() == 0 ? nullptr : t.data_ptr<float>(); }

void check(int rc, const char* what)
{
    TORCH_CHECK(rc == GSLIC_OK, what, " failed (", rc, "): ", gslic_last_error());
}

at::Tensor f32c(const at::Tensor& t) { return t.contiguous(); }

}  // namespace

std::tuple<int, int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                       const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
                       const int image_width, const float limx_neg, const float limx_pos, const float limy_neg, const float limy_pos,
                       const torch::Tensor& dc, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                       const bool prefiltered, const bool debug, const bool no_color)
{
    if (means3D.ndimension() != 2 || means3D.size(1) != 3) {
        AT_ERROR("means3D must have dimensions (num_points, 3)");  // rasterize_points.cu:77-80
    }
    const int P = means3D.size(0);
    const int H = image_height, W = image_width;
    int M = 0;
    if (sh.size(0) != 0) M = sh.size(1);

    auto float_opts = means3D.options().dtype(at::kFloat);
    auto int_opts = means3D.options().dtype(at::kInt);
    // the kernels write every element they own; only the degenerate cases need the reference's zero fill
    at::Tensor out_color = (P == 0 || no_color) ? at::zeros({3, H, W}, float_opts) : at::empty({3, H, W}, float_opts);
    at::Tensor out_final_T = (P == 0) ? at::zeros({H, W}, float_opts) : at::empty({H, W}, float_opts);
    at::Tensor radii = at::empty({P}, int_opts);

    auto byte_opts = means3D.options().dtype(at::kByte);
    at::Tensor geomBuffer = at::empty({0}, byte_opts), binningBuffer = at::empty({0}, byte_opts);
    at::Tensor imgBuffer = at::empty({0}, byte_opts), sampleBuffer = at::empty({0}, byte_opts);

    int rendered = 0, num_buckets = 0;
    if (P != 0) {
        gslic_raster_params prm{P, degree, M, W, H, tan_fovx, tan_fovy, limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier,
                                prefiltered ? 1 : 0, debug ? 1 : 0, no_color ? 1 : 0, /*raw_params=*/0};
        at::Tensor bg = f32c(background), m3 = f32c(means3D), col = f32c(colors), op = f32c(opacity), sc = f32c(scales),
                   rot = f32c(rotations), cov = f32c(cov3D_precomp), vm = f32c(viewmatrix), pm = f32c(projmatrix), dcc = f32c(dc),
                   shc = f32c(sh), cp = f32c(campos);
        check(gslic_rasterize_forward(&prm, resize_cb, &geomBuffer, resize_cb, &binningBuffer, resize_cb, &imgBuffer, resize_cb,
                                      &sampleBuffer, fptr(bg), fptr(m3), fptr(dcc), fptr(shc), fptr(col), fptr(op), fptr(sc), fptr(rot),
                                      fptr(cov), fptr(vm), fptr(pm), fptr(cp), fptr_mut(out_color), fptr_mut(out_final_T),
                                      radii.data_ptr<int>(), &rendered, &num_buckets, gslic::current_stream()),
              "gslic_rasterize_forward");
    }
    return std::make_tuple(rendered, num_buckets, out_color, out_final_T, radii, geomBuffer, binningBuffer, imgBuffer, sampleBuffer);
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const float limx_neg,
                               const float limx_pos, const float limy_neg, const float limy_pos, const torch::Tensor& dL_dout_color,
                               const torch::Tensor& dc, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                               const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer,
                               const torch::Tensor& imageBuffer, const int B, const torch::Tensor& sampleBuffer,
                               const float lambda_erank, const bool debug)
{
    const int P = means3D.size(0);
    const int H = dL_dout_color.size(1), W = dL_dout_color.size(2);
    int M = 0;
    if (sh.size(0) != 0) M = sh.size(1);
    auto opts = means3D.options();
    // every row is written by the library (zeros for invisible Gaussians): no torch::zeros needed unless P == 0
    auto mk = [&](std::initializer_list<int64_t> s) { return P == 0 ? at::zeros(s, opts) : at::empty(s, opts); };
    at::Tensor dL_dmeans3D = mk({P, 3}), dL_dmeans2D = mk({P, 3}), dL_dcolors = mk({P, 3}), dL_dconic = mk({P, 2, 2});
    at::Tensor dL_dopacities = mk({P, 1}), dL_dcov3D = mk({P, 6}), dL_ddc = mk({P, 1, 3}), dL_dsh = mk({P, M, 3});
    at::Tensor dL_dscales = mk({P, 3}), dL_drotations = mk({P, 4});
    if (P != 0) {
        gslic_raster_params prm{P, degree, M, W, H, tan_fovx, tan_fovy, limx_neg, limx_pos, limy_neg, limy_pos, scale_modifier, 0,
                                debug ? 1 : 0, 0, /*raw_params=*/0};
        at::Tensor bg = f32c(background), m3 = f32c(means3D), col = f32c(colors), sc = f32c(scales), rot = f32c(rotations),
                   cov = f32c(cov3D_precomp), vm = f32c(viewmatrix), pm = f32c(projmatrix), dcc = f32c(dc), shc = f32c(sh),
                   cp = f32c(campos), dl = f32c(dL_dout_color), rad = radii.contiguous();
        at::Tensor g = geomBuffer.contiguous(), b = binningBuffer.contiguous(), i = imageBuffer.contiguous(), s = sampleBuffer.contiguous();
        check(gslic_rasterize_backward(&prm, R, B, fptr(bg), fptr(m3), fptr(dcc), fptr(shc), fptr(col), fptr(sc), fptr(rot), fptr(cov),
                                       fptr(vm), fptr(pm), fptr(cp), rad.data_ptr<int>(), reinterpret_cast<char*>(g.data_ptr()),
                                       reinterpret_cast<char*>(b.data_ptr()), reinterpret_cast<char*>(i.data_ptr()),
                                       reinterpret_cast<char*>(s.data_ptr()), fptr(dl), fptr_mut(dL_dmeans2D), fptr_mut(dL_dconic),
                                       fptr_mut(dL_dopacities), fptr_mut(dL_dcolors), fptr_mut(dL_dmeans3D), fptr_mut(dL_dcov3D),
                                       fptr_mut(dL_ddc), fptr_mut(dL_dsh), fptr_mut(dL_dscales), fptr_mut(dL_drotations), lambda_erank,
                                       gslic::current_stream()),
              "gslic_rasterize_backward");
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacities, dL_dmeans3D, dL_dcov3D, dL_ddc, dL_dsh, dL_dscales, dL_drotations);
}

void adamUpdate(torch::Tensor& param, torch::Tensor& param_grad, torch::Tensor& exp_avg, torch::Tensor& exp_avg_sq,
                torch::Tensor& visible, const float lr, const float b1, const float b2, const float eps, const uint32_t N,
                const uint32_t M)
{
    // in place on param / exp_avg / exp_avg_sq: a .contiguous() copy of a strided tensor would be updated and thrown away
    // (the reference takes data_ptr of whatever it is given, rasterize_points.cu:262-272)
    TORCH_CHECK(param.is_contiguous() && exp_avg.is_contiguous() && exp_avg_sq.is_contiguous(),
                "adamUpdate: param / exp_avg / exp_avg_sq must be contiguous (they are updated in place)");
    TORCH_CHECK(param.scalar_type() == at::kFloat && exp_avg.scalar_type() == at::kFloat && exp_avg_sq.scalar_type() == at::kFloat,
                "adamUpdate: fp32 tensors expected");
    at::Tensor vis = visible.contiguous(), grad = param_grad.contiguous();
    check(gslic_adam_update(param.data_ptr<float>(), grad.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(),
                            reinterpret_cast<const uint8_t*>(vis.data_ptr<bool>()), lr, b1, b2, eps, N, M, gslic::current_stream()),
          "gslic_adam_update");
}

void adamUpdateGroups(std::vector<torch::Tensor>& params, std::vector<torch::Tensor>& grads, std::vector<torch::Tensor>& exp_avgs,
                      std::vector<torch::Tensor>& exp_avg_sqs, torch::Tensor& visible, const std::vector<double>& lrs, const float b1,
                      const float b2, const float eps, const uint32_t N)
{
    const size_t n = params.size();
    TORCH_CHECK(grads.size() == n && exp_avgs.size() == n && exp_avg_sqs.size() == n && lrs.size() == n, "adamUpdateGroups: list sizes differ");
    if (n == 0 || N == 0) return;
    std::vector<gslic_adam_group> groups(n);
    std::vector<at::Tensor> keep;  // contiguous gradient copies (if any) must outlive the launch call
    keep.reserve(n);
    for (size_t i = 0; i < n; i++) {
        TORCH_CHECK(params[i].is_contiguous() && exp_avgs[i].is_contiguous() && exp_avg_sqs[i].is_contiguous(),
                    "adamUpdateGroups: group ", i, ": param / exp_avg / exp_avg_sq must be contiguous (updated in place)");
        TORCH_CHECK(params[i].numel() % (int64_t)N == 0, "adamUpdateGroups: group ", i, ": numel is not a multiple of N");
        keep.push_back(grads[i].contiguous());
        groups[i].param = params[i].data_ptr<float>();
        groups[i].grad = keep.back().data_ptr<float>();
        groups[i].exp_avg = exp_avgs[i].data_ptr<float>();
        groups[i].exp_avg_sq = exp_avg_sqs[i].data_ptr<float>();
        groups[i].lr = (float)lrs[i];
        groups[i].M = (uint32_t)(params[i].numel() / (int64_t)N);
    }
    at::Tensor vis = visible.contiguous();
    check(gslic_adam_update_groups(groups.data(), (int32_t)n, reinterpret_cast<const uint8_t*>(vis.data_ptr<bool>()), b1, b2, eps, N, gslic::current_stream()),
          "gslic_adam_update_groups");
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
fusedssim(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, bool train)
{
    const int B = img1.size(0), CH = img1.size(1), H = img1.size(2), W = img1.size(3);
    at::Tensor a = img1.contiguous(), b = img2.contiguous();
    at::Tensor target = at::empty_like(a);
    at::Tensor dm_dmu1 = train ? at::empty_like(a) : at::empty({0}, a.options());
    at::Tensor dm_dsigma1_sq = train ? at::empty_like(a) : at::empty({0}, a.options());
    at::Tensor dm_dsigma12 = train ? at::empty_like(a) : at::empty({0}, a.options());
    check(gslic_fusedssim_forward(B, CH, H, W, C1, C2, fptr(a), fptr(b), fptr_mut(target), fptr_mut(dm_dmu1), fptr_mut(dm_dsigma1_sq),
                                  fptr_mut(dm_dsigma12), gslic::current_stream()),
          "gslic_fusedssim_forward");
    return std::make_tuple(target, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
}

torch::Tensor fusedssim_backward(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, torch::Tensor& dL_dmap,
                                 torch::Tensor& dm_dmu1, torch::Tensor& dm_dsigma1_sq, torch::Tensor& dm_dsigma12)
{
    const int B = img1.size(0), CH = img1.size(1), H = img1.size(2), W = img1.size(3);
    at::Tensor a = img1.contiguous(), b = img2.contiguous(), dl = dL_dmap.contiguous();
    at::Tensor d1 = dm_dmu1.contiguous(), d2 = dm_dsigma1_sq.contiguous(), d3 = dm_dsigma12.contiguous();
    at::Tensor out = at::empty_like(a);
    check(gslic_fusedssim_backward(B, CH, H, W, C1, C2, fptr(a), fptr(b), fptr(dl), fptr(d1), fptr(d2), fptr(d3), fptr_mut(out), gslic::current_stream()),
          "gslic_fusedssim_backward");
    return out;
}

torch::Tensor distCUDA2(const torch::Tensor& points)
{
    const int P = points.size(0);
    at::Tensor pts = points.contiguous();
    at::Tensor means = at::zeros({P}, points.options().dtype(at::kFloat));
    at::Tensor scratch = at::empty({0}, points.options().dtype(at::kByte));
    check(gslic_knn_mean_dist2(P, fptr(pts), fptr_mut(means), resize_cb, &scratch, gslic::current_stream()), "gslic_knn_mean_dist2");
    return means;
}

// ---- the same six functions as torch.ops.gslic.* so the Python tests can drive the C++ shim ----
namespace {
using T = at::Tensor;
std::tuple<int64_t, int64_t, T, T, T, T, T, T, T> op_fwd(const T& bg, const T& means3D, const T& colors, const T& opacity, const T& scales,
                                                         const T& rotations, double scale_modifier, const T& cov3D_precomp,
                                                         const T& viewmatrix, const T& projmatrix, double tan_fovx, double tan_fovy,
                                                         int64_t H, int64_t W, double lxn, double lxp, double lyn, double lyp, const T& dc,
                                                         const T& sh, int64_t degree, const T& campos, bool prefiltered, bool debug,
                                                         bool no_color)
{
    auto r = RasterizeGaussiansCUDA(bg, means3D, colors, opacity, scales, rotations, (float)scale_modifier, cov3D_precomp, viewmatrix,
                                    projmatrix, (float)tan_fovx, (float)tan_fovy, (int)H, (int)W, (float)lxn, (float)lxp, (float)lyn,
                                    (float)lyp, dc, sh, (int)degree, campos, prefiltered, debug, no_color);
    return std::make_tuple((int64_t)std::get<0>(r), (int64_t)std::get<1>(r), std::get<2>(r), std::get<3>(r), std::get<4>(r), std::get<5>(r),
                           std::get<6>(r), std::get<7>(r), std::get<8>(r));
}
std::tuple<T, T, T, T, T, T, T, T, T> op_bwd(const T& bg, const T& means3D, const T& radii, const T& colors, const T& scales,
                                             const T& rotations, double scale_modifier, const T& cov3D_precomp, const T& viewmatrix,
                                             const T& projmatrix, double tan_fovx, double tan_fovy, double lxn, double lxp, double lyn,
                                             double lyp, const T& dL, const T& dc, const T& sh, int64_t degree, const T& campos,
                                             const T& geom, int64_t R, const T& binning, const T& img, int64_t B, const T& sample,
                                             double lambda_erank, bool debug)
{
    return RasterizeGaussiansBackwardCUDA(bg, means3D, radii, colors, scales, rotations, (float)scale_modifier, cov3D_precomp, viewmatrix,
                                          projmatrix, (float)tan_fovx, (float)tan_fovy, (float)lxn, (float)lxp, (float)lyn, (float)lyp, dL,
                                          dc, sh, (int)degree, campos, geom, (int)R, binning, img, (int)B, sample, (float)lambda_erank,
                                          debug);
}
void op_adam(T param, T grad, T m, T v, T visible, double lr, double b1, double b2, double eps, int64_t N, int64_t M)
{
    adamUpdate(param, grad, m, v, visible, (float)lr, (float)b1, (float)b2, (float)eps, (uint32_t)N, (uint32_t)M);
}
std::tuple<T, T, T, T> op_ssim(double C1, double C2, T a, T b, bool train) { return fusedssim((float)C1, (float)C2, a, b, train); }
T op_ssim_bwd(double C1, double C2, T a, T b, T dl, T d1, T d2, T d3) { return fusedssim_backward((float)C1, (float)C2, a, b, dl, d1, d2, d3); }
}  // namespace

TORCH_LIBRARY(gslic, m)
{
    m.def("RasterizeGaussiansCUDA", &op_fwd);
    m.def("RasterizeGaussiansBackwardCUDA", &op_bwd);
    m.def("adamUpdate", &op_adam);
    m.def("fusedssim", &op_ssim);
    m.def("fusedssim_backward", &op_ssim_bwd);
    m.def("distCUDA2", &distCUDA2);
}
