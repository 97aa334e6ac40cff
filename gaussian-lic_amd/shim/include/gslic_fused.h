// gslic_fused.h — the fused training step for the reference's C++ host (header-only: LibTorch + include/gslic_hip.h).
//
// What optimize()'s loop body does per view (src/gaussian.cpp:674-716) —
//     render() -> l1_loss + fused_ssim -> loss.backward() -> set_visibility_and_N() -> SparseGaussianAdam::step()
// — as four C-ABI calls with no autograd graph, no LibTorch elementwise launch and no gradient tensor:
//     gslic_rasterize_forward (raw_params = 1: sigmoid / exp / normalize of gaussian.cpp:147-175 inside the kernels)
//     gslic_l1_ssim_loss_forward / _backward (loss_utils.h:30-33,130-193; the loss of gaussian.cpp:685-691)
//     gslic_rasterize_backward_adam (backward + the masked Adam of optim_utils.h:102-137 in the per-Gaussian kernel)
// The arithmetic equals the operator path's up to fp32 rounding (tests/test_fused_gpu.py); the C++ and the Python host
// (gaussian-lic_amd/trainer.py:training_step_fused) issue the same calls, so their parameters agree bit for bit
// (tests/test_shim_gpu.py::test_fused_cpp_host).  Single GPU: with N > 1 the gradients must exist for the exchange (gslic_dist.h).
//
//   gslic::FusedStep fs({xyz_, features_dc_, features_rest_, opacity_, scaling_, rotation_},
//                       {1.6e-4f, 2.5e-3f, 2.5e-3f / 20, 5e-2f, 5e-3f, 1e-3f}, sh_degree_);
//   for (view : keyframes) { auto terms = fs.step(cam, gt_image); }        // terms = device [mean |img - gt|, mean ssim]
//   float loss = fs.loss_value(terms);                                     // only when the host wants the number (one sync)
#pragma once
#include <torch/torch.h>

#include <algorithm>
#include <array>
#include <vector>

#include "gslic_stream.h"
#include "../../../include/gslic_hip.h"

namespace gslic {

// the fields of Camera that reach the rasterizer (src/camera.h:38-110, renderer.cpp:28-50)
struct FusedCamera {
    int image_width = 0, image_height = 0;
    float tanfovx = 0, tanfovy = 0, limx_neg = 0, limx_pos = 0, limy_neg = 0, limy_pos = 0;
    torch::Tensor world_view_transform, full_proj_transform, camera_center;  // device fp32 [4,4] (stored transposed, camera.h:86,109), [4,4], [3]
};

class FusedStep {
public:
    // params: the six raw leaf tensors in the group order of trainingSetup (gaussian.cpp:399-418); updated IN PLACE.
    FusedStep(std::array<torch::Tensor, 6> params, std::array<float, 6> lrs, int sh_degree, float lambda_dssim = 0.2f,
              float lambda_erank = 0.0f, float b1 = 0.9f, float b2 = 0.999f, float eps = 1e-15f)
        : prm_(std::move(params)), lrs_(lrs), deg_(sh_degree), lambda_dssim_(lambda_dssim), lambda_erank_(lambda_erank), b1_(b1), b2_(b2), eps_(eps)
    {
        for (int i = 0; i < 6; i++) {
            TORCH_CHECK(prm_[i].is_cuda() && prm_[i].scalar_type() == torch::kFloat32 && prm_[i].is_contiguous(),
                        "FusedStep: parameter group ", i, " must be a contiguous fp32 device tensor (it is updated in place)");
            m_[i] = torch::zeros_like(prm_[i].detach());   // SparseGaussianAdam creates its state as zeros on the first step (optim_utils.h:116-123)
            v_[i] = torch::zeros_like(prm_[i].detach());
        }
        auto bytes = torch::TensorOptions().dtype(torch::kByte).device(prm_[0].device());
        for (auto& s : scratch_) s = torch::empty({0}, bytes);
        bg_ = torch::zeros({3}, prm_[0].options().requires_grad(false));
    }

    // continue from a SparseGaussianAdam (exp_avg / exp_avg_sq of group i), e.g. when switching an existing map over
    void adopt_state(int group, torch::Tensor exp_avg, torch::Tensor exp_avg_sq)
    {
        TORCH_CHECK(exp_avg.sizes() == prm_[group].sizes() && exp_avg_sq.sizes() == prm_[group].sizes() && exp_avg.is_contiguous() &&
                    exp_avg_sq.is_contiguous(), "FusedStep::adopt_state: shape / layout mismatch");
        m_[group] = std::move(exp_avg);
        v_[group] = std::move(exp_avg_sq);
    }
    // after densificationPostfix (gaussian.cpp:444-487) replaced the tensors: new parameters, moments extended with zeros by the host
    void rebind(std::array<torch::Tensor, 6> params, std::array<torch::Tensor, 6> exp_avg, std::array<torch::Tensor, 6> exp_avg_sq)
    {
        prm_ = std::move(params); m_ = std::move(exp_avg); v_ = std::move(exp_avg_sq);
    }
    void set_lr(int group, float lr) { lrs_[group] = lr; }
    // A host that keeps the map's rows in a PERMUTED order (e.g. sorted along a space-filling curve so that what a view sees is contiguous in
    // memory) passes the rows' ORIGINAL indices (device int32 [P]): the forward then lists Gaussians of equal depth in the original order, as the
    // reference's stable sort does (gslic_raster_params.tie_rank), and the map renders / trains bit-identically to the unpermuted one.
    // Rows appended by extend() get their row index.
    void set_tie_rank(torch::Tensor original_index)
    {
        TORCH_CHECK(original_index.defined() && original_index.device() == prm_[0].device() && original_index.numel() == prm_[0].size(0),
                    "FusedStep::set_tie_rank: expected one index per Gaussian (", prm_[0].size(0), ") on the model's device; got ", original_index.numel(),
                    " on ", original_index.device());   // (its data pointer goes straight to the kernels)
        tie_ = original_index.to(torch::kInt32).contiguous();
    }

    // extend() of gaussian.cpp:499-638 for one new LiDAR frame: transmittance-only render of `cam` (no_color, :501-507), device-side
    // selection of the points that land on not-yet-opaque pixels (nearest per pixel, gslic_extend_select), append of the new Gaussians'
    // rows (gslic_extend_emit) and of zero Adam moments.  Storage grows by capacity doubling, so an append is in place most of the time
    // instead of the reference's six torch::cat of every parameter and moment (densificationPostfix, :426-497).  After a call that
    // returned k > 0 the six parameter tensors are NEW views: fetch them with param(i).
    //   points, colors [n,3], depths_rsp [n]: device fp32; R_cw [3,3] row-major, t_cw [3] (any device); returns the number inserted.
    int64_t extend(const FusedCamera& cam, const torch::Tensor& points, const torch::Tensor& colors, const torch::Tensor& depths_rsp,
                   const torch::Tensor& R_cw, const torch::Tensor& t_cw, float fx, float fy, float cx, float cy, float scaling_scale = 1.0f)
    {
        torch::NoGradGuard ng;
        const int64_t P = prm_[0].size(0), n = points.size(0);
        const int W = cam.image_width, H = cam.image_height;
        auto fo = prm_[0].options().requires_grad(false);
        // render(no_color) through the operator entry point with LibTorch activations, exactly as renderer.cpp:57-63 feeds it
        gslic_raster_params rp{};
        rp.tie_rank = tie_.defined() ? reinterpret_cast<const uint32_t*>(tie_.data_ptr<int32_t>()) : nullptr;
        rp.P = (int32_t)P; rp.D = deg_; rp.M = prm_[2].numel() ? (int32_t)prm_[2].size(1) : 0; rp.width = W; rp.height = H;
        rp.tan_fovx = cam.tanfovx; rp.tan_fovy = cam.tanfovy;
        rp.limx_neg = cam.limx_neg; rp.limx_pos = cam.limx_pos; rp.limy_neg = cam.limy_neg; rp.limy_pos = cam.limy_pos;
        rp.scale_modifier = 1.0f; rp.no_color = 1;
        torch::Tensor op = torch::sigmoid(prm_[3]).contiguous(), sc = torch::exp(prm_[4]).contiguous();
        torch::Tensor rot = torch::nn::functional::normalize(prm_[5]).contiguous();
        torch::Tensor final_T = torch::empty({H, W}, fo), color = torch::empty({3, H, W}, fo), radii = torch::empty({P}, fo.dtype(torch::kInt32));
        int32_t R = 0, B = 0;
        check(gslic_rasterize_forward(&rp, grow_cb, &scratch_[0], grow_cb, &scratch_[1], grow_cb, &scratch_[2], grow_cb, &scratch_[3], f(bg_), f(prm_[0]),
                                      f(prm_[1]), f(prm_[2]), nullptr, f(op), f(sc), f(rot), nullptr, f(cam.world_view_transform),
                                      f(cam.full_proj_transform), f(cam.camera_center), color.data_ptr<float>(), final_T.data_ptr<float>(),
                                      radii.data_ptr<int32_t>(), &R, &B, current_stream()),
              "gslic_rasterize_forward (no_color)");
        torch::Tensor pts = points.to(fo).contiguous(), col = colors.to(fo).contiguous(), rsp = depths_rsp.to(fo).contiguous();
        torch::Tensor Rc = R_cw.to(fo).contiguous(), tc = t_cw.to(fo).contiguous();
        torch::Tensor sel_scratch = torch::empty({0}, fo.dtype(torch::kByte));
        uint32_t *flags = nullptr, *pos = nullptr;
        int32_t count = 0;
        check(gslic_extend_select((int32_t)n, f(pts), f(rsp), f(Rc), f(tc), fx, fy, cx, cy, W, H, f(final_T), grow_cb, &sel_scratch, &flags, &pos, &count,
                                  current_stream()),
              "gslic_extend_select");
        if (count == 0) return 0;
        reserve(P + count);
        const int64_t M = buf_[2].numel() ? buf_[2].size(1) : 0;
        auto row = [&](int g) { return buf_[g].numel() ? buf_[g].data_ptr<float>() + P * (buf_[g].numel() / buf_[g].size(0)) : nullptr; };
        check(gslic_extend_emit((int32_t)n, flags, pos, f(pts), f(col), f(rsp), scaling_scale, (fx + fy) / 2.0f, (int32_t)M, row(0), row(1), row(2), row(3),
                                row(4), row(5), current_stream()),
              "gslic_extend_emit");
        for (int g = 0; g < 6; g++) {   // new rows start with zero moments (gaussian.cpp:458-459)
            mbuf_[g].narrow(0, P, count).zero_();
            vbuf_[g].narrow(0, P, count).zero_();
        }
        if (tie_.defined()) tie_ = torch::cat({tie_, torch::arange(P, P + count, tie_.options())});
        bind(P + count);
        torch::cuda::synchronize();   // (the selection scratch is released on return)
        return count;
    }
    const torch::Tensor& param(int group) const { return prm_[group]; }
    int64_t size() const { return prm_[0].size(0); }
    int64_t capacity() const { return buf_[0].defined() ? buf_[0].size(0) : prm_[0].size(0); }

    // One optimisation step on one view.  Returns the device tensor [mean |image - gt|, mean ssim]; nothing synchronises.
    torch::Tensor step(const FusedCamera& cam, const torch::Tensor& gt_image)
    {
        torch::NoGradGuard ng;
        const int64_t P = prm_[0].size(0);
        const int W = cam.image_width, H = cam.image_height;
        TORCH_CHECK(gt_image.is_contiguous() && gt_image.dim() == 3 && gt_image.size(1) == H && gt_image.size(2) == W, "gt_image must be contiguous [3,H,W]");
        gslic_raster_params rp{};
        rp.tie_rank = tie_.defined() ? reinterpret_cast<const uint32_t*>(tie_.data_ptr<int32_t>()) : nullptr;
        rp.P = (int32_t)P; rp.D = deg_; rp.M = prm_[2].numel() ? (int32_t)prm_[2].size(1) : 0; rp.width = W; rp.height = H;
        rp.tan_fovx = cam.tanfovx; rp.tan_fovy = cam.tanfovy;
        rp.limx_neg = cam.limx_neg; rp.limx_pos = cam.limx_pos; rp.limy_neg = cam.limy_neg; rp.limy_pos = cam.limy_pos;
        rp.scale_modifier = 1.0f; rp.raw_params = 1;
        auto fo = prm_[0].options().requires_grad(false);
        if (!image_.defined() || image_.size(1) != H || image_.size(2) != W) {
            image_ = torch::empty({3, H, W}, fo);
            final_T_ = torch::empty({H, W}, fo);
            for (auto& d : dm_) d = torch::empty({3, H, W}, fo);
            dL_dimage_ = torch::empty({3, H, W}, fo);
            partials_ = torch::empty({gslic_loss_partials_count(1, 3, H, W)}, fo);
        }
        if (!radii_.defined() || radii_.size(0) != P) radii_ = torch::empty({P}, fo.dtype(torch::kInt32));
        torch::Tensor terms = torch::empty({2}, fo);
        const float *xyz = f(prm_[0]), *dc = f(prm_[1]), *rest = f(prm_[2]), *op = f(prm_[3]), *sc = f(prm_[4]), *rot = f(prm_[5]);
        const float *view = f(cam.world_view_transform), *proj = f(cam.full_proj_transform), *cpos = f(cam.camera_center);
        int32_t R = 0, B = 0;
        check(gslic_rasterize_forward(&rp, grow_cb, &scratch_[0], grow_cb, &scratch_[1], grow_cb, &scratch_[2], grow_cb, &scratch_[3], f(bg_), xyz, dc,
                                      rest, nullptr, op, sc, rot, nullptr, view, proj, cpos, image_.data_ptr<float>(), final_T_.data_ptr<float>(),
                                      radii_.data_ptr<int32_t>(), &R, &B, current_stream()),
              "gslic_rasterize_forward");
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;   // loss_utils.h:187-188
        // (forward + backward of the loss in two launches: the reduction of the partial sums rides on the backward kernel)
        check(gslic_l1_ssim_loss_forward_backward(1, 3, H, W, C1, C2, lambda_dssim_, f(image_), f(gt_image), dm_[0].data_ptr<float>(),
                                                  dm_[1].data_ptr<float>(), dm_[2].data_ptr<float>(), partials_.data_ptr<float>(),
                                                  terms.data_ptr<float>(), dL_dimage_.data_ptr<float>(), current_stream()),
              "gslic_l1_ssim_loss_forward_backward");
        gslic_adam_fused ad{};
        for (int i = 0; i < 6; i++) {
            const bool on = prm_[i].numel() != 0;   // features_rest is [P,0,3] at SH degree 0: an empty group is a no-op, as in the reference
            ad.param[i] = on ? prm_[i].data_ptr<float>() : nullptr;
            ad.exp_avg[i] = on ? m_[i].data_ptr<float>() : nullptr;
            ad.exp_avg_sq[i] = on ? v_[i].data_ptr<float>() : nullptr;
            ad.lr[i] = lrs_[i];
        }
        ad.b1 = b1_; ad.b2 = b2_; ad.eps = eps_;
        check(gslic_rasterize_backward_adam(&rp, R, B, f(bg_), xyz, dc, rest, nullptr, sc, rot, nullptr, view, proj, cpos, radii_.data_ptr<int32_t>(),
                                            cptr(scratch_[0]), cptr(scratch_[1]), cptr(scratch_[2]), cptr(scratch_[3]), f(dL_dimage_), nullptr, nullptr,
                                            nullptr, nullptr, nullptr, nullptr, lambda_erank_, &ad, current_stream()),
              "gslic_rasterize_backward_adam");
        return terms;
    }

    // Gradient of the training loss w.r.t. a LEFT se(3) increment of the camera pose, T_cw <- exp(xi^) T_cw, xi = (rho, phi) — the "cam" of the
    // north-star; the reference's autograd node returns nothing for its camera inputs (rasterizer.cpp:171-182).  Forward + loss kernels +
    // gslic_rasterize_backward_camera (the map is NOT updated), then the chain of gaussian-lic_amd/camera.py:Camera.pose_gradient:
    //   G = dL/dV + Proj^T dL/d(Proj V) (top three rows; Proj = (Proj V) V^-1),  G_R = G[:, :3] - t g_c^T,  G_t = G[:, 3] - R g_c,
    //   dL/drho = G_t,  dL/dphi = vee(M - M^T) + t x G_t,  M = G_R R^T.          Returns {d/drho[3], d/dphi[3]}; synchronises (35 floats to the host).
    std::array<double, 6> pose_gradient(const FusedCamera& cam, const torch::Tensor& gt_image)
    {
        torch::NoGradGuard ng;
        const int64_t P = prm_[0].size(0);
        const int W = cam.image_width, H = cam.image_height;
        gslic_raster_params rp{};
        rp.tie_rank = tie_.defined() ? reinterpret_cast<const uint32_t*>(tie_.data_ptr<int32_t>()) : nullptr;
        rp.P = (int32_t)P; rp.D = deg_; rp.M = prm_[2].numel() ? (int32_t)prm_[2].size(1) : 0; rp.width = W; rp.height = H;
        rp.tan_fovx = cam.tanfovx; rp.tan_fovy = cam.tanfovy;
        rp.limx_neg = cam.limx_neg; rp.limx_pos = cam.limx_pos; rp.limy_neg = cam.limy_neg; rp.limy_pos = cam.limy_pos;
        rp.scale_modifier = 1.0f; rp.raw_params = 1;
        auto fo = prm_[0].options().requires_grad(false);
        torch::Tensor image = torch::empty({3, H, W}, fo), final_T = torch::empty({H, W}, fo), radii = torch::empty({P}, fo.dtype(torch::kInt32));
        torch::Tensor d1 = torch::empty({3, H, W}, fo), d2 = torch::empty({3, H, W}, fo), d3 = torch::empty({3, H, W}, fo), dL = torch::empty({3, H, W}, fo);
        torch::Tensor partials = torch::empty({gslic_loss_partials_count(1, 3, H, W)}, fo), terms = torch::empty({2}, fo), camg = torch::zeros({35}, fo);
        std::array<torch::Tensor, 6> g;
        for (int i = 0; i < 6; i++) g[i] = torch::empty_like(prm_[i], fo);
        const float *xyz = f(prm_[0]), *dc = f(prm_[1]), *rest = f(prm_[2]), *op = f(prm_[3]), *sc = f(prm_[4]), *rot = f(prm_[5]);
        const float *view = f(cam.world_view_transform), *proj = f(cam.full_proj_transform), *cpos = f(cam.camera_center);
        int32_t R = 0, B = 0;
        check(gslic_rasterize_forward(&rp, grow_cb, &scratch_[0], grow_cb, &scratch_[1], grow_cb, &scratch_[2], grow_cb, &scratch_[3], f(bg_), xyz, dc, rest,
                                      nullptr, op, sc, rot, nullptr, view, proj, cpos, image.data_ptr<float>(), final_T.data_ptr<float>(),
                                      radii.data_ptr<int32_t>(), &R, &B, current_stream()), "gslic_rasterize_forward");
        check(gslic_l1_ssim_loss_forward(1, 3, H, W, 0.01f * 0.01f, 0.03f * 0.03f, f(image), f(gt_image), d1.data_ptr<float>(), d2.data_ptr<float>(),
                                         d3.data_ptr<float>(), partials.data_ptr<float>(), terms.data_ptr<float>(), current_stream()), "gslic_l1_ssim_loss_forward");
        check(gslic_l1_ssim_loss_backward(1, 3, H, W, lambda_dssim_, f(image), f(gt_image), f(d1), f(d2), f(d3), dL.data_ptr<float>(), current_stream()),
              "gslic_l1_ssim_loss_backward");
        auto w = [](torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; };
        float* cg = camg.data_ptr<float>();
        check(gslic_rasterize_backward_camera(&rp, R, B, f(bg_), xyz, dc, rest, nullptr, sc, rot, nullptr, view, proj, cpos, radii.data_ptr<int32_t>(),
                                              cptr(scratch_[0]), cptr(scratch_[1]), cptr(scratch_[2]), cptr(scratch_[3]), f(dL), nullptr, nullptr, w(g[3]), nullptr,
                                              w(g[0]), nullptr, w(g[1]), w(g[2]), w(g[4]), w(g[5]), lambda_erank_, cg, cg + 16, cg + 32, current_stream()),
              "gslic_rasterize_backward_camera");
        torch::Tensor hc = camg.to(torch::kCPU), hv = cam.world_view_transform.to(torch::kCPU).contiguous(), hp = cam.full_proj_transform.to(torch::kCPU).contiguous();
        return se3_pose_gradient(hv.data_ptr<float>(), hp.data_ptr<float>(), hc.data_ptr<float>(), hc.data_ptr<float>() + 16, hc.data_ptr<float>() + 32);
    }
    // the chain alone, on host arrays in the element order of the kernels' inputs (float[16] with (r, c) at [4c + r])
    static std::array<double, 6> se3_pose_gradient(const float* view16, const float* fullproj16, const float* dview16, const float* dproj16, const float* dcampos3)
    {
        auto at = [](const float* m, int r, int c) { return (double)m[4 * c + r]; };
        double V[4][4], PV[4][4], Vi[4][4], Pj[4][4];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { V[r][c] = at(view16, r, c); PV[r][c] = at(fullproj16, r, c); }
        // V^-1 of a rigid [R | t; 0 0 0 1]: [R^T | -R^T t]
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Vi[r][c] = V[c][r]; Vi[r][3] = -(V[0][r] * V[0][3] + V[1][r] * V[1][3] + V[2][r] * V[2][3]); }
        Vi[3][0] = Vi[3][1] = Vi[3][2] = 0.0; Vi[3][3] = 1.0;
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { Pj[r][c] = 0; for (int k = 0; k < 4; k++) Pj[r][c] += PV[r][k] * Vi[k][c]; }   // Proj = (Proj V) V^-1
        double G[3][4];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) { G[r][c] = at(dview16, r, c); for (int k = 0; k < 4; k++) G[r][c] += Pj[k][r] * at(dproj16, k, c); }
        const double gc[3] = {dcampos3[0], dcampos3[1], dcampos3[2]}, t[3] = {V[0][3], V[1][3], V[2][3]};
        double GR[3][3], Gt[3], M[3][3];
        for (int j = 0; j < 3; j++) {
            for (int i = 0; i < 3; i++) GR[j][i] = G[j][i] - t[j] * gc[i];
            Gt[j] = G[j][3] - (V[j][0] * gc[0] + V[j][1] * gc[1] + V[j][2] * gc[2]);
        }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { M[i][j] = 0; for (int k = 0; k < 3; k++) M[i][j] += GR[i][k] * V[j][k]; }   // G_R R^T
        return {Gt[0], Gt[1], Gt[2], M[2][1] - M[1][2] + (t[1] * Gt[2] - t[2] * Gt[1]), M[0][2] - M[2][0] + (t[2] * Gt[0] - t[0] * Gt[2]),
                M[1][0] - M[0][1] + (t[0] * Gt[1] - t[1] * Gt[0])};
    }

    float loss_value(const torch::Tensor& terms) const   // (1 - lambda) L1 + lambda (1 - SSIM), gaussian.cpp:685-691; synchronises
    {
        torch::Tensor t = terms.to(torch::kCPU);
        return (1.0f - lambda_dssim_) * t[0].item<float>() + lambda_dssim_ * (1.0f - t[1].item<float>());
    }
    const torch::Tensor& image() const { return image_; }                 // last render [3,H,W]
    torch::Tensor visible() const { return radii_ > 0; }                  // render_pkg's visibility filter (renderer.cpp:86)
    const torch::Tensor& exp_avg(int group) const { return m_[group]; }
    const torch::Tensor& exp_avg_sq(int group) const { return v_[group]; }

private:
    // capacity storage: created by the first extend(); until then the tensors handed to the constructor are used in place
    void reserve(int64_t newP)
    {
        const int64_t P = prm_[0].size(0);
        if (buf_[0].defined() && newP <= buf_[0].size(0)) return;
        const int64_t cap = std::max<int64_t>(2 * capacity(), newP);
        for (int g = 0; g < 6; g++) {
            std::vector<int64_t> shp = prm_[g].sizes().vec();
            shp[0] = cap;
            torch::Tensor nb = torch::empty(shp, prm_[g].options().requires_grad(false)), nm = torch::zeros(shp, m_[g].options()), nv = torch::zeros(shp, v_[g].options());
            nb.narrow(0, 0, P).copy_(prm_[g].detach());
            nm.narrow(0, 0, P).copy_(m_[g]);
            nv.narrow(0, 0, P).copy_(v_[g]);
            buf_[g] = nb; mbuf_[g] = nm; vbuf_[g] = nv;
        }
        bind(P);
    }
    void bind(int64_t P)
    {
        for (int g = 0; g < 6; g++) { prm_[g] = buf_[g].narrow(0, 0, P); m_[g] = mbuf_[g].narrow(0, 0, P); v_[g] = vbuf_[g].narrow(0, 0, P); }
    }
    static const float* f(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }
    static char* cptr(torch::Tensor& t) { return t.numel() ? reinterpret_cast<char*>(t.data_ptr()) : nullptr; }
    static void check(int rc, const char* what) { TORCH_CHECK(rc == GSLIC_OK, what, " failed (", rc, "): ", gslic_last_error()); }
    // the role of resizeFunctional (rasterize_points.cu:40-48) for buffers that persist across steps: grow only, with headroom, so a
    // steady-state step allocates nothing
    static char* grow_cb(void* ctx, size_t n)
    {
        torch::Tensor& t = *static_cast<torch::Tensor*>(ctx);
        if ((size_t)t.numel() < n) t = torch::empty({(int64_t)(n + n / 8 + 256)}, t.options());
        return reinterpret_cast<char*>(t.data_ptr());
    }

    std::array<torch::Tensor, 6> prm_, m_, v_;
    std::array<torch::Tensor, 6> buf_, mbuf_, vbuf_;
    std::array<float, 6> lrs_;
    int deg_;
    float lambda_dssim_, lambda_erank_, b1_, b2_, eps_;
    torch::Tensor scratch_[4];   // geom, binning, img, sample — order of the C-ABI's allocator arguments
    torch::Tensor bg_, image_, final_T_, radii_, dm_[3], dL_dimage_, partials_, tie_;
};

}  // namespace gslic
