// loss_utils_fused.h — OPTIONAL one-call replacement for the loss lines of optimize() (src/gaussian.cpp:685-691):
//
//     auto Ll1 = loss_utils::l1_loss(rendered_image, gt_image);                       //  sub, abs, mean            (loss_utils.h:30-33)
//     ssim_value = loss_utils::fused_ssim(rendered_image_unsq, gt_image_unsq);        //  map kernel + mean         (loss_utils.h:130-193)
//     auto loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim_value);     //  four scalar ops
// becomes
//     auto loss = loss_utils::l1_ssim_loss(rendered_image, gt_image, lambda_dssim);
//
// one autograd node on gslic_l1_ssim_loss_forward / _backward (two launches forward, one backward, instead of ~9 + ~9 with their autograd
// nodes).  dL/dimage is the chain's bit for bit (tests/test_vs_reference_kernels_gpu.py::test_fused_loss_gradient_is_the_reference_chain_bit_for_bit)
// times the upstream scalar.  Header-only: LibTorch + include/gslic_hip.h.  The reference's loss_utils.h keeps working unchanged next to it.
#pragma once
#include <torch/torch.h>

#include "gslic_stream.h"
#include "../../../include/gslic_hip.h"

namespace loss_utils {

struct L1SsimLossFunction : public torch::autograd::Function<L1SsimLossFunction> {
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, torch::Tensor image, torch::Tensor gt, double lambda_dssim)
    {
        TORCH_CHECK(image.dim() >= 3 && image.sizes() == gt.sizes(), "l1_ssim_loss: image / gt must be [3,H,W] (or [1,3,H,W]) of the same shape");
        torch::Tensor img = image.contiguous(), g = gt.contiguous();
        const int64_t CH = img.size(-3), H = img.size(-2), W = img.size(-1);
        auto fo = img.options().requires_grad(false);
        torch::Tensor d1 = torch::empty_like(img, fo), d2 = torch::empty_like(img, fo), d3 = torch::empty_like(img, fo), dL = torch::empty_like(img, fo);
        torch::Tensor partials = torch::empty({gslic_loss_partials_count(1, (int32_t)CH, (int32_t)H, (int32_t)W)}, fo), terms = torch::empty({2}, fo);
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;   // loss_utils.h:187-188
        int rc = gslic_l1_ssim_loss_forward(1, (int32_t)CH, (int32_t)H, (int32_t)W, C1, C2, img.data_ptr<float>(), g.data_ptr<float>(), d1.data_ptr<float>(),
                                            d2.data_ptr<float>(), d3.data_ptr<float>(), partials.data_ptr<float>(), terms.data_ptr<float>(), gslic::current_stream());
        TORCH_CHECK(rc == GSLIC_OK, "gslic_l1_ssim_loss_forward failed: ", gslic_last_error());
        rc = gslic_l1_ssim_loss_backward(1, (int32_t)CH, (int32_t)H, (int32_t)W, (float)lambda_dssim, img.data_ptr<float>(), g.data_ptr<float>(), d1.data_ptr<float>(),
                                         d2.data_ptr<float>(), d3.data_ptr<float>(), dL.data_ptr<float>(), gslic::current_stream());
        TORCH_CHECK(rc == GSLIC_OK, "gslic_l1_ssim_loss_backward failed: ", gslic_last_error());
        ctx->save_for_backward({dL});
        return (1.0 - lambda_dssim) * terms[0] + lambda_dssim * (1.0 - terms[1]);
    }
    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::tensor_list grad_outputs)
    {
        torch::Tensor dL = ctx->get_saved_variables()[0];
        return {dL * grad_outputs[0], torch::Tensor(), torch::Tensor()};
    }
};

inline torch::Tensor l1_ssim_loss(const torch::Tensor& image, const torch::Tensor& gt, double lambda_dssim)
{
    return L1SsimLossFunction::apply(image, gt, lambda_dssim);
}

}  // namespace loss_utils
