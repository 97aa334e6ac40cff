// eval_driver.cpp — runs the REFERENCE's own evaluation metrics (src/loss_utils.h:30-128: l1_loss, psnr, psnr_gaussian_splatting, ssim through
// gaussian() / create_window() / _ssim()) on CPU LibTorch.  Test infrastructure (never shipped): loss_utils.h is the reference's header, read in
// place and compiled unmodified; what is supplied here are the two fused-SSIM entry points its FusedSSIMMap class names (src/fused-ssim/ssim.h:7-26),
// which the evaluation path never calls — they abort if reached.
//
//   eval_driver <dir>   reads <dir>/meta.f64 (C H W), img1.f32, img2.f32 [C,H,W]; prints "name value" lines (%.9g) for
//                       l1  psnr  psnr_gs  ssim  ssim_per_image  and writes <dir>/window.f32 (the 11x11 window create_window builds)
#include <torch/torch.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "loss_utils.h"   // /root/reference/src (include path)

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor> fusedssim(float, float, torch::Tensor&, torch::Tensor&, bool)
{
    std::fprintf(stderr, "eval_driver: fusedssim() is not part of the evaluation path\n");
    std::abort();
}
torch::Tensor fusedssim_backward(float, float, torch::Tensor&, torch::Tensor&, torch::Tensor&, torch::Tensor&, torch::Tensor&, torch::Tensor&)
{
    std::fprintf(stderr, "eval_driver: fusedssim_backward() is not part of the evaluation path\n");
    std::abort();
}

static torch::Tensor load(const std::string& path, std::vector<int64_t> shape)
{
    int64_t n = 1;
    for (auto s : shape) n *= s;
    std::vector<float> buf((size_t)n);
    std::ifstream f(path, std::ios::binary);
    f.read(reinterpret_cast<char*>(buf.data()), n * 4);
    if (!f) { std::fprintf(stderr, "eval_driver: cannot read %s\n", path.c_str()); std::exit(2); }
    return torch::from_blob(buf.data(), shape, torch::kFloat).clone();
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    const std::string d = argv[1];
    double meta[3];
    { std::ifstream f(d + "/meta.f64", std::ios::binary); f.read(reinterpret_cast<char*>(meta), sizeof(meta)); }
    const int64_t C = (int64_t)meta[0], H = (int64_t)meta[1], W = (int64_t)meta[2];
    torch::Tensor img1 = load(d + "/img1.f32", {C, H, W}), img2 = load(d + "/img2.f32", {C, H, W});
    std::printf("l1 %.9g\n", loss_utils::l1_loss(img1, img2).item<float>());
    std::printf("psnr %.9g\n", loss_utils::psnr(img1, img2).item<float>());
    std::printf("psnr_gs %.9g\n", loss_utils::psnr_gaussian_splatting(img1, img2).item<float>());
    std::printf("ssim %.9g\n", loss_utils::ssim(img1, img2, torch::kCPU).item<float>());          // evaluateVisualQuality's call (gaussian.cpp:762,800)
    torch::Tensor b1 = img1.unsqueeze(0), b2 = img2.unsqueeze(0);
    std::printf("ssim_per_image %.9g\n", loss_utils::ssim(b1, b2, torch::kCPU, 11, false).item<float>());
    torch::Tensor w = loss_utils::create_window(11, C, torch::kCPU).contiguous();
    std::ofstream o(d + "/window.f32", std::ios::binary);
    o.write(reinterpret_cast<const char*>(w.data_ptr<float>()), w.numel() * 4);
    return 0;
}
