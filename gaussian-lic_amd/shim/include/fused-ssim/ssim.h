// Drop-in for the reference's src/fused-ssim/ssim.h (:7-26).
#pragma once
#include <ATen/ATen.h>
#include <tuple>
namespace torch { using at::Tensor; }

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
fusedssim(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, bool train);

torch::Tensor fusedssim_backward(float C1, float C2, torch::Tensor& img1, torch::Tensor& img2, torch::Tensor& dL_dmap,
                                 torch::Tensor& dm_dmu1, torch::Tensor& dm_dsigma1_sq, torch::Tensor& dm_dsigma12);
