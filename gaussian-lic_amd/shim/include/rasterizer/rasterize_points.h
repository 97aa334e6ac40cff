// Drop-in for the reference's src/rasterizer/rasterize_points.h (declarations at :25-96): the same three free
// functions, same signatures, implemented in gslic_torch_shim.cpp on top of the C-ABI of include/gslic_hip.h.
// rasterizer.cpp / optim_utils.h of the reference compile against this header unchanged.
#pragma once
#include <ATen/ATen.h>
#include <cstdint>
#include <tuple>
namespace torch { using at::Tensor; }

std::tuple<int, int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors,
                       const torch::Tensor& opacity, const torch::Tensor& scales, const torch::Tensor& rotations,
                       const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                       const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const int image_height,
                       const int image_width, const float limx_neg, const float limx_pos, const float limy_neg, const float limy_pos,
                       const torch::Tensor& dc, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                       const bool prefiltered, const bool debug, const bool no_color = false);

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizeGaussiansBackwardCUDA(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii,
                               const torch::Tensor& colors, const torch::Tensor& scales, const torch::Tensor& rotations,
                               const float scale_modifier, const torch::Tensor& cov3D_precomp, const torch::Tensor& viewmatrix,
                               const torch::Tensor& projmatrix, const float tan_fovx, const float tan_fovy, const float limx_neg,
                               const float limx_pos, const float limy_neg, const float limy_pos, const torch::Tensor& dL_dout_color,
                               const torch::Tensor& dc, const torch::Tensor& sh, const int degree, const torch::Tensor& campos,
                               const torch::Tensor& geomBuffer, const int R, const torch::Tensor& binningBuffer,
                               const torch::Tensor& imageBuffer, const int B, const torch::Tensor& sampleBuffer,
                               const float lambda_erank, const bool debug);

void adamUpdate(torch::Tensor& param, torch::Tensor& param_grad, torch::Tensor& exp_avg, torch::Tensor& exp_avg_sq,
                torch::Tensor& visible, const float lr, const float b1, const float b2, const float eps, const uint32_t N,
                const uint32_t M);

#include "adam_groups.h"
