// Not in the reference: all parameter groups of one optimiser step in ONE launch (gslic_adam_update_groups); used by
// shim/include/optim_utils.h.  Its own header so that optim_utils.h can be combined with either rasterize_points.h (the
// reference's or this repo's — both declare a default argument, so a translation unit may see only one of them).
// params[i] / grads[i] / exp_avgs[i] / exp_avg_sqs[i] are [N, M_i] fp32 tensors; lrs[i] the group's learning rate.
#pragma once
#include <ATen/ATen.h>
#include <cstdint>
#include <vector>
void adamUpdateGroups(std::vector<at::Tensor>& params, std::vector<at::Tensor>& grads, std::vector<at::Tensor>& exp_avgs,
                      std::vector<at::Tensor>& exp_avg_sqs, at::Tensor& visible, const std::vector<double>& lrs, const float b1,
                      const float b2, const float eps, const uint32_t N);
