// Drop-in for the reference's src/simple-knn/spatial.h (:14).
#pragma once
#include <ATen/ATen.h>
namespace torch { using at::Tensor; }

torch::Tensor distCUDA2(const torch::Tensor& points);
