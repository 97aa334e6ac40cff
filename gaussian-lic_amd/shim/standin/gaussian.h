// gaussian.h — TEST stand-in for the reference's src/gaussian.h (which pulls in PCL / OpenCV / yaml): the members and accessors render()
// touches (renderer.cpp:46-63; the accessor bodies are gaussian.cpp:147-175's one-liners).  Only for the in-tree check programs.
#pragma once
#include <torch/torch.h>
class GaussianModel
{
public:
    torch::Tensor getScaling() { return torch::exp(scaling_); }
    torch::Tensor getRotation() { return torch::nn::functional::normalize(rotation_); }
    torch::Tensor getXYZ() { return xyz_; }
    torch::Tensor getFeaturesDc() { return features_dc_; }
    torch::Tensor getFeaturesRest() { return features_rest_; }
    torch::Tensor getOpacity() { return torch::sigmoid(opacity_); }
    int sh_degree_ = 3;
    double lambda_erank_ = 0.0;
    double lambda_dssim_ = 0.2;
    bool apply_exposure_ = false;
    torch::Tensor xyz_, features_dc_, features_rest_, scaling_, rotation_, opacity_;
};
