// camera.h — TEST stand-in for the reference's src/camera.h (which needs Eigen / OpenCV): exactly the members render() reads
// (renderer.cpp:31-50), filled by the check program from raw files.  Only on the include path of the in-tree check programs; a real host
// build uses the reference's own header.
#pragma once
#include <torch/torch.h>
class Camera
{
public:
    int image_height_ = 0, image_width_ = 0;
    float FoVx_ = 0, FoVy_ = 0;
    float limx_neg_ = 0, limx_pos_ = 0, limy_neg_ = 0, limy_pos_ = 0;
    torch::Tensor world_view_transform_, full_proj_transform_, camera_center_;
};
