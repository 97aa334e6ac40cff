// fused_check.cpp — the fused training step driven from C++ (shim/include/gslic_fused.h): the program a Gaussian-LIC maintainer's
// optimize() loop turns into when it adopts the fused entry points.  Same file protocol as dropin_check.cpp:
//   fused_check <dir> <P> <W> <H> <deg> <iters> [timed_iters [lr_scale]]
// reads  <dir>/{xyz,scaling,rotation,opacity,dc,rest,view,proj,campos,gt}.f32 and scalars.f32 (tanfovx, tanfovy, 4 lims),
// writes <dir>/out_{image,xyz,scaling,rotation,opacity,dc,rest}.f32 after <iters> steps; with timed_iters > 0 it then times that many
// further steps (wall clock between two device synchronisations) and prints "views_per_s <v> ms_per_step <t>".
// When <dir>/frame_pts.f32 exists (+ frame_col.f32 [n,3], frame_rsp.f32 [n], frame_pose.f32 = R_cw[9] | t_cw[3] | fx fy cx cy, frame_n.f32 = n),
// one extend() with that LiDAR frame follows the <iters> steps, then <iters> more steps, and "extend inserted <k>" is printed.
// Needs no reference source: LibTorch + libgslic_hip.so only.
#include "gslic_fused.h"

#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

static torch::Tensor load(const std::string& path, std::vector<int64_t> shape)
{
    int64_t n = 1;
    for (auto s : shape) n *= s;
    std::vector<float> buf(n);
    std::ifstream f(path, std::ios::binary);
    TORCH_CHECK(f.good(), "cannot open ", path);
    f.read(reinterpret_cast<char*>(buf.data()), n * sizeof(float));
    return torch::from_blob(buf.data(), shape, torch::kFloat32).clone().to(torch::kCUDA);
}
static void save(const std::string& path, const torch::Tensor& t)
{
    torch::Tensor c = t.detach().to(torch::kCPU).contiguous();
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(c.data_ptr<float>()), c.numel() * sizeof(float));
}

int main(int argc, char** argv)
{
    TORCH_CHECK(argc >= 7 && argc <= 9, "usage: fused_check <dir> <P> <W> <H> <deg> <iters> [timed_iters [lr_scale]]");
    const std::string d = argv[1];
    const int64_t P = std::stoll(argv[2]), W = std::stoll(argv[3]), H = std::stoll(argv[4]);
    const int deg = std::stoi(argv[5]), iters = std::stoi(argv[6]), timed = argc >= 8 ? std::stoi(argv[7]) : 0;
    const float ls = argc >= 9 ? std::stof(argv[8]) : 1.0f;   // bench.py times a static synthetic scene at scaled-down rates
    const int64_t M = deg > 0 ? 15 : 0;
    torch::Tensor xyz = load(d + "/xyz.f32", {P, 3}), scaling = load(d + "/scaling.f32", {P, 3}), rotation = load(d + "/rotation.f32", {P, 4});
    torch::Tensor opacity = load(d + "/opacity.f32", {P, 1}), dc = load(d + "/dc.f32", {P, 1, 3});
    torch::Tensor rest = M > 0 ? load(d + "/rest.f32", {P, M, 3}) : torch::zeros({P, 0, 3}, torch::kCUDA);
    gslic::FusedCamera cam;
    cam.image_width = (int)W; cam.image_height = (int)H;
    cam.world_view_transform = load(d + "/view.f32", {4, 4}); cam.full_proj_transform = load(d + "/proj.f32", {4, 4}); cam.camera_center = load(d + "/campos.f32", {3});
    torch::Tensor gt = load(d + "/gt.f32", {3, H, W});
    torch::Tensor sc = load(d + "/scalars.f32", {6}).to(torch::kCPU);
    const float* s = sc.data_ptr<float>();
    cam.tanfovx = s[0]; cam.tanfovy = s[1]; cam.limx_neg = s[2]; cam.limx_pos = s[3]; cam.limy_neg = s[4]; cam.limy_pos = s[5];

    // trainingSetup (gaussian.cpp:399-418) with config/fastlivo.yaml learning rates
    gslic::FusedStep fs({xyz, dc, rest, opacity, scaling, rotation}, {1.6e-4f * ls, 2.5e-3f * ls, (float)(2.5e-3 / 20.0) * ls, 5e-2f * ls, 5e-3f * ls, 1e-3f * ls}, deg);
    {   // optional <dir>/tie_rank.f32: the rows' original indices of a map handed over in a permuted (Morton) order
        std::ifstream probe(d + "/tie_rank.f32", std::ios::binary);
        if (probe.good()) fs.set_tie_rank(load(d + "/tie_rank.f32", {P}).to(torch::kInt32));
    }
    if (std::getenv("GSLIC_CHECK_POSE")) {   // the camera-pose gradient of the initial map (before any step), six numbers
        const auto g = fs.pose_gradient(cam, gt);
        std::cout.precision(9);
        std::cout << "pose_gradient " << g[0] << " " << g[1] << " " << g[2] << " " << g[3] << " " << g[4] << " " << g[5] << std::endl;
    }
    for (int it = 0; it < iters; it++) {
        torch::Tensor terms = fs.step(cam, gt);
        std::cout << "iter " << it << " loss " << fs.loss_value(terms) << " visible " << fs.visible().sum().item<int>() << std::endl;
    }
    {
        std::ifstream probe(d + "/frame_n.f32", std::ios::binary);
        if (probe.good()) {
            const int64_t n = (int64_t)load(d + "/frame_n.f32", {1}).item<float>();
            torch::Tensor pts = load(d + "/frame_pts.f32", {n, 3}), col = load(d + "/frame_col.f32", {n, 3}), rsp = load(d + "/frame_rsp.f32", {n});
            torch::Tensor pose = load(d + "/frame_pose.f32", {16}).to(torch::kCPU);
            const float* q = pose.data_ptr<float>();
            torch::Tensor Rcw = pose.narrow(0, 0, 9).reshape({3, 3}).clone(), tcw = pose.narrow(0, 9, 3).clone();
            const int64_t k = fs.extend(cam, pts, col, rsp, Rcw, tcw, q[12], q[13], q[14], q[15]);
            std::cout << "extend inserted " << k << " size " << fs.size() << " capacity " << fs.capacity() << std::endl;
            for (int it = 0; it < iters; it++) {
                torch::Tensor terms = fs.step(cam, gt);
                std::cout << "iter " << (iters + it) << " loss " << fs.loss_value(terms) << " visible " << fs.visible().sum().item<int>() << std::endl;
            }
            xyz = fs.param(0); dc = fs.param(1); rest = fs.param(2); opacity = fs.param(3); scaling = fs.param(4); rotation = fs.param(5);
        }
    }
    save(d + "/out_image.f32", fs.image());
    save(d + "/out_xyz.f32", xyz); save(d + "/out_scaling.f32", scaling); save(d + "/out_rotation.f32", rotation);
    save(d + "/out_opacity.f32", opacity); save(d + "/out_dc.f32", dc);
    if (M > 0) save(d + "/out_rest.f32", rest);
    if (timed > 0) {
        for (int it = 0; it < 5; it++) fs.step(cam, gt);
        torch::cuda::synchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < timed; it++) fs.step(cam, gt);
        torch::cuda::synchronize();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "views_per_s " << timed / sec << " ms_per_step " << 1e3 * sec / timed << std::endl;
    }
    return 0;
}
