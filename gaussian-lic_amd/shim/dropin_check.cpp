// dropin_check.cpp — drives the REFERENCE's own host code (rasterizer/rasterizer.{h,cpp}, loss_utils.h, optim_utils.h,
// compiled in place from /root/reference, unmodified) on top of libgslic_torch_shim.so / libgslic_hip.so: a few
// iterations of the optimize() loop body (gaussian.cpp:674-716) on tensors read from raw fp32 files.
//   dropin_check <dir> <P> <W> <H> <deg> <iters>
// reads  <dir>/{xyz,scaling,rotation,opacity,dc,rest,view,proj,campos,gt}.f32 and scalars.f32 (tanfovx, tanfovy, 4 lims),
// writes <dir>/out_{image,xyz,scaling,rotation,opacity,dc,rest}.f32 after <iters> steps (image = last render).
#ifdef GSLIC_CHECK_RENDER
// render() behind the reference's own signature (src/rasterizer/renderer.h, read in place) on stand-in Camera / GaussianModel types
// (shim/standin: the reference's camera.h / gaussian.h need Eigen / OpenCV / PCL).  Linked either with the REFERENCE's renderer.cpp
// (dropin_check_render_ref) or with this repository's drop-in replacement shim/renderer.cpp (dropin_check_render).
#include "rasterizer/renderer.h"    // reference (includes rasterizer.h)
#else
#include "rasterizer/rasterizer.h"  // reference
#endif
#ifdef GSLIC_ONE_NODE_LOSS
#include "loss_utils_fused.h"       // optional one-node loss (this repo)
#endif
#include "loss_utils.h"             // reference
#include "optim_utils.h"            // reference
#ifdef GSLIC_DIST
#include "gslic_dist.h"             // the N > 1 exchange step (this repo): RANK / WORLD_SIZE / MASTER_PORT from the environment
#include <cstdlib>
#include <unistd.h>
#endif

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
    TORCH_CHECK(argc == 7, "usage: dropin_check <dir> <P> <W> <H> <deg> <iters>");
    const std::string d = argv[1];
    const int64_t P = std::stoll(argv[2]), W = std::stoll(argv[3]), H = std::stoll(argv[4]);
    const int deg = std::stoi(argv[5]), iters = std::stoi(argv[6]);
    const int64_t M = deg > 0 ? 15 : 0;
    auto leaf = [](torch::Tensor t) { return t.requires_grad_(true); };
    torch::Tensor xyz = leaf(load(d + "/xyz.f32", {P, 3})), scaling = leaf(load(d + "/scaling.f32", {P, 3}));
    torch::Tensor rotation = leaf(load(d + "/rotation.f32", {P, 4})), opacity = leaf(load(d + "/opacity.f32", {P, 1}));
    torch::Tensor dc = leaf(load(d + "/dc.f32", {P, 1, 3}));
    torch::Tensor rest = M > 0 ? leaf(load(d + "/rest.f32", {P, M, 3})) : leaf(torch::zeros({P, 0, 3}, torch::kCUDA));
#ifdef GSLIC_DIST
    const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0, world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
    const int port = std::getenv("MASTER_PORT") ? std::atoi(std::getenv("MASTER_PORT")) : 29591;
    const bool sparse = std::getenv("GSLIC_SPARSE_EXCHANGE") && std::atoi(std::getenv("GSLIC_SPARSE_EXCHANGE")) != 0;
    const bool rank1 = std::getenv("GSLIC_EXCHANGE") && std::string(std::getenv("GSLIC_EXCHANGE")) == "rank1";
    // one process per GPU: the launcher gives every rank its device through HIP_VISIBLE_DEVICES, so "cuda:0" is this rank's GPU
    auto pg = gslic::make_rccl_group("127.0.0.1", port, rank, world);
    const std::string sfx = world > 1 ? "_" + std::to_string(rank) : "";   // rank k renders view k: view_k.f32, proj_k.f32, campos_k.f32, gt_k.f32
#else
    const std::string sfx;
#endif
    torch::Tensor view = load(d + "/view" + sfx + ".f32", {4, 4}), proj = load(d + "/proj" + sfx + ".f32", {4, 4}), campos = load(d + "/campos" + sfx + ".f32", {3});
    torch::Tensor gt = load(d + "/gt" + sfx + ".f32", {3, H, W});
    torch::Tensor sc = load(d + "/scalars.f32", {6}).to(torch::kCPU);
    const float* s = sc.data_ptr<float>();
    torch::Tensor bg = torch::zeros({3}, torch::kFloat32).cuda();

    // trainingSetup (gaussian.cpp:399-418) with config/fastlivo.yaml learning rates
    std::vector<torch::Tensor> g0{xyz}, g1{dc}, g2{rest}, g3{opacity}, g4{scaling}, g5{rotation};
    SparseGaussianAdam opt(g0, 0.0, 1e-15);
    opt.param_groups()[0].options().set_lr(1.6e-4);
    opt.add_param_group(g1); opt.param_groups()[1].options().set_lr(2.5e-3);
    opt.add_param_group(g2); opt.param_groups()[2].options().set_lr(2.5e-3 / 20.0);
    opt.add_param_group(g3); opt.param_groups()[3].options().set_lr(5e-2);
    opt.add_param_group(g4); opt.param_groups()[4].options().set_lr(5e-3);
    opt.add_param_group(g5); opt.param_groups()[5].options().set_lr(1e-3);

    torch::Tensor image;
    const bool quiet = std::getenv("GSLIC_CHECK_TIME") != nullptr;   // no per-iteration .item() synchronisation: the run is timed
    auto t_start = std::chrono::steady_clock::now();
#ifdef GSLIC_CHECK_RENDER
    auto cam = std::make_shared<Camera>();
    cam->image_height_ = (int)H; cam->image_width_ = (int)W;
    cam->FoVx_ = 2.0f * std::atan(s[0]); cam->FoVy_ = 2.0f * std::atan(s[1]);   // render() takes tan(FoV / 2) of these (renderer.cpp:31-32)
    cam->limx_neg_ = s[2]; cam->limx_pos_ = s[3]; cam->limy_neg_ = s[4]; cam->limy_pos_ = s[5];
    cam->world_view_transform_ = view; cam->full_proj_transform_ = proj; cam->camera_center_ = campos;
    auto pc = std::make_shared<GaussianModel>();
    pc->sh_degree_ = deg;
    pc->xyz_ = xyz; pc->features_dc_ = dc; pc->features_rest_ = rest; pc->opacity_ = opacity; pc->scaling_ = scaling; pc->rotation_ = rotation;
#endif
    for (int it = 0; it < iters; it++) {
#ifdef GSLIC_CHECK_RENDER
        auto render_pkg = render(cam, pc, bg, pc->apply_exposure_);        // gaussian.cpp:683
        image = std::get<0>(render_pkg);
        torch::Tensor radii = std::get<4>(render_pkg);
#else
        // render() (renderer.cpp:21-88) without the Camera/GaussianModel wrappers (those need Eigen/OpenCV/PCL)
        GaussianRasterizationSettings rs((int)H, (int)W, s[0], s[1], s[2], s[3], s[4], s[5], bg, 1.0f, view, proj, deg, campos, false, false,
                                         false, 0.0f);
        GaussianRasterizer rasterizer(rs);
        auto screenspace = torch::zeros_like(xyz, torch::TensorOptions().requires_grad(true));
        torch::Tensor colors_precomp, cov3D_precomp;
        auto res = rasterizer.forward(xyz, screenspace, torch::sigmoid(opacity), dc, rest, colors_precomp, torch::exp(scaling),
                                      torch::nn::functional::normalize(rotation), cov3D_precomp);
        image = std::get<0>(res);
        torch::Tensor radii = std::get<1>(res);
#endif
        // optimize() body (gaussian.cpp:685-707)
#ifdef GSLIC_ONE_NODE_LOSS
        auto loss = loss_utils::l1_ssim_loss(image, gt, 0.2);
#else
        auto Ll1 = loss_utils::l1_loss(image, gt);
        torch::Tensor iu = image.unsqueeze(0), gu = gt.unsqueeze(0);
        auto ssim_value = loss_utils::fused_ssim(iu, gu);
        auto loss = (1.0 - 0.2) * Ll1 + 0.2 * (1.0 - ssim_value);
#endif
        loss.backward();
        auto visible = radii > 0;
#ifdef GSLIC_DIST
        // the one exchange of the N-GPU step, between loss.backward() and step() (gaussian.cpp:697-707): SUM of the gradients, OR of the masks
        if (rank1) visible = gslic::exchange_gradients_rank1(*pg, {xyz, dc, rest, opacity, scaling, rotation}, visible, campos, deg);   // 11 floats all-reduced + 3 all-gathered
        else visible = gslic::exchange_gradients(*pg, {xyz, dc, rest, opacity, scaling, rotation}, visible, sparse);
#endif
        opt.set_visibility_and_N(visible, xyz.size(0));
        opt.step();
        opt.zero_grad(true);
        if (!quiet) std::cout << "iter " << it << " loss " << loss.item<float>() << " visible " << visible.sum().item<int>() << std::endl;
        if (quiet && it == 2) { torch::cuda::synchronize(); t_start = std::chrono::steady_clock::now(); }
    }
    if (quiet && iters > 3) {   // timing mode (GSLIC_CHECK_TIME=1): views per second over the iterations after three warm-up ones
        torch::cuda::synchronize();
        const double sec = std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t_start).count();
        std::cout << "views_per_s " << (iters - 3) / sec << " ms_per_step " << 1e3 * sec / (iters - 3) << std::endl;
    }
#ifdef GSLIC_DIST
    if (rank != 0) { torch::cuda::synchronize(); std::cout.flush(); _exit(0); }   // replicas are identical: rank 0 reports
#endif
    save(d + "/out_image.f32", image);
    save(d + "/out_xyz.f32", xyz); save(d + "/out_scaling.f32", scaling); save(d + "/out_rotation.f32", rotation);
    save(d + "/out_opacity.f32", opacity); save(d + "/out_dc.f32", dc);
    if (M > 0) save(d + "/out_rest.f32", rest);
#ifdef GSLIC_DIST
    // leave without running static destructors: tearing down c10d's ProcessGroupNCCL / RCCL at process exit aborts intermittently
    // ("double free or corruption") after all work is done and saved — not something this check is about
    torch::cuda::synchronize();
    std::cout << "saved" << std::endl;
    _exit(0);
#endif
    return 0;
}
