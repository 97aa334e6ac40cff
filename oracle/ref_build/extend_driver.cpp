// extend_driver.cpp — runs the REFERENCE's own extend() (src/gaussian.cpp:499-638) on the CPU, against stand-ins for the types around it.
//
// Test infrastructure (never shipped).  The function body is NOT in this repository: make_extend_golden.py cuts the text between
// "void extend(" and "double optimize(" out of /root/reference/src/gaussian.cpp at build time into oracle/_ref/extend_extract.inc
// (git-ignored) and this file #includes it — the per-pixel `unordered_map<string>` z-buffer (:549-572), the filter lambda (:584-603)
// and the new-Gaussian rows (:605-626) that produce tests/golden/extend_*.npz are the reference's statements, compiled unmodified.
// What is supplied here: CPU LibTorch (every `.cuda()` of the extract becomes `.cpu()` through a macro), minimal Dataset / Camera /
// GaussianModel types holding exactly the members the function touches, a 3x3 double matrix with the four Eigen operations it uses,
// render() returning the transmittance image the caller passes in, and RGB2SH (gaussian.h:46-48: one line).
// general_utils.h (inverse_sigmoid) is the reference's header, read in place.
//
//   extend_driver <dir>      reads  <dir>/meta.f64 (n W H fx fy cx cy sh_degree scaling_scale R_wc[9] t_wc[3]),
//                                   points.f32 [n,3], colors.f32 [n,3], depths_rsp.f32 [n], final_T.f32 [H,W]
//                            writes <dir>/out_count.i64 and out_{xyz,dc,rest,opacity,scaling,rotation}.f32 (the densificationPostfix arguments)
#include <torch/torch.h>

#include <array>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "general_utils.h"   // /root/reference/src (include path)

// ---- the four Eigen operations extend() uses on R_wc (Matrix3d) / t_wc (Vector3d): transpose(), unary minus, matrix * vector, element access
struct V3 {
    double v[3];
    double operator[](int i) const { return v[i]; }
};
struct M3 {
    double a[3][3];
    M3 transpose() const { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[i][j] = a[j][i]; return r; }
    double operator()(int i, int j) const { return a[i][j]; }
};
static M3 operator-(const M3& m) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[i][j] = -m.a[i][j]; return r; }
static V3 operator*(const M3& m, const V3& x) { V3 r; for (int i = 0; i < 3; i++) r.v[i] = m.a[i][0] * x.v[0] + m.a[i][1] * x.v[1] + m.a[i][2] * x.v[2]; return r; }

struct Camera {
    double fx_, fy_, cx_, cy_;
    int image_height_, image_width_;
};
struct Dataset {
    std::vector<std::shared_ptr<Camera>> train_cameras_;
    std::vector<M3> R_wc_;
    std::vector<V3> t_wc_;
    std::vector<std::array<double, 3>> pointcloud_, pointcolor_;
    std::vector<float> pointdepth_;
};
struct GaussianModel {
    int sh_degree_ = 3;
    bool white_background_ = false, apply_exposure_ = false;
    double scaling_scale_ = 1.0;
    std::vector<torch::Tensor> appended;   // what the function hands to densificationPostfix
    void densificationPostfix(torch::Tensor& xyz, torch::Tensor& dc, torch::Tensor& rest, torch::Tensor& opacity, torch::Tensor& scaling, torch::Tensor& rotation)
    {
        appended = {xyz, dc, rest, opacity, scaling, rotation};
    }
};
static torch::Tensor g_final_T;   // [1, H, W]: the no_color render of the caller's map
static std::tuple<torch::Tensor, torch::Tensor> render(const std::shared_ptr<Camera>&, const std::shared_ptr<GaussianModel>&, const torch::Tensor&, bool, bool)
{
    return std::make_tuple(torch::Tensor(), g_final_T);
}
static const double C0 = 0.28209479177387814;
static inline torch::Tensor RGB2SH(torch::Tensor& rgb) { return (rgb - 0.5f) / C0; }

#define cuda cpu
#include "extend_extract.inc"   // GENERATED from the reference at build time: void extend(const std::shared_ptr<Dataset>&, std::shared_ptr<GaussianModel>&)
#undef cuda

template <typename T>
static std::vector<T> slurp(const std::string& path, size_t n)
{
    std::vector<T> v(n);
    std::ifstream f(path, std::ios::binary);
    if (n && !f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)))) { std::cerr << "short read: " << path << "\n"; std::exit(2); }
    return v;
}
static void dump(const std::string& path, const torch::Tensor& t)
{
    auto c = t.contiguous().to(torch::kFloat32);
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(c.data_ptr<float>()), (std::streamsize)(c.numel() * sizeof(float)));
}

int main(int argc, char** argv)
{
    if (argc != 2) { std::cerr << "usage: extend_driver <dir>\n"; return 1; }
    const std::string d = std::string(argv[1]) + "/";
    auto meta = slurp<double>(d + "meta.f64", 21);
    const size_t n = (size_t)meta[0];
    const int W = (int)meta[1], H = (int)meta[2];
    auto cam = std::make_shared<Camera>();
    cam->fx_ = meta[3]; cam->fy_ = meta[4]; cam->cx_ = meta[5]; cam->cy_ = meta[6]; cam->image_width_ = W; cam->image_height_ = H;
    auto pc = std::make_shared<GaussianModel>();
    pc->sh_degree_ = (int)meta[7]; pc->scaling_scale_ = meta[8];
    auto ds = std::make_shared<Dataset>();
    ds->train_cameras_.push_back(cam);
    M3 R; V3 t;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R.a[i][j] = meta[9 + 3 * i + j]; t.v[i] = meta[18 + i]; }
    ds->R_wc_.push_back(R); ds->t_wc_.push_back(t);
    auto pts = slurp<float>(d + "points.f32", 3 * n), col = slurp<float>(d + "colors.f32", 3 * n);
    ds->pointdepth_ = slurp<float>(d + "depths_rsp.f32", n);
    for (size_t i = 0; i < n; i++) {
        ds->pointcloud_.push_back({pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]});
        ds->pointcolor_.push_back({col[3 * i], col[3 * i + 1], col[3 * i + 2]});
    }
    auto T = slurp<float>(d + "final_T.f32", (size_t)W * H);
    g_final_T = torch::from_blob(T.data(), {1, H, W}, torch::kFloat32).clone();
    extend(ds, pc);
    std::cout << "\n";
    const int64_t k = pc->appended.empty() ? 0 : pc->appended[0].size(0);
    { std::ofstream f(d + "out_count.i64", std::ios::binary); f.write(reinterpret_cast<const char*>(&k), sizeof(k)); }
    const char* names[6] = {"xyz", "dc", "rest", "opacity", "scaling", "rotation"};
    for (int i = 0; i < 6 && !pc->appended.empty(); i++) dump(d + "out_" + names[i] + ".f32", pc->appended[i]);
    return 0;
}
