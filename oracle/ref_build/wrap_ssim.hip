// wrap_ssim.hip — the reference's fused-SSIM kernels (src/fused-ssim/ssim.cu, read in place, unmodified) behind C entry points.
// Checker only (golden vectors on the MI355X, direct comparison in tests/test_vs_reference_kernels_gpu.py); never shipped.
// All pointer arguments are HOST arrays.
#include "/root/reference/src/fused-ssim/ssim.cu"
#include <vector>

#define SCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "ref_ssim: %s -> %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

namespace {
struct DevBuf {
    float* p = nullptr;
    DevBuf(const float* host, size_t n, bool zero = false)
    {
        if (hipMalloc((void**)&p, n * sizeof(float) + 256) != hipSuccess) { p = nullptr; return; }
        if (host) hipMemcpy(p, host, n * sizeof(float), hipMemcpyHostToDevice);
        else if (zero) hipMemset(p, 0, n * sizeof(float));
    }
    ~DevBuf() { if (p) hipFree(p); }
};
}  // namespace

extern "C" {

// fusedssim (ssim.cu:367-402) with train = true
int ref_ssim_forward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, float* ssim_map, float* dm_dmu1,
                     float* dm_dsigma1_sq, float* dm_dsigma12)
{
    const size_t n = (size_t)B * CH * H * W;
    DevBuf a(img1, n), b(img2, n), m(nullptr, n, true), d1(nullptr, n, true), d2(nullptr, n, true), d3(nullptr, n, true);
    if (!a.p || !b.p || !m.p || !d1.p || !d2.p || !d3.p) return -1;
    dim3 grid((W + BX - 1) / BX, (H + BY - 1) / BY, B);
    dim3 block(BX, BY, 1);
    fusedssimCUDA<<<grid, block>>>(H, W, CH, C1, C2, a.p, b.p, m.p, d1.p, d2.p, d3.p);
    SCHK(hipGetLastError());
    SCHK(hipDeviceSynchronize());
    SCHK(hipMemcpy(ssim_map, m.p, n * sizeof(float), hipMemcpyDeviceToHost));
    SCHK(hipMemcpy(dm_dmu1, d1.p, n * sizeof(float), hipMemcpyDeviceToHost));
    SCHK(hipMemcpy(dm_dsigma1_sq, d2.p, n * sizeof(float), hipMemcpyDeviceToHost));
    SCHK(hipMemcpy(dm_dsigma12, d3.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

// fusedssim_backward (ssim.cu:404-441)
int ref_ssim_backward(int B, int CH, int H, int W, float C1, float C2, const float* img1, const float* img2, const float* dL_dmap,
                      const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1)
{
    const size_t n = (size_t)B * CH * H * W;
    DevBuf a(img1, n), b(img2, n), g(dL_dmap, n), d1(dm_dmu1, n), d2(dm_dsigma1_sq, n), d3(dm_dsigma12, n), o(nullptr, n, true);
    if (!a.p || !b.p || !g.p || !d1.p || !d2.p || !d3.p || !o.p) return -1;
    dim3 grid((W + BX - 1) / BX, (H + BY - 1) / BY, B);
    dim3 block(BX, BY, 1);
    fusedssim_backwardCUDA<<<grid, block>>>(H, W, CH, C1, C2, a.p, b.p, g.p, o.p, d1.p, d2.p, d3.p);
    SCHK(hipGetLastError());
    SCHK(hipDeviceSynchronize());
    SCHK(hipMemcpy(dL_dimg1, o.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
