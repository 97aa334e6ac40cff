// adam.hip — visibility-masked Adam without bias correction.
// adam_kernel replaces adamUpdateCUDA (cuda_rasterizer/adam.cu:9-38).  One thread per scalar; rows whose
// visibility byte is 0 are not read or written at all (exp_avg / exp_avg_sq stay untouched).
// adam_groups_kernel runs the six parameter groups of gaussian.cpp:399-418 in ONE launch (blockIdx.y = group).
#include "gslic_common.h"

namespace gslic {

__device__ __forceinline__ void adam_one(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                                         float* __restrict__ v, size_t p, float lr, float b1, float b2, float eps)
{
    const float g = grad[p];
    const float m1 = b1 * m[p] + (1.0f - b1) * g;
    const float v1 = b2 * v[p] + (1.0f - b2) * g * g;
    const float step = -lr * m1 / (sqrtf(v1) + eps);
    param[p] += step;
    m[p] = m1;
    v[p] = v1;
}

__global__ __launch_bounds__(256) void adam_kernel(float* param, const float* grad, float* m, float* v,
                                                   const uint8_t* __restrict__ visible, float lr, float b1, float b2, float eps,
                                                   uint32_t N, uint32_t M)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t g = p / M;
    if (g >= N) return;
    if (visible[g]) adam_one(param, grad, m, v, p, lr, b1, b2, eps);
}

struct AdamGroups {
    gslic_adam_group g[8];
    int n;
};

__global__ __launch_bounds__(256) void adam_groups_kernel(AdamGroups gs, const uint8_t* __restrict__ visible, float b1, float b2,
                                                          float eps, uint32_t N)
{
    const gslic_adam_group grp = gs.g[blockIdx.y];
    const size_t total = (size_t)N * grp.M;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const size_t g = p / grp.M;
        if (visible[g]) adam_one(grp.param, grp.grad, grp.exp_avg, grp.exp_avg_sq, p, grp.lr, b1, b2, eps);
    }
}

int adam_update(float* param, const float* grad, float* m, float* v, const uint8_t* visible, float lr, float b1, float b2,
                float eps, uint32_t N, uint32_t M, hipStream_t s)
{
    const size_t cnt = (size_t)N * M;
    if (cnt == 0) return GSLIC_OK;
    GS_LAUNCH(K_ADAM, adam_kernel, dim3((unsigned)div_up_sz(cnt, 256)), dim3(256), 0, s, param, grad, m, v, visible, lr, b1, b2,
              eps, N, M);
    return GSLIC_OK;
}

int adam_update_groups(const gslic_adam_group* groups, int n, const uint8_t* visible, float b1, float b2, float eps, uint32_t N,
                       hipStream_t s)
{
    if (n <= 0 || N == 0) return GSLIC_OK;
    for (int base = 0; base < n; base += 8) {
        AdamGroups gs;
        gs.n = (n - base) < 8 ? (n - base) : 8;
        uint32_t maxM = 1;
        for (int i = 0; i < gs.n; i++) {
            gs.g[i] = groups[base + i];
            if (gs.g[i].M > maxM) maxM = gs.g[i].M;
        }
        size_t blocks = div_up_sz((size_t)N * maxM, 256);
        if (blocks > 65535u * 16u) blocks = 65535u * 16u;
        GS_LAUNCH(K_ADAM, adam_groups_kernel, dim3((unsigned)blocks, (unsigned)gs.n), dim3(256), 0, s, gs, visible, b1, b2, eps, N);
    }
    return GSLIC_OK;
}

}  // namespace gslic
