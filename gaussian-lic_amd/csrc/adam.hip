// adam.hip — visibility-masked Adam without bias correction.
// Replaces adamUpdateCUDA (cuda_rasterizer/adam.cu:9-38): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr m / (sqrt(v) + eps), only where visible[row]; rows whose visibility byte is 0 keep p, m, v bit for bit.
//
// HBM-bound design: the tensors are walked as flat float4 arrays (16 B per lane per array, seven streams), the row of
// each scalar comes from a compile-time row width M (multiply-shift, no integer division in the loop; the reference
// divides per scalar), a float4 whose four rows are all invisible issues no memory traffic at all, and the six parameter
// groups of gaussian.cpp:399-418 run in ONE launch (blockIdx.y = group) — no per-group launch, no grad.clone()
// (optim_utils.h:130).
#include "gslic_common.h"

namespace gslic {

struct AdamGroups {
    gslic_adam_group g[8];
    int n;
};

// rows are M scalars wide; element index e -> row e / M with M a compile-time constant (0 = runtime fallback)
template <uint32_t M>
__device__ __forceinline__ uint32_t row_of(uint32_t e, uint32_t m_rt)
{
    if constexpr (M == 0) return e / m_rt;
    else return e / M;
}

template <uint32_t M>
__device__ __forceinline__ void adam_span(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ ea,
                                          float* __restrict__ es, const uint8_t* __restrict__ visible, float lr, float b1, float b2,
                                          float eps, uint32_t total, uint32_t m_rt, uint32_t first, uint32_t stride)
{
    const uint32_t nvec = total >> 2;
    const bool aligned = (((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(ea) |
                            reinterpret_cast<uintptr_t>(es)) & 15) == 0);
    if (aligned) {
        for (uint32_t q = first; q < nvec; q += stride) {
            const uint32_t e = q << 2;
            const uint32_t r0 = row_of<M>(e, m_rt), r3 = row_of<M>(e + 3, m_rt);
            bool vis[4];
            if (r0 == r3) {
                vis[0] = vis[1] = vis[2] = vis[3] = visible[r0] != 0;
            } else {
                vis[0] = visible[r0] != 0;
                vis[1] = visible[row_of<M>(e + 1, m_rt)] != 0;
                vis[2] = visible[row_of<M>(e + 2, m_rt)] != 0;
                vis[3] = visible[r3] != 0;
            }
            if (!(vis[0] | vis[1] | vis[2] | vis[3])) continue;
            float4 p = reinterpret_cast<float4*>(param)[q];
            const float4 g = reinterpret_cast<const float4*>(grad)[q];
            float4 m = reinterpret_cast<float4*>(ea)[q];
            float4 v = reinterpret_cast<float4*>(es)[q];
            if (vis[0]) adam_scalar(p.x, g.x, m.x, v.x, lr, b1, b2, eps);
            if (vis[1]) adam_scalar(p.y, g.y, m.y, v.y, lr, b1, b2, eps);
            if (vis[2]) adam_scalar(p.z, g.z, m.z, v.z, lr, b1, b2, eps);
            if (vis[3]) adam_scalar(p.w, g.w, m.w, v.w, lr, b1, b2, eps);
            reinterpret_cast<float4*>(param)[q] = p;
            reinterpret_cast<float4*>(ea)[q] = m;
            reinterpret_cast<float4*>(es)[q] = v;
        }
    }
    // tail (or everything, for unaligned views): scalar path
    for (uint32_t e = (aligned ? (nvec << 2) : 0u) + first; e < total; e += stride) {
        if (visible[row_of<M>(e, m_rt)]) adam_scalar(param[e], grad[e], ea[e], es[e], lr, b1, b2, eps);
    }
}

__device__ __forceinline__ void adam_dispatch(const gslic_adam_group& grp, const uint8_t* visible, float b1, float b2, float eps,
                                              uint32_t N, uint32_t first, uint32_t stride)
{
    const uint32_t total = N * grp.M;
    switch (grp.M) {
        case 1: adam_span<1>(grp.param, grp.grad, grp.exp_avg, grp.exp_avg_sq, visible, grp.lr, b1, b2, eps, total, 1, first, stride); break;
        case 3: adam_span<3>(grp.param, grp.grad, grp.exp_avg, grp.exp_avg_sq, visible, grp.lr, b1, b2, eps, total, 3, first, stride); break;
        case 4: adam_span<4>(grp.param, grp.grad, grp.exp_avg, grp.exp_avg_sq, visible, grp.lr, b1, b2, eps, total, 4, first, stride); break;
        case 45: adam_span<45>(grp.param, grp.grad, grp.exp_avg, grp.exp_avg_sq, visible, grp.lr, b1, b2, eps, total, 45, first, stride); break;
        default: adam_span<0>(grp.param, grp.grad, grp.exp_avg, grp.exp_avg_sq, visible, grp.lr, b1, b2, eps, total, grp.M, first, stride); break;
    }
}

__global__ __launch_bounds__(256) void adam_groups_kernel(AdamGroups gs, const uint8_t* __restrict__ visible, float b1, float b2,
                                                          float eps, uint32_t N)
{
    adam_dispatch(gs.g[blockIdx.y], visible, b1, b2, eps, N, blockIdx.x * 256u + threadIdx.x, gridDim.x * 256u);
}

int adam_update_groups(const gslic_adam_group* groups, int n, const uint8_t* visible, float b1, float b2, float eps, uint32_t N,
                       hipStream_t s)
{
    if (n <= 0 || N == 0) return GSLIC_OK;
    for (int i = 0; i < n; i++)
        if ((uint64_t)N * groups[i].M >= (1ull << 32)) return set_error(GSLIC_ERR_INVALID_ARG, "adam: N*M must be < 2^32");
    for (int base = 0; base < n; base += 8) {
        AdamGroups gs;
        gs.n = (n - base) < 8 ? (n - base) : 8;
        uint32_t maxM = 1;
        for (int i = 0; i < gs.n; i++) {
            gs.g[i] = groups[base + i];
            if (gs.g[i].M > maxM) maxM = gs.g[i].M;
        }
        size_t blocks = div_up_sz(div_up_sz((size_t)N * maxM, 4), 256);
        if (blocks > 8192) blocks = 8192;  // grid-stride beyond 8 blocks of 256 per CU
        if (blocks == 0) blocks = 1;
        GS_LAUNCH(K_ADAM, adam_groups_kernel, dim3((unsigned)blocks, (unsigned)gs.n), dim3(256), 0, s, gs, visible, b1, b2, eps, N);
    }
    return GSLIC_OK;
}

int adam_update(float* param, const float* grad, float* m, float* v, const uint8_t* visible, float lr, float b1, float b2,
                float eps, uint32_t N, uint32_t M, hipStream_t s)
{
    gslic_adam_group g;
    g.param = param; g.grad = grad; g.exp_avg = m; g.exp_avg_sq = v; g.lr = lr; g.M = M;
    return adam_update_groups(&g, 1, visible, b1, b2, eps, N, s);
}

}  // namespace gslic
