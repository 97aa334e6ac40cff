// Checks the DPP idioms render_bwd_scan_kernel relies on against plain reference values, on the device:
//   row_shr:1/2/4/8 in-place product scan, v_add_f32_dpp bound_ctrl sum scan, row_newbcast:15, the 64-lane OR through row_bcast15 / row_bcast31.
// hipcc --offload-arch=gfx950 -O3 -o dpp_check dpp_check.hip && ./dpp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float src) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false)); }
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_u(unsigned old, unsigned src) { return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false); }
__global__ void k(const float* in, const unsigned* uin, float* out, unsigned* uout)
{
    const int l = threadIdx.x;
    float x = in[l];
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    out[l] = x;
    float y = in[l];
    y += dpp_f<0x111>(0.0f, y); y += dpp_f<0x112>(0.0f, y); y += dpp_f<0x114>(0.0f, y); y += dpp_f<0x118>(0.0f, y);
    out[64 + l] = y;
    out[128 + l] = dpp_f<0x15f>(in[l], in[l]);
    unsigned v = uin[l];
    v |= dpp_u<0x111>(0u, v); v |= dpp_u<0x112>(0u, v); v |= dpp_u<0x114>(0u, v); v |= dpp_u<0x118>(0u, v);
    v |= dpp_u<0x142, 0xa>(0u, v);
    v |= dpp_u<0x143, 0xc>(0u, v);
    uout[l] = v;
}
int main()
{
    float h[64], *d, *o, r[192]; unsigned hu[64], *du, *duo, ru[64];
    for (int i = 0; i < 64; i++) { h[i] = 0.5f + 0.01f * i; hu[i] = 1u << (i % 32) | (i >= 32 ? 0x80000000u >> (i % 7) : 0u); }
    (void)hipMalloc(&d, 256); (void)hipMalloc(&o, 768); (void)hipMalloc(&du, 256); (void)hipMalloc(&duo, 256);
    (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice); (void)hipMemcpy(du, hu, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, du, o, duo);
    (void)hipMemcpy(r, o, 768, hipMemcpyDeviceToHost); (void)hipMemcpy(ru, duo, 256, hipMemcpyDeviceToHost);
    int bad_mul = 0, bad_add = 0, bad_bc = 0;
    for (int i = 0; i < 64; i++) {
        double p = 1, s = 0;
        for (int j = i & ~15; j <= i; j++) { p *= h[j]; s += h[j]; }
        if (fabs(r[i] - p) > 1e-5 * fabs(p)) bad_mul++;
        if (fabs(r[64 + i] - s) > 1e-5 * fabs(s)) bad_add++;
        if (r[128 + i] != h[i | 15]) bad_bc++;
    }
    unsigned all = 0; for (int i = 0; i < 64; i++) all |= hu[i];
    printf("row product scan: %d lanes wrong; row sum scan: %d; row_newbcast:15: %d; wave OR at lane 63: %08x expected %08x\n", bad_mul, bad_add, bad_bc, ru[63], all);
    if (bad_mul) { printf("mul:"); for (int i = 0; i < 20; i++) printf(" %.4g", r[i]); printf("\n"); }
    if (bad_bc) { printf("bcast:"); for (int i = 0; i < 20; i++) printf(" %.4g", r[128 + i]); printf("\n"); }
    return 0;
}
