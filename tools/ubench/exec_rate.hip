// Micro-benchmark: does a wave64 VALU instruction get cheaper when part of EXEC is zero?  (cycles per v_fma_f32 per SIMD for several masks)
// hipcc --offload-arch=gfx950 -O3 -o exec_rate exec_rate.hip && ./exec_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
__global__ __launch_bounds__(256) void k(float* out, float a, float b, unsigned long long mask)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < N_ITER; i++) {
        asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %10\n"
                     "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     "s_mov_b64 exec, s[20:21]\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(mask) : "s20", "s21");
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int blocks = 256 * 8;
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0x1ull, 0xffff0000ffff0000ull, 0x1000100010001ull, 0xffffffff00000000ull};
    for (unsigned long long m : masks) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, m);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, m);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)blocks * 4 * N_ITER * 8 / 1024.0;
        printf("exec %016llx  %.3f ms  -> %.2f cycles / v_fma_f32 / SIMD at 2.4 GHz\n", m, ms, ms * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
