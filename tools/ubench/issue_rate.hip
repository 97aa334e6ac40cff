// Micro-benchmark: issue cost of a wave64 VALU instruction on gfx950 as a function of its VGPR source-operand count / register banks.
// hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
#define REP8(S) S S S S S S S S
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b)
{
    // fixed registers so that the banks (register number mod 4) are known: x in v8..v15, operands in v16..v23
    for (int i = 0; i < N_ITER; i++) {
        if (MODE == 0)      asm volatile(REP8("v_mul_f32 v8, v8, s4\n v_mul_f32 v9, v9, s4\n v_mul_f32 v10, v10, s4\n v_mul_f32 v11, v11, s4\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 1) asm volatile(REP8("v_mul_f32 v8, v8, v16\n v_mul_f32 v9, v9, v17\n v_mul_f32 v10, v10, v18\n v_mul_f32 v11, v11, v19\n") ::: "v8", "v9", "v10", "v11");   // same bank
        else if (MODE == 2) asm volatile(REP8("v_mul_f32 v8, v8, v17\n v_mul_f32 v9, v9, v18\n v_mul_f32 v10, v10, v19\n v_mul_f32 v11, v11, v16\n") ::: "v8", "v9", "v10", "v11");   // other bank
        else if (MODE == 3) asm volatile(REP8("v_fma_f32 v8, v8, v17, v18\n v_fma_f32 v9, v9, v18, v19\n v_fma_f32 v10, v10, v19, v16\n v_fma_f32 v11, v11, v16, v17\n") ::: "v8", "v9", "v10", "v11");  // 3 banks
        else if (MODE == 4) asm volatile(REP8("v_fma_f32 v8, v8, v16, v20\n v_fma_f32 v9, v9, v17, v21\n v_fma_f32 v10, v10, v18, v22\n v_fma_f32 v11, v11, v19, v23\n") ::: "v8", "v9", "v10", "v11");  // all same bank
        else if (MODE == 5) asm volatile(REP8("v_fma_f32 v8, v8, s4, v17\n v_fma_f32 v9, v9, s4, v18\n v_fma_f32 v10, v10, s4, v19\n v_fma_f32 v11, v11, s4, v16\n") ::: "v8", "v9", "v10", "v11");      // 2 VGPR + SGPR
        else if (MODE == 6) asm volatile(REP8("v_mov_b32 v8, v17\n v_mov_b32 v9, v18\n v_mov_b32 v10, v19\n v_mov_b32 v11, v16\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 7) asm volatile(REP8("v_pk_fma_f32 v[8:9], v[8:9], v[16:17], v[18:19]\n v_pk_fma_f32 v[10:11], v[10:11], v[18:19], v[16:17]\n v_pk_fma_f32 v[12:13], v[12:13], v[16:17], v[18:19]\n v_pk_fma_f32 v[14:15], v[14:15], v[18:19], v[16:17]\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
        else if (MODE == 8) asm volatile(REP8("v_pk_mul_f32 v[8:9], v[8:9], v[18:19]\n v_pk_mul_f32 v[10:11], v[10:11], v[16:17]\n v_pk_mul_f32 v[12:13], v[12:13], v[18:19]\n v_pk_mul_f32 v[14:15], v[14:15], v[16:17]\n") ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
        else if (MODE == 9) asm volatile(REP8("v_exp_f32 v8, v8\n v_exp_f32 v9, v9\n v_exp_f32 v10, v10\n v_exp_f32 v11, v11\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 11) asm volatile(REP8("v_fma_f32 v8, v8, v16, v17\n v_fma_f32 v9, v9, v16, v17\n v_fma_f32 v10, v10, v16, v17\n v_fma_f32 v11, v11, v16, v17\n") ::: "v8", "v9", "v10", "v11");  // every instruction reads the SAME two multiplicand registers (what round 1's valu_rate did)
        else if (MODE == 12) asm volatile(REP8("v_fma_f32 v8, v8, v16, v17\n v_fma_f32 v9, v9, v18, v19\n v_fma_f32 v10, v10, v16, v17\n v_fma_f32 v11, v11, v18, v19\n") ::: "v8", "v9", "v10", "v11");  // alternating operand pairs
        else if (MODE == 10) asm volatile(REP8("v_mov_b32_dpp v8, v17 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v9, v18 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v10, v19 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v11, v16 wave_shr:1 row_mask:0xf bank_mask:0xf\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 13) asm volatile(REP8("v_mul_f32_dpp v8, v17, v8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp v9, v18, v9 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp v10, v19, v10 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp v11, v16, v11 row_shr:8 row_mask:0xf bank_mask:0xf\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 14) asm volatile(REP8("v_mov_b32_dpp v8, v17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v9, v18 row_newbcast:15 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v10, v19 row_newbcast:15 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v11, v16 row_newbcast:15 row_mask:0xf bank_mask:0xf\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 15) asm volatile(REP8("v_fma_mix_f32 v8, v17, v18, v8 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v9, v18, v19, v9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v10, v19, v16, v10 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v11, v16, v17, v11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 16) asm volatile(REP8("v_permlane32_swap_b32 v8, v9\n v_permlane16_swap_b32 v10, v11\n v_permlane32_swap_b32 v9, v10\n v_permlane16_swap_b32 v11, v8\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 17) asm volatile(REP8("v_rcp_f32 v8, v8\n v_rcp_f32 v9, v9\n v_rcp_f32 v10, v10\n v_rcp_f32 v11, v11\n") ::: "v8", "v9", "v10", "v11");
        else if (MODE == 18) asm volatile(REP8("s_nop 1\n v_mul_f32_dpp v8, v8, v8 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mul_f32_dpp v8, v8, v8 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mul_f32_dpp v8, v8, v8 row_shr:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mul_f32_dpp v8, v8, v8 row_shr:8 row_mask:0xf bank_mask:0xf\n") ::: "v8");   // a 16-lane scan: each step depends on the last
        else if (MODE == 19) asm volatile(REP8("v_bfe_i32 v8, v17, v18, 1\n v_bfi_b32 v9, v18, v19, v16\n v_and_b32 v10, v19, v16\n v_med3_f32 v11, v16, v17, v18\n") ::: "v8", "v9", "v10", "v11");
    }
    float r;
    asm volatile("v_add_f32 %0, v8, v9" : "=v"(r));
    out[blockIdx.x * 256 + threadIdx.x] = r + a + b;
}
template <int MODE> void run(const char* name, float* d, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * N_ITER * 32 / 1024.0;
    printf("%-44s %d waves/SIMD  %.3f ms -> %.2f cycles / instruction / SIMD at 2.4 GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main()
{
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {8, 2, 1}) {
        run<0>("v_mul_f32 v, v, s (1 VGPR source)", d, w);
        run<1>("v_mul_f32 v, v, v (sources in one bank)", d, w);
        run<2>("v_mul_f32 v, v, v (two banks)", d, w);
        run<3>("v_fma_f32 v, v, v, v (three banks)", d, w);
        run<4>("v_fma_f32 v, v, v, v (one bank)", d, w);
        run<5>("v_fma_f32 v, v, s, v", d, w);
        run<6>("v_mov_b32 v, v", d, w);
        run<7>("v_pk_fma_f32", d, w);
        run<8>("v_pk_mul_f32", d, w);
        run<9>("v_exp_f32", d, w);
        run<10>("v_mov_b32_dpp wave_shr:1", d, w);
        run<11>("v_fma_f32, same two source registers in every instruction", d, w);
        run<12>("v_fma_f32, two alternating source pairs", d, w);
        run<13>("v_mul_f32_dpp row_shr:1/2/4/8", d, w);
        run<14>("v_mov_b32_dpp row_newbcast:15", d, w);
        run<15>("v_fma_mix_f32 (f16 lo / hi source)", d, w);
        run<16>("v_permlane32_swap / v_permlane16_swap", d, w);
        run<17>("v_rcp_f32", d, w);
        run<18>("dependent v_mul_f32_dpp row_shr chain + s_nop 1", d, w);
        run<19>("v_bfe_i32 / v_bfi_b32 / v_and_b32 / v_med3_f32", d, w);
    }
    return 0;
}
