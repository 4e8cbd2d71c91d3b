// Micro-benchmark: cycles per wave-instruction per SIMD for the VALU ops the blend kernels use (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters) {                         \
        float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f,        \
              a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = 0.999f, c = 1e-4f;  \
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc" :: "v"(a0), "v"(a1) : "vcc", "s20", "s21"); \
        for (int i = 0; i < iters; ++i) {                                                        \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                         "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");               \
        }                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;            \
    }
// 8 independent instructions per ASM string (one per accumulator)
#define OP8(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n"
#define OP8_2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                  op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define OP8_1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
BODY(k_fma, OP8("v_fma_f32"))
BODY(k_mul, OP8_2("v_mul_f32"))
BODY(k_exp, OP8_1("v_exp_f32"))
BODY(k_rcp, OP8_1("v_rcp_f32"))
BODY(k_sqrt, OP8_1("v_sqrt_f32"))
BODY(k_cmp_sgpr, "v_cmp_le_f32 s[20:21], %8, %0\n v_cmp_le_f32 s[20:21], %8, %1\n v_cmp_le_f32 s[20:21], %8, %2\n v_cmp_le_f32 s[20:21], %8, %3\n v_cmp_le_f32 s[20:21], %8, %4\n v_cmp_le_f32 s[20:21], %8, %5\n v_cmp_le_f32 s[20:21], %8, %6\n v_cmp_le_f32 s[20:21], %8, %7\n")
BODY(k_cmp_vcc, "v_cmp_le_f32 vcc, %8, %0\n v_cmp_le_f32 vcc, %8, %1\n v_cmp_le_f32 vcc, %8, %2\n v_cmp_le_f32 vcc, %8, %3\n v_cmp_le_f32 vcc, %8, %4\n v_cmp_le_f32 vcc, %8, %5\n v_cmp_le_f32 vcc, %8, %6\n v_cmp_le_f32 vcc, %8, %7\n")
BODY(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
BODY(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
BODY(k_med3, OP8("v_med3_f32"))
BODY(k_min, OP8_2("v_min_f32"))
BODY(k_mov, OP8_1("v_mov_b32"))
BODY(k_dpp_add, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
BODY(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n")
__global__ __launch_bounds__(256) void k_pk_fma(float *out, int iters) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f a0 = {threadIdx.x * 1e-3f + 1.f, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
        a6 = a0 + 6.f, a7 = a0 + 7.f, b = {0.999f, 0.998f}, c = {1e-4f, 2e-4f};
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
template <typename K> void run(const char *name, K kern, float *d, int blocks_per_cu) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<grid, 256>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(a); kern<<<grid, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: each CU runs blocks_per_cu*4 waves over 4 SIMDs -> blocks_per_cu waves per SIMD
    double insts_per_simd = (double)blocks_per_cu * iters * 64.0;
    printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name,
           blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int bpc : {4}) {
        run("v_fma_f32", k_fma, d, bpc); run("v_mul_f32", k_mul, d, bpc); run("v_pk_fma_f32", k_pk_fma, d, bpc);
        run("v_exp_f32", k_exp, d, bpc); run("v_rcp_f32", k_rcp, d, bpc); run("v_sqrt_f32", k_sqrt, d, bpc);
        run("v_cmp->sgpr", k_cmp_sgpr, d, bpc); run("v_cmp->vcc", k_cmp_vcc, d, bpc); run("v_cndmask", k_cndmask, d, bpc); run("v_cndmask_sgpr", k_cndmask_s, d, bpc); run("v_med3", k_med3, d, bpc); run("v_min", k_min, d, bpc); run("v_mov", k_mov, d, bpc);
        run("v_add_dpp", k_dpp_add, d, bpc); run("permlane32swap", k_swap32, d, bpc);
    }
    return 0;
}
