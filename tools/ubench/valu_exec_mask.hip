// Micro-benchmark (gfx950): what a wave-instruction costs its SIMD as a function of the EXEC mask and of the instruction
// class -- does the SIMD-32 skip a half-wave pass whose 32 lanes are all masked off?  What do DPP / permlane / packed /
// transcendental instructions cost relative to a plain fp32 one?  Shader-clock cycles from s_memtime, one wave per SIMD
// (issue cost) and four waves per SIMD (throughput).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_exec_mask.hip -o /tmp/valu_exec_mask && /tmp/valu_exec_mask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// eight independent accumulators a0..a7 (operands %0..%7), b = %12, c = %13; EXEC is set from the kernel argument
// (ONE asm statement from the EXEC write to its restore: nothing the compiler schedules -- the address arithmetic of the
// final stores, say -- can land inside the masked region)
#define TIMED_LOOP(BODY)                                                                                                  \
    "s_mov_b64 %[saved], exec\n s_mov_b64 exec, %[mask]\n s_mov_b32 %[ctr], %[iters]\n"                                   \
    "s_waitcnt lgkmcnt(0)\n s_memtime %[t0]\n s_waitcnt lgkmcnt(0)\n"                                                     \
    "1:\n" REP16(BODY) "s_sub_u32 %[ctr], %[ctr], 1\n s_cmp_lg_u32 %[ctr], 0\n s_cbranch_scc1 1b\n"                        \
    "s_waitcnt lgkmcnt(0)\n s_memtime %[t1]\n s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, %[saved]\n"
#define KERNEL(NAME, BODY8)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *cycles, int iters, unsigned long long mask) { \
        float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,   \
              a6 = a0 + 6.f, a7 = a0 + 7.f, b = 0.999f, c = 1e-4f;                                                        \
        unsigned long long t0, t1, saved;                                                                                 \
        int ctr;                                                                                                          \
        asm volatile(TIMED_LOOP(BODY8)                                                                                    \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [t0] "=&s"(t0),     \
                       [t1] "=&s"(t1), [saved] "=&s"(saved), [ctr] "=&s"(ctr)                                              \
                     : "v"(b), "v"(c), [mask] "s"(mask), [iters] "s"(iters)                                                \
                     : "vcc", "scc");                                                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                      \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                               \
    }

#define OP3(op) op " %0, %0, %12, %13\n" op " %1, %1, %12, %13\n" op " %2, %2, %12, %13\n" op " %3, %3, %12, %13\n" \
                op " %4, %4, %12, %13\n" op " %5, %5, %12, %13\n" op " %6, %6, %12, %13\n" op " %7, %7, %12, %13\n"
#define OP2(op) op " %0, %0, %12\n" op " %1, %1, %12\n" op " %2, %2, %12\n" op " %3, %3, %12\n" \
                op " %4, %4, %12\n" op " %5, %5, %12\n" op " %6, %6, %12\n" op " %7, %7, %12\n"
#define OP1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define DPP8(ctrl) "v_add_f32_dpp %0, %0, %0 " ctrl "\n v_add_f32_dpp %1, %1, %1 " ctrl "\n v_add_f32_dpp %2, %2, %2 " ctrl "\n" \
                   "v_add_f32_dpp %3, %3, %3 " ctrl "\n v_add_f32_dpp %4, %4, %4 " ctrl "\n v_add_f32_dpp %5, %5, %5 " ctrl "\n" \
                   "v_add_f32_dpp %6, %6, %6 " ctrl "\n v_add_f32_dpp %7, %7, %7 " ctrl "\n"

KERNEL(k_fma, OP3("v_fma_f32"))
KERNEL(k_mul, OP2("v_mul_f32"))
KERNEL(k_add, OP2("v_add_f32"))
KERNEL(k_exp, OP1("v_exp_f32"))
KERNEL(k_rcp, OP1("v_rcp_f32"))
KERNEL(k_sqrt, OP1("v_sqrt_f32"))
KERNEL(k_med3, OP3("v_med3_f32"))
KERNEL(k_cmp_vcc, "v_cmp_le_f32 vcc, %12, %0\n v_cmp_le_f32 vcc, %12, %1\n v_cmp_le_f32 vcc, %12, %2\n v_cmp_le_f32 vcc, %12, %3\n"
                  "v_cmp_le_f32 vcc, %12, %4\n v_cmp_le_f32 vcc, %12, %5\n v_cmp_le_f32 vcc, %12, %6\n v_cmp_le_f32 vcc, %12, %7\n")
KERNEL(k_dpp_shr1, DPP8("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"))
KERNEL(k_dpp_ror8, DPP8("row_ror:8 row_mask:0xf bank_mask:0xf"))
KERNEL(k_dpp_quad, DPP8("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                 "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n")
KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                 "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n")

// MIXTURES: do special instructions (transcendental, permlane swap, DPP, packed) overlap with plain ones of OTHER waves on the
// same SIMD, or do they simply add up?  Body = 8 v_fma_f32 + 2 specials (10 instructions).
#define FMA8 OP3("v_fma_f32")
KERNEL(k_mix_exp, FMA8 "v_exp_f32 %0, %0\n v_exp_f32 %4, %4\n")
KERNEL(k_mix_swap, FMA8 "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %4, %5\n")
KERNEL(k_mix_dpp, FMA8 "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
KERNEL(k_mix_cmp, FMA8 "v_cmp_le_f32 vcc, %12, %0\n v_cmp_le_f32 vcc, %12, %4\n")
KERNEL(k_mix_med3, FMA8 "v_med3_f32 %0, %0, %12, %13\n v_med3_f32 %4, %4, %12, %13\n")
KERNEL(k_mix_exp_swap, "v_exp_f32 %0, %0\n v_permlane32_swap_b32 %2, %3\n v_exp_f32 %4, %4\n v_permlane32_swap_b32 %6, %7\n v_exp_f32 %1, %1\n v_permlane32_swap_b32 %2, %3\n v_exp_f32 %5, %5\n v_permlane32_swap_b32 %6, %7\n")
KERNEL(k_mix_exp_dpp, "v_exp_f32 %0, %0\n v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_exp_f32 %4, %4\n v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_exp_f32 %1, %1\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_exp_f32 %5, %5\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
// packed fp32: operands are register pairs
#define KERNEL_PK(NAME, OP)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *cycles, int iters, unsigned long long mask) { \
        typedef float v2f __attribute__((ext_vector_type(2)));                                                            \
        v2f a0 = {threadIdx.x * 1e-3f + 1.f, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,             \
            a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = {0.999f, 0.998f}, c = {1e-4f, 2e-4f};                         \
        unsigned long long t0, t1, saved;                                                                                 \
        int ctr;                                                                                                          \
        asm volatile(TIMED_LOOP(OP)                                                                                       \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [t0] "=&s"(t0),     \
                       [t1] "=&s"(t1), [saved] "=&s"(saved), [ctr] "=&s"(ctr)                                              \
                     : "v"(b), "v"(c), [mask] "s"(mask), [iters] "s"(iters)                                                \
                     : "scc");                                                                                            \
        v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;                                                                  \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                               \
    }
KERNEL_PK(k_pk_fma, OP3("v_pk_fma_f32"))
KERNEL_PK(k_pk_mul, OP2("v_pk_mul_f32"))
KERNEL_PK(k_pk_add, OP2("v_pk_add_f32"))

// LDS-pipe cross-lane moves (not VALU): ds_swizzle / ds_bpermute, and a 16-byte broadcast read
__global__ __launch_bounds__(256) void k_ds_swizzle(float *out, unsigned long long *cycles, int iters, unsigned long long mask) {
    float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,16)\n ds_swizzle_b32 %1, %1 offset:swizzle(SWAP,16)\n"
                           "ds_swizzle_b32 %2, %2 offset:swizzle(SWAP,16)\n ds_swizzle_b32 %3, %3 offset:swizzle(SWAP,16)\n"
                           "s_waitcnt lgkmcnt(0)\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_ds_bpermute(float *out, unsigned long long *cycles, int iters, unsigned long long mask) {
    float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    const int addr = ((threadIdx.x & 63) ^ 32) * 4;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n"
                           "ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(addr));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// a VALU stream (v_fma) with one LDS cross-lane move per 8 VALU instructions riding along: does the LDS pipe's work overlap?
__global__ __launch_bounds__(256) void k_fma_with_bpermute(float *out, unsigned long long *cycles, int iters, unsigned long long mask) {
    float a0 = threadIdx.x * 1e-3f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
          a7 = a0 + 7.f, b = 0.999f, c = 1e-4f, x = threadIdx.x;
    const int addr = ((threadIdx.x & 63) ^ 32) * 4;
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16(OP3("v_fma_f32") "s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %10, %11, %10\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "v"(x), "v"(addr));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + x;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(float *, unsigned long long *, int, unsigned long long);
static void run(const char *name, kern_t kern, float *d, unsigned long long *dc, int waves_per_simd, unsigned long long mask,
                int instr_per_iter) {
    const int iters = 1000, grid = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
    kern<<<grid, 256>>>(d, dc, 10, mask);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    kern<<<grid, 256>>>(d, dc, iters, mask);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double per = med / ((double)iters * instr_per_iter);   // shader cycles a wave spends per instruction
    // (s_memtime ticks at a constant rate that is NOT the shader clock on this part: the launch time is the yardstick)
    const double ns = ms * 1e6 / ((double)iters * instr_per_iter * waves_per_simd);
    printf("%-18s exec=%016llx waves/SIMD=%d  %7.2f memtime ticks/instr/wave  %6.3f ns of the SIMD per wave-instr (= %.2f cycles at 2.4 GHz)  launch %.3f ms\n",
           name, mask, waves_per_simd, per, ns, ns * 2.4, ms);
}

int main() {
    float *d;
    unsigned long long *dc;
    hipMalloc(&d, 256 * 16 * 256 * 4);
    hipMalloc(&dc, 256 * 16 * 4 * 8);
    const unsigned long long FULL = ~0ull, LOW32 = 0xffffffffull, HIGH32 = 0xffffffff00000000ull, LOW16 = 0xffffull,
                             ROWS02 = 0x0000ffff0000ffffull, ONE = 1ull, EVEN = 0x5555555555555555ull;
    struct { const char *n; kern_t k; int per; } plain[] = {
        {"v_fma_f32", k_fma, 128}, {"v_mul_f32", k_mul, 128}, {"v_add_f32", k_add, 128}, {"v_pk_fma_f32", k_pk_fma, 128},
        {"v_pk_mul_f32", k_pk_mul, 128}, {"v_pk_add_f32", k_pk_add, 128}, {"v_exp_f32", k_exp, 128}, {"v_rcp_f32", k_rcp, 128},
        {"v_sqrt_f32", k_sqrt, 128}, {"v_med3_f32", k_med3, 128}, {"v_cmp->vcc", k_cmp_vcc, 128},
        {"v_add_dpp shr1", k_dpp_shr1, 128}, {"v_add_dpp ror8", k_dpp_ror8, 128}, {"v_add_dpp quad", k_dpp_quad, 128},
        {"permlane32_swap", k_swap32, 128}, {"permlane16_swap", k_swap16, 128}};
    for (int wps : {1, 4})
        for (auto &p : plain) run(p.n, p.k, d, dc, wps, FULL, p.per);
    printf("--- EXEC masks (does a fully masked half-wave pass cost nothing?)\n");
    for (int wps : {4})
        for (auto m : {FULL, LOW32, HIGH32, LOW16, ROWS02, ONE, EVEN}) {
            run("v_fma_f32", k_fma, d, dc, wps, m, 128);
            run("v_pk_fma_f32", k_pk_fma, d, dc, wps, m, 128);
            run("v_exp_f32", k_exp, d, dc, wps, m, 128);
            run("v_add_dpp shr1", k_dpp_shr1, d, dc, wps, m, 128);
        }
    printf("--- mixtures (8 v_fma + 2 specials = 10 instructions; additive cost would be 8 x fma + 2 x special)\n");
    for (int wps : {1, 4, 8}) {
        run("8fma", k_fma, d, dc, wps, FULL, 128);
        run("8fma+2exp", k_mix_exp, d, dc, wps, FULL, 160);
        run("8fma+2swap32", k_mix_swap, d, dc, wps, FULL, 160);
        run("8fma+2dpp", k_mix_dpp, d, dc, wps, FULL, 160);
        run("8fma+2cmp", k_mix_cmp, d, dc, wps, FULL, 160);
        run("8fma+2med3", k_mix_med3, d, dc, wps, FULL, 160);
        run("4exp+4swap32", k_mix_exp_swap, d, dc, wps, FULL, 128);
        run("4exp+4dpp", k_mix_exp_dpp, d, dc, wps, FULL, 128);
    }
    printf("--- LDS-pipe cross-lane moves\n");
    for (int wps : {1, 4}) {
        run("ds_swizzle_b32", k_ds_swizzle, d, dc, wps, FULL, 64);
        run("ds_bpermute_b32", k_ds_bpermute, d, dc, wps, FULL, 64);
        run("8 fma + 1 bperm", k_fma_with_bpermute, d, dc, wps, FULL, 16 * 9);
    }
    return 0;
}
