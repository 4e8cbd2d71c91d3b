// gs_optim.hip -- Adam step for the trainer's two parameter tensors, gfx950.
// The reference uses torch.optim.Adam (GaussianPointTrainer.py:126-129: betas (0.9, 0.999), eps 1e-8, no weight
// decay, no amsgrad).  One streaming pass: 16 B read + 12 B written per element, float4-vectorised; the update
// is written in the operation order of torch's implementation
//   m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g g; p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// so results agree with it to the last bit or ulp.
#include "gs_common.h"

namespace {

struct AdamScalars { float lr_over_bc1, inv_sqrt_bc2, one_minus_beta1, beta2, one_minus_beta2, eps; };

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamScalars &k) {
    m = m + k.one_minus_beta1 * (g - m);                        // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = k.beta2 * v + k.one_minus_beta2 * g * g;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) * k.inv_sqrt_bc2 + k.eps;
    p = p - k.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(GS_BLOCK) void adam_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                        float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                        long long n, AdamScalars k) {
    const long long n4 = n / 4;
    float4 *p4 = reinterpret_cast<float4 *>(param);
    const float4 *g4 = reinterpret_cast<const float4 *>(grad);
    float4 *m4 = reinterpret_cast<float4 *>(exp_avg), *v4 = reinterpret_cast<float4 *>(exp_avg_sq);
    const long long stride = (long long)gridDim.x * GS_BLOCK;
    for (long long i = (long long)blockIdx.x * GS_BLOCK + threadIdx.x; i < n4; i += stride) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        const float4 g = g4[i];
        adam_one(p.x, g.x, m.x, v.x, k);
        adam_one(p.y, g.y, m.y, v.y, k);
        adam_one(p.z, g.z, m.z, v.z, k);
        adam_one(p.w, g.w, m.w, v.w, k);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {   // tail (n not a multiple of 4)
        const long long i = 4 * n4 + threadIdx.x;
        adam_one(param[i], grad[i], exp_avg[i], exp_avg_sq[i], k);
    }
}

}  // namespace

extern "C" int gs_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, double lr,
                            double beta1, double beta2, double eps, int step, void *stream) {
    GS_REQUIRE(n >= 0 && step >= 1, "gs_adam_step: n >= 0 and step >= 1");
    GS_REQUIRE(param && grad && exp_avg && exp_avg_sq, "gs_adam_step: null pointer");
    GS_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
               "gs_adam_step: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    AdamScalars k;
    // scalars are prepared in double like the Python implementation does (1 - 0.999 is 1e-3, not 1 - 0.999f)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    k.lr_over_bc1 = (float)(lr / bc1);
    k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    k.one_minus_beta1 = (float)(1.0 - beta1);
    k.beta2 = (float)beta2;
    k.one_minus_beta2 = (float)(1.0 - beta2);
    k.eps = (float)eps;
    long long want = (n / 4 + GS_BLOCK - 1) / GS_BLOCK;
    const int blocks = (int)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, k);
    GS_CHECK_LAUNCH();
    return 0;
}
