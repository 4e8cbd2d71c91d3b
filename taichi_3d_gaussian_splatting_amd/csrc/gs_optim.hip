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

// Feature-matrix variant: rows of 56 floats = 14 float4; float4 #1 of a row holds (s0, s1, s2, opacity logit).  The
// gradient of the trainer's scale regulariser  w * mean_live ||exp(s)||  (LossFunction.py:42-54) is a function of those
// three parameters only, so it is added to the incoming gradient right here instead of in a pass of its own over the
// feature and gradient matrices (75 us per iteration at 1.2e6 rows): g_k += w / n_live * exp(s_k)^2 / ||exp(s)||.
__global__ __launch_bounds__(GS_BLOCK) void count_live_rows_kernel(const int8_t *__restrict__ invalid, int n,
                                                                   int *__restrict__ partial_counts) {
    __shared__ int red[GS_BLOCK / GS_WAVE];
    int cnt = 0;
    for (int i = blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += gridDim.x * GS_BLOCK) cnt += invalid[i] == 0 ? 1 : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, GS_WAVE);
    if (gs_lane() == 0) red[threadIdx.x / GS_WAVE] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) t += red[i];
        partial_counts[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(GS_BLOCK) void adam_features_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                                 float *__restrict__ exp_avg,
                                                                 float *__restrict__ exp_avg_sq, long long n_rows,
                                                                 AdamScalars k, const int8_t *__restrict__ invalid,
                                                                 const int *__restrict__ partial_counts,
                                                                 float reg_weight) {
    __shared__ int red[GS_BLOCK / GS_WAVE];
    int c = partial_counts[threadIdx.x];     // GS_BLOCK partial live counts -> n_live in every block
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, GS_WAVE);
    if (gs_lane() == 0) red[threadIdx.x / GS_WAVE] = c;
    __syncthreads();
    int n_live = 0;
    for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) n_live += red[i];
    const float reg_scale = reg_weight / (float)n_live;
    float4 *p4 = reinterpret_cast<float4 *>(param);
    const float4 *g4 = reinterpret_cast<const float4 *>(grad);
    float4 *m4 = reinterpret_cast<float4 *>(exp_avg), *v4 = reinterpret_cast<float4 *>(exp_avg_sq);
    // 32-bit indices (the entry point checks 14 * n_rows < 2^31): a 64-bit division per float4 costs more than the update
    const unsigned n4 = (unsigned)n_rows * 14u, stride = gridDim.x * GS_BLOCK;
    for (unsigned i = blockIdx.x * GS_BLOCK + threadIdx.x; i < n4; i += stride) {
        float4 p = p4[i], m = m4[i], v = v4[i];
        float4 g = g4[i];
        const unsigned row = i / 14u;
        if (i - row * 14u == 1u && invalid[row] == 0) {
            const float a = expf(p.x), b = expf(p.y), cc = expf(p.z);
            const float inv = reg_scale / sqrtf(a * a + b * b + cc * cc);
            g.x = fmaf(a * a, inv, g.x);
            g.y = fmaf(b * b, inv, g.y);
            g.z = fmaf(cc * cc, inv, g.z);
        }
        adam_one(p.x, g.x, m.x, v.x, k);
        adam_one(p.y, g.y, m.y, v.y, k);
        adam_one(p.z, g.z, m.z, v.z, k);
        adam_one(p.w, g.w, m.w, v.w, k);
        p4[i] = p; m4[i] = m; v4[i] = v;
    }
}

// Row-masked variant for the trainer's fixed-capacity tensors (the reference's Truck configuration allocates 10x the
// initial points, config/tat_truck_every_8_test.yaml:20): rows whose point is invalid are not touched at all.  torch's
// Adam would keep decaying their moments (their gradient is exactly zero) and move their parameters; the parameters of
// a dead row are never read again -- the controller overwrites them when it re-uses the row (ADC:303-321) -- and the
// decay is applied lazily: `last_step[row]` remembers the last step that updated the row, and the first update after a
// gap of d steps starts from m * beta1^d, v * beta2^d, which is what d zero-gradient steps leave behind.
template <int ROW_FLOAT4>   // 14: feature rows (with the optional scale regulariser); 0: rows of three floats
__global__ __launch_bounds__(GS_BLOCK) void adam_rows_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                             float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq,
                                                             long long n_rows, AdamScalars k, float beta1, int step,
                                                             const int8_t *__restrict__ invalid,
                                                             int32_t *__restrict__ last_step,
                                                             const int *__restrict__ partial_counts, float reg_weight) {
    float reg_scale = 0.f;
    if (ROW_FLOAT4 > 0 && reg_weight != 0.f) {
        __shared__ int red[GS_BLOCK / GS_WAVE];
        int c = partial_counts[threadIdx.x];     // GS_BLOCK partial live counts -> n_live in every block
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, GS_WAVE);
        if (gs_lane() == 0) red[threadIdx.x / GS_WAVE] = c;
        __syncthreads();
        int n_live = 0;
        for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) n_live += red[i];
        reg_scale = reg_weight / (float)n_live;
    }
    const unsigned stride = gridDim.x * GS_BLOCK;
    if (ROW_FLOAT4 > 0) {
        float4 *p4 = reinterpret_cast<float4 *>(param);
        const float4 *g4 = reinterpret_cast<const float4 *>(grad);
        float4 *m4 = reinterpret_cast<float4 *>(exp_avg), *v4 = reinterpret_cast<float4 *>(exp_avg_sq);
        const unsigned n4 = (unsigned)n_rows * (unsigned)ROW_FLOAT4;
        for (unsigned i = blockIdx.x * GS_BLOCK + threadIdx.x; i < n4; i += stride) {
            const unsigned row = i / (unsigned)(ROW_FLOAT4 > 0 ? ROW_FLOAT4 : 1), col = i - row * (unsigned)ROW_FLOAT4;
            if (invalid[row] != 0) continue;
            const int gap = step - 1 - last_step[row];
            float4 p = p4[i], m = m4[i], v = v4[i];
            float4 g = g4[i];
            if (gap > 0) {
                const float dm = powf(beta1, (float)gap), dv = powf(k.beta2, (float)gap);
                m.x *= dm; m.y *= dm; m.z *= dm; m.w *= dm;
                v.x *= dv; v.y *= dv; v.z *= dv; v.w *= dv;
            }
            if (col == 1u && reg_scale != 0.f) {
                const float a = expf(p.x), b = expf(p.y), cc = expf(p.z);
                const float inv = reg_scale / sqrtf(a * a + b * b + cc * cc);
                g.x = fmaf(a * a, inv, g.x);
                g.y = fmaf(b * b, inv, g.y);
                g.z = fmaf(cc * cc, inv, g.z);
            }
            adam_one(p.x, g.x, m.x, v.x, k);
            adam_one(p.y, g.y, m.y, v.y, k);
            adam_one(p.z, g.z, m.z, v.z, k);
            adam_one(p.w, g.w, m.w, v.w, k);
            p4[i] = p; m4[i] = m; v4[i] = v;
        }
        // (the row's stamp is advanced by adam_stamp_rows_kernel afterwards: the 14 threads of a row all need the old one)
    } else {
        for (unsigned row = blockIdx.x * GS_BLOCK + threadIdx.x; row < (unsigned)n_rows; row += stride) {
            if (invalid[row] != 0) continue;
            const int gap = step - 1 - last_step[row];
            const float dm = gap > 0 ? powf(beta1, (float)gap) : 1.f, dv = gap > 0 ? powf(k.beta2, (float)gap) : 1.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t e = 3 * (size_t)row + c;
                float p = param[e], m = exp_avg[e] * dm, v = exp_avg_sq[e] * dv;
                adam_one(p, grad[e], m, v, k);
                param[e] = p; exp_avg[e] = m; exp_avg_sq[e] = v;
            }
            last_step[row] = step;
        }
    }
}

// stamps of the feature rows: advanced after the update kernel (the 14 threads of a row all read the old stamp)
__global__ __launch_bounds__(GS_BLOCK) void adam_stamp_rows_kernel(const int8_t *__restrict__ invalid, long long n_rows,
                                                                   int step, int32_t *__restrict__ last_step) {
    const long long row = (long long)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (row < n_rows && invalid[row] == 0) last_step[row] = step;
}

static AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, int step) {
    // scalars are prepared in double like the Python implementation does (1 - 0.999 is 1e-3, not 1 - 0.999f)
    AdamScalars k;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    k.lr_over_bc1 = (float)(lr / bc1);
    k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    k.one_minus_beta1 = (float)(1.0 - beta1);
    k.beta2 = (float)beta2;
    k.one_minus_beta2 = (float)(1.0 - beta2);
    k.eps = (float)eps;
    return k;
}

}  // namespace

extern "C" int gs_adam_step_features(float *features, const float *grad, float *exp_avg, float *exp_avg_sq,
                                     long long n_rows, double lr, double beta1, double beta2, double eps, int step,
                                     const int8_t *point_invalid_mask, double scale_regulariser_weight,
                                     int32_t *workspace, void *stream) {
    GS_REQUIRE(n_rows >= 0 && step >= 1, "gs_adam_step_features: n_rows >= 0 and step >= 1");
    GS_REQUIRE(n_rows * 14 < 0x7fffffffLL, "gs_adam_step_features: more than 2^31 / 14 rows");
    GS_REQUIRE(features && grad && exp_avg && exp_avg_sq && point_invalid_mask && workspace,
               "gs_adam_step_features: null pointer");
    GS_REQUIRE((((uintptr_t)features | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
               "gs_adam_step_features: buffers must be 16-byte aligned");
    if (n_rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(count_live_rows_kernel, dim3(GS_BLOCK), dim3(GS_BLOCK), 0, s, point_invalid_mask, (int)n_rows,
                       workspace);
    GS_CHECK_LAUNCH();
    long long want = (n_rows * 14 + GS_BLOCK - 1) / GS_BLOCK;
    const int blocks = (int)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
    hipLaunchKernelGGL(adam_features_kernel, dim3(blocks), dim3(GS_BLOCK), 0, s, features, grad, exp_avg, exp_avg_sq,
                       n_rows, adam_scalars(lr, beta1, beta2, eps, step), point_invalid_mask, workspace,
                       (float)scale_regulariser_weight);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int gs_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n, double lr,
                            double beta1, double beta2, double eps, int step, void *stream) {
    GS_REQUIRE(n >= 0 && step >= 1, "gs_adam_step: n >= 0 and step >= 1");
    GS_REQUIRE(param && grad && exp_avg && exp_avg_sq, "gs_adam_step: null pointer");
    GS_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
               "gs_adam_step: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    const AdamScalars k = adam_scalars(lr, beta1, beta2, eps, step);
    long long want = (n / 4 + GS_BLOCK - 1) / GS_BLOCK;
    const int blocks = (int)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, k);
    GS_CHECK_LAUNCH();
    return 0;
}


extern "C" int gs_adam_step_rows(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n_rows,
                                 int row_len, double lr, double beta1, double beta2, double eps, int step,
                                 const int8_t *point_invalid_mask, int32_t *last_step,
                                 double scale_regulariser_weight, int32_t *workspace, void *stream) {
    GS_REQUIRE(n_rows >= 0 && step >= 1, "gs_adam_step_rows: n_rows >= 0 and step >= 1");
    GS_REQUIRE(row_len == 3 || row_len == GS_FEATURE_DIM, "gs_adam_step_rows: rows of 3 or 56 floats");
    GS_REQUIRE(n_rows * 14 < 0x7fffffffLL, "gs_adam_step_rows: more than 2^31 / 14 rows");
    GS_REQUIRE(param && grad && exp_avg && exp_avg_sq && point_invalid_mask && last_step, "gs_adam_step_rows: null pointer");
    GS_REQUIRE(scale_regulariser_weight == 0.0 || (row_len == GS_FEATURE_DIM && workspace != nullptr),
               "gs_adam_step_rows: the scale regulariser needs feature rows and a workspace");
    GS_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
               "gs_adam_step_rows: buffers must be 16-byte aligned");
    if (n_rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const AdamScalars k = adam_scalars(lr, beta1, beta2, eps, step);
    if (row_len == GS_FEATURE_DIM) {
        if (scale_regulariser_weight != 0.0) {
            hipLaunchKernelGGL(count_live_rows_kernel, dim3(GS_BLOCK), dim3(GS_BLOCK), 0, s, point_invalid_mask,
                               (int)n_rows, workspace);
            GS_CHECK_LAUNCH();
        }
        long long want = (n_rows * 14 + GS_BLOCK - 1) / GS_BLOCK;
        const int blocks = (int)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
        hipLaunchKernelGGL(adam_rows_kernel<14>, dim3(blocks), dim3(GS_BLOCK), 0, s, param, grad, exp_avg, exp_avg_sq,
                           n_rows, k, (float)beta1, step, point_invalid_mask, last_step, workspace,
                           (float)scale_regulariser_weight);
        GS_CHECK_LAUNCH();
        hipLaunchKernelGGL(adam_stamp_rows_kernel, dim3(gs_div_up(n_rows, GS_BLOCK)), dim3(GS_BLOCK), 0, s,
                           point_invalid_mask, n_rows, step, last_step);
    } else {
        long long want = (n_rows + GS_BLOCK - 1) / GS_BLOCK;
        const int blocks = (int)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
        hipLaunchKernelGGL(adam_rows_kernel<0>, dim3(blocks), dim3(GS_BLOCK), 0, s, param, grad, exp_avg, exp_avg_sq,
                           n_rows, k, (float)beta1, step, point_invalid_mask, last_step, (const int *)nullptr, 0.f);
    }
    GS_CHECK_LAUNCH();
    return 0;
}
