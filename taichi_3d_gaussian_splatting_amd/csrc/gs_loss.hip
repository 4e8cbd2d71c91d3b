// gs_loss.hip -- fused photometric loss of the trainer, forward and hand-derived backward, gfx950.
//
//   L = (1 - lambda) * mean|x - y| + lambda * (1 - SSIM(x, y)),   x = clamp(prediction, 0, 1) (optional clamp)
//
// replaces, for one [H,W,3] / [3,H,W] image pair, the chain  clamp -> permute -> L1 -> pytorch_msssim.ssim
// (five separable 11-tap Gaussian convolutions) -> autograd of all of it, i.e. LossFunction.py:20-39 of the
// reference as it is driven by GaussianPointTrainer.py:167-176.  Eager PyTorch spends ~6.5 ms per 1920x1072
// training iteration there (5x the rasteriser); these two kernels move ~0.3 GB and are HBM-bound.
//
// SSIM definition (pytorch_msssim, the reference's dependency): 'valid' 11x11 Gaussian window (sigma 1.5),
// per channel; with mu = E[.], l = (2 mu_x mu_y + C1)/(mu_x^2 + mu_y^2 + C1),
// cs = (2 cov + C2)/(var_x + var_y + C2), map = l * cs, SSIM = mean(map).
//
// Backward: with the window sums as intermediate variables (mu_x, s_xx = E[x^2], s_xy = E[xy]),
//   d map/d mu_x = cs (2 mu_y - 2 mu_x l)/D1 + l (2 mu_x cs - 2 mu_y)/D2        =: A
//   d map/d s_xx = -l cs / D2                                                    =: B
//   d map/d s_xy = 2 l / D2                                                      =: C
//   d SSIM/d x(q) = 1/n * sum_p w(q - p) [A(p) + 2 x(q) B(p) + y(q) C(p)]
// so the forward kernel stores the three maps and the backward kernel is one more separable convolution.
//
// Tiling: a 256-thread block owns 32x16 pixels; the 42x26 halo of all three channels is staged in LDS with
// row-contiguous (coalesced) loads in either layout, then per channel: horizontal pass -> LDS -> vertical pass.
// Reductions are deterministic: per-block partial sums, then one fixed-order pass in double.
#include "gs_common.h"

namespace {

constexpr int LT_W = 32, LT_H = 16, WIN = 11, HALO = WIN - 1;
constexpr int IN_W = LT_W + HALO, IN_H = LT_H + HALO;
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

// exp(-k^2 / (2 * 1.5^2)) normalised, evaluated in fp32 like the eager implementation
__device__ const float kWin[WIN] = {1.028380357e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                    2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                    3.600077331e-02f, 7.598758209e-03f, 1.028380357e-03f};

__device__ __forceinline__ float clamp01(float v, int on) { return on ? fminf(fmaxf(v, 0.f), 1.f) : v; }

// Stage rows [y0, y0+rows) x cols [x0, x0+cols) of a 3-channel image into dst[c][r][col] (zero outside the image).
template <int ROWS, int COLS, int PITCH>
__device__ __forceinline__ void stage_image(const float *__restrict__ img, int hwc, int H, int W, int x0, int y0,
                                            int clamp, float (*dst)[ROWS][PITCH]) {
    if (hwc) {
        for (int i = threadIdx.x; i < ROWS * COLS * 3; i += GS_BLOCK) {
            const int r = i / (COLS * 3), rem = i - r * (COLS * 3), col = rem / 3, c = rem - col * 3;
            const int gy = y0 + r, gx = x0 + col;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            dst[c][r][col] = in ? clamp01(img[((size_t)gy * W + gx) * 3 + c], clamp) : 0.f;
        }
    } else {
        for (int i = threadIdx.x; i < ROWS * COLS * 3; i += GS_BLOCK) {
            const int c = i / (ROWS * COLS), rem = i - c * (ROWS * COLS), r = rem / COLS, col = rem - r * COLS;
            const int gy = y0 + r, gx = x0 + col;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
            dst[c][r][col] = in ? clamp01(img[((size_t)c * H + gy) * W + gx], clamp) : 0.f;
        }
    }
}

__global__ __launch_bounds__(GS_BLOCK) void loss_forward_kernel(
    const float *__restrict__ pred, int pred_hwc, int clamp, const float *__restrict__ gt, int H, int W,
    float *__restrict__ dmap, float *__restrict__ partials) {
    __shared__ float sx[3][IN_H][IN_W + 1];
    __shared__ float sy[3][IN_H][IN_W + 1];
    __shared__ float hz[5][IN_H][LT_W + 1];
    __shared__ float red[2][GS_BLOCK / GS_WAVE];
    const int x0 = blockIdx.x * LT_W, y0 = blockIdx.y * LT_H;
    stage_image<IN_H, IN_W, IN_W + 1>(pred, pred_hwc, H, W, x0, y0, clamp, sx);
    stage_image<IN_H, IN_W, IN_W + 1>(gt, 0, H, W, x0, y0, 0, sy);
    __syncthreads();
    const size_t plane = (size_t)H * W;
    float l1 = 0.f, ssim_sum = 0.f;
    for (int c = 0; c < 3; ++c) {
        for (int i = threadIdx.x; i < IN_H * LT_W; i += GS_BLOCK) {   // horizontal pass, 5 window sums
            const int r = i / LT_W, col = i - r * LT_W;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float a = sx[c][r][col + k], b = sy[c][r][col + k], w = kWin[k];
                s0 = fmaf(w, a, s0); s1 = fmaf(w, b, s1);
                s2 = fmaf(w, a * a, s2); s3 = fmaf(w, b * b, s3); s4 = fmaf(w, a * b, s4);
            }
            hz[0][r][col] = s0; hz[1][r][col] = s1; hz[2][r][col] = s2; hz[3][r][col] = s3; hz[4][r][col] = s4;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < LT_H * LT_W; i += GS_BLOCK) {   // vertical pass + SSIM map and its partials
            const int r = i / LT_W, col = i - r * LT_W;
            const int gy = y0 + r, gx = x0 + col;
            if (gy >= H || gx >= W) continue;
            l1 += fabsf(sx[c][r][col] - sy[c][r][col]);
            float A = 0.f, B = 0.f, Cm = 0.f;
            if (gy < H - HALO && gx < W - HALO) {
                float mx = 0.f, my = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
                for (int k = 0; k < WIN; ++k) {
                    const float w = kWin[k];
                    mx = fmaf(w, hz[0][r + k][col], mx); my = fmaf(w, hz[1][r + k][col], my);
                    sxx = fmaf(w, hz[2][r + k][col], sxx); syy = fmaf(w, hz[3][r + k][col], syy);
                    sxy = fmaf(w, hz[4][r + k][col], sxy);
                }
                const float mxx = mx * mx, myy = my * my, mxy = mx * my;
                const float d1 = mxx + myy + C1, d2 = (sxx - mxx) + (syy - myy) + C2;
                const float l = (2.f * mxy + C1) / d1, cs = (2.f * (sxy - mxy) + C2) / d2;
                ssim_sum += l * cs;
                A = cs * (2.f * my - 2.f * mx * l) / d1 + l * (2.f * mx * cs - 2.f * my) / d2;
                B = -l * cs / d2;
                Cm = 2.f * l / d2;
            }
            if (dmap) {
                const size_t o = (size_t)gy * W + gx;
                dmap[(0 * 3 + c) * plane + o] = A;
                dmap[(1 * 3 + c) * plane + o] = B;
                dmap[(2 * 3 + c) * plane + o] = Cm;
            }
        }
        __syncthreads();
    }
    l1 = gs_wave_sum_to_lane63(l1);
    ssim_sum = gs_wave_sum_to_lane63(ssim_sum);
    const int wave = threadIdx.x / GS_WAVE;
    if (gs_lane() == GS_WAVE - 1) { red[0][wave] = l1; red[1][wave] = ssim_sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) { a += red[0][i]; b += red[1][i]; }
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        partials[2 * blk] = a;
        partials[2 * blk + 1] = b;
    }
}

// out = {total, L1, 1 - SSIM}; fixed summation order, double accumulators
__global__ __launch_bounds__(GS_BLOCK) void loss_finalize_kernel(const float *__restrict__ partials, int n_blocks,
                                                                 int H, int W, float lambda, float *__restrict__ out) {
    __shared__ double red[2][GS_BLOCK];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += GS_BLOCK) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = GS_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l1 = red[0][0] / (3.0 * H * W);
        const double dssim = 1.0 - red[1][0] / (3.0 * (H - HALO) * (double)(W - HALO));
        out[0] = (float)((1.0 - lambda) * l1 + lambda * dssim);
        out[1] = (float)l1;
        out[2] = (float)dssim;
    }
}

__global__ __launch_bounds__(GS_BLOCK) void loss_backward_kernel(
    const float *__restrict__ pred, int pred_hwc, int clamp, const float *__restrict__ gt,
    const float *__restrict__ dmap, int H, int W, float lambda, const float *__restrict__ g_total,
    const float *__restrict__ g_l1, const float *__restrict__ g_dssim, float *__restrict__ grad) {
    __shared__ float sm[3][IN_H][IN_W + 1];    // A, B, C of one channel, with the 10-pixel halo up/left
    __shared__ float hz[3][IN_H][LT_W + 1];
    __shared__ float sx[3][LT_H][LT_W + 1];    // raw prediction (unclamped: the clamp mask needs it)
    __shared__ float sy[3][LT_H][LT_W + 1];
    __shared__ float sg[3][LT_H][LT_W + 1];
    const int x0 = blockIdx.x * LT_W, y0 = blockIdx.y * LT_H;
    const float gt_total = g_total ? *g_total : 0.f;
    const float w_l1 = (gt_total * (1.f - lambda) + (g_l1 ? *g_l1 : 0.f)) / (3.f * (float)H * (float)W);
    const float w_ss = -(gt_total * lambda + (g_dssim ? *g_dssim : 0.f)) /
                       (3.f * (float)(H - HALO) * (float)(W - HALO));
    stage_image<LT_H, LT_W, LT_W + 1>(pred, pred_hwc, H, W, x0, y0, 0, sx);
    stage_image<LT_H, LT_W, LT_W + 1>(gt, 0, H, W, x0, y0, 0, sy);
    const size_t plane = (size_t)H * W;
    for (int c = 0; c < 3; ++c) {
        for (int i = threadIdx.x; i < 3 * IN_H * IN_W; i += GS_BLOCK) {
            const int m = i / (IN_H * IN_W), rem = i - m * (IN_H * IN_W), r = rem / IN_W, col = rem - r * IN_W;
            const int gy = y0 - HALO + r, gx = x0 - HALO + col;
            const bool in = gy >= 0 && gx >= 0 && gy < H && gx < W;
            sm[m][r][col] = in ? dmap[(m * 3 + c) * plane + (size_t)gy * W + gx] : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 3 * IN_H * LT_W; i += GS_BLOCK) {
            const int m = i / (IN_H * LT_W), rem = i - m * (IN_H * LT_W), r = rem / LT_W, col = rem - r * LT_W;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) s = fmaf(kWin[k], sm[m][r][col + k], s);
            hz[m][r][col] = s;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < LT_H * LT_W; i += GS_BLOCK) {
            const int r = i / LT_W, col = i - r * LT_W;
            float cA = 0.f, cB = 0.f, cC = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) {
                const float w = kWin[k];
                cA = fmaf(w, hz[0][r + k][col], cA); cB = fmaf(w, hz[1][r + k][col], cB);
                cC = fmaf(w, hz[2][r + k][col], cC);
            }
            const float raw = sx[c][r][col], y = sy[c][r][col];
            const float x = clamp01(raw, clamp);
            const float d = x - y;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            float g = w_l1 * sgn + w_ss * (cA + 2.f * x * cB + y * cC);
            if (clamp && !(raw >= 0.f && raw <= 1.f)) g = 0.f;   // torch.clamp passes the gradient on [min, max]
            sg[c][r][col] = g;
        }
        __syncthreads();
    }
    if (pred_hwc) {
        for (int i = threadIdx.x; i < LT_H * LT_W * 3; i += GS_BLOCK) {
            const int r = i / (LT_W * 3), rem = i - r * (LT_W * 3), col = rem / 3, c = rem - col * 3;
            const int gy = y0 + r, gx = x0 + col;
            if (gy < H && gx < W) grad[((size_t)gy * W + gx) * 3 + c] = sg[c][r][col];
        }
    } else {
        for (int i = threadIdx.x; i < LT_H * LT_W * 3; i += GS_BLOCK) {
            const int c = i / (LT_H * LT_W), rem = i - c * (LT_H * LT_W), r = rem / LT_W, col = rem - r * LT_W;
            const int gy = y0 + r, gx = x0 + col;
            if (gy < H && gx < W) grad[((size_t)c * H + gy) * W + gx] = sg[c][r][col];
        }
    }
}

}  // namespace

extern "C" {

long long gs_loss_workspace_floats(int height, int width) {
    return 2LL * gs_div_up(width, LT_W) * gs_div_up(height, LT_H);
}

int gs_loss_forward(const float *prediction, int prediction_is_hwc, int clamp01_prediction, const float *target,
                    int height, int width, float lambda, float *ssim_grad_maps, float *workspace, float *losses,
                    void *stream) {
    GS_REQUIRE(height >= WIN && width >= WIN, "gs_loss_forward: the image must be at least 11x11");
    GS_REQUIRE(prediction && target && workspace && losses, "gs_loss_forward: null pointer");
    const dim3 grid(gs_div_up(width, LT_W), gs_div_up(height, LT_H));
    hipLaunchKernelGGL(loss_forward_kernel, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, prediction,
                       prediction_is_hwc, clamp01_prediction, target, height, width, ssim_grad_maps, workspace);
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(GS_BLOCK), 0, (hipStream_t)stream, workspace,
                       (int)(grid.x * grid.y), height, width, lambda, losses);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_loss_backward(const float *prediction, int prediction_is_hwc, int clamp01_prediction, const float *target,
                     const float *ssim_grad_maps, int height, int width, float lambda, const float *grad_total,
                     const float *grad_l1, const float *grad_dssim, float *grad_prediction, void *stream) {
    GS_REQUIRE(height >= WIN && width >= WIN, "gs_loss_backward: the image must be at least 11x11");
    GS_REQUIRE(prediction && target && ssim_grad_maps && grad_prediction, "gs_loss_backward: null pointer");
    const dim3 grid(gs_div_up(width, LT_W), gs_div_up(height, LT_H));
    hipLaunchKernelGGL(loss_backward_kernel, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, prediction,
                       prediction_is_hwc, clamp01_prediction, target, ssim_grad_maps, height, width, lambda,
                       grad_total, grad_l1, grad_dssim, grad_prediction);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Scale regulariser of the trainer: R = mean over live Gaussians of || exp(s) ||_2   (LossFunction.py:42-54).
// Eager autograd materialises a dense [N,56] gradient for it and adds it to the rasteriser's (3 x 224 MB of
// traffic at N = 1e6); here the value is one reduction and the gradient is added in place into columns 4..6 of the
// existing feature gradient.
namespace {

__global__ __launch_bounds__(GS_BLOCK) void scale_reg_partials_kernel(const float *__restrict__ feat,
                                                                       const int8_t *__restrict__ invalid, int n,
                                                                       float *__restrict__ partials) {
    __shared__ float red[2][GS_BLOCK / GS_WAVE];
    float sum = 0.f, cnt = 0.f;
    for (int i = blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += gridDim.x * GS_BLOCK) {
        if (invalid[i] != 0) continue;
        const float *s = feat + (size_t)GS_FEATURE_DIM * i + 4;
        const float a = expf(s[0]), b = expf(s[1]), c = expf(s[2]);
        sum += sqrtf(a * a + b * b + c * c);
        cnt += 1.f;
    }
    sum = gs_wave_sum_to_lane63(sum);
    cnt = gs_wave_sum_to_lane63(cnt);
    if (gs_lane() == GS_WAVE - 1) { red[0][threadIdx.x / GS_WAVE] = sum; red[1][threadIdx.x / GS_WAVE] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) { a += red[0][i]; b += red[1][i]; }
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(GS_BLOCK) void scale_reg_finalize_kernel(const float *__restrict__ partials, int n_blocks,
                                                                       float *__restrict__ out) {
    __shared__ double red[2][GS_BLOCK];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += GS_BLOCK) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int s = GS_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = (float)(red[0][0] / red[1][0]); out[1] = (float)red[1][0]; }   // 0/0 = nan, like torch
}

__global__ __launch_bounds__(GS_BLOCK) void scale_reg_backward_kernel(const float *__restrict__ feat,
                                                                       const int8_t *__restrict__ invalid, int n,
                                                                       const float *__restrict__ value_and_count,
                                                                       float weight, const float *__restrict__ upstream,
                                                                       float *__restrict__ grad_feat) {
    const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i >= n || invalid[i] != 0) return;
    const float scale = weight * (upstream ? *upstream : 1.f) / value_and_count[1];
    const float *s = feat + (size_t)GS_FEATURE_DIM * i + 4;
    float *g = grad_feat + (size_t)GS_FEATURE_DIM * i + 4;
    const float a = expf(s[0]), b = expf(s[1]), c = expf(s[2]);
    const float inv = scale / sqrtf(a * a + b * b + c * c);
    g[0] = fmaf(a * a, inv, g[0]);
    g[1] = fmaf(b * b, inv, g[1]);
    g[2] = fmaf(c * c, inv, g[2]);
}

constexpr int REG_BLOCKS = 1024;

}  // namespace

extern "C" {

long long gs_scale_regulariser_workspace_floats(void) { return 2LL * REG_BLOCKS; }

int gs_scale_regulariser_forward(const float *features, const int8_t *point_invalid_mask, int n_points,
                                 float *workspace, float *value_and_count, void *stream) {
    GS_REQUIRE(n_points >= 0 && workspace && value_and_count, "gs_scale_regulariser_forward: bad argument");
    const int blocks = n_points > 0 ? (gs_div_up(n_points, GS_BLOCK) < REG_BLOCKS ? gs_div_up(n_points, GS_BLOCK) : REG_BLOCKS) : 0;
    if (blocks > 0) {
        hipLaunchKernelGGL(scale_reg_partials_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, features,
                           point_invalid_mask, n_points, workspace);
        GS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(scale_reg_finalize_kernel, dim3(1), dim3(GS_BLOCK), 0, (hipStream_t)stream, workspace, blocks,
                       value_and_count);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_scale_regulariser_backward(const float *features, const int8_t *point_invalid_mask, int n_points,
                                  const float *value_and_count, float weight, const float *upstream,
                                  float *grad_features, void *stream) {
    GS_REQUIRE(n_points >= 0 && value_and_count && grad_features, "gs_scale_regulariser_backward: bad argument");
    if (n_points == 0) return 0;
    hipLaunchKernelGGL(scale_reg_backward_kernel, dim3(gs_div_up(n_points, GS_BLOCK)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, features, point_invalid_mask, n_points, value_and_count, weight, upstream,
                       grad_features);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
