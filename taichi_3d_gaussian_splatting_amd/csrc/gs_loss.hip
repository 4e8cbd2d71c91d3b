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
// Tiling: a 256-thread block owns one channel of a 32x16 pixel tile; its 42x26 halo is staged in LDS (26 KB per
// block -> 6 blocks per CU), then horizontal pass -> LDS -> vertical pass, each thread producing 4 (2) adjacent
// outputs from a sliding register window so that every LDS value is read once per thread, not once per tap.
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

__device__ __forceinline__ size_t pixel_index(int hwc, int c, int gy, int gx, int H, int W) {
    return hwc ? ((size_t)gy * W + gx) * 3 + c : ((size_t)c * H + gy) * W + gx;
}

// A workgroup owns one channel of a 32x16 pixel tile (blockIdx.x = 3 * tile_x + channel, so the three workgroups
// that share the cache lines of an HWC image are dispatched together).
struct TileId { int c, x0, y0; };
__device__ __forceinline__ TileId tile_of_block() {
    TileId t;
    t.c = blockIdx.x % 3;
    t.x0 = (blockIdx.x / 3) * LT_W;
    t.y0 = blockIdx.y * LT_H;
    return t;
}

// Stage rows [y0, y0+IN_H) x cols [x0, x0+IN_W) of one channel into dst[r][col] (zero outside the image).
// Lane = column (42 of 64 lanes active), wave = row phase: no integer division, one address increment per row.
__device__ __forceinline__ void stage_plane(const float *__restrict__ img, int hwc, int c, int H, int W, int x0, int y0,
                                            int clamp, float (*dst)[IN_W + 1]) {
    const int col = threadIdx.x & (GS_WAVE - 1), phase = threadIdx.x / GS_WAVE;
    if (col >= IN_W) return;
    const int gx = x0 + col;
    const bool col_in = gx >= 0 && gx < W;
    const size_t row_stride = hwc ? (size_t)3 * W : (size_t)W;
    const float *src = img + (hwc ? (size_t)gx * 3 + c : (size_t)c * H * W + gx);
#pragma unroll
    for (int r = phase; r < IN_H; r += GS_BLOCK / GS_WAVE) {
        const int gy = y0 + r;
        float v = 0.f;
        if (col_in && gy >= 0 && gy < H) v = clamp01(src[(size_t)gy * row_stride], clamp);
        dst[r][col] = v;
    }
}

constexpr int SEG = 4;                  // outputs per thread in the horizontal passes (sliding 14-value window)
constexpr int SEGS = LT_W / SEG;        // 8 segments per row
constexpr int VROWS = 2;                // outputs per thread in the vertical passes (sliding 12-value window)

__global__ __launch_bounds__(GS_BLOCK) void loss_forward_kernel(
    const float *__restrict__ pred, int pred_hwc, int clamp, const float *__restrict__ gt, int H, int W,
    float *__restrict__ dmap, float *__restrict__ partials) {
    __shared__ float sx[IN_H][IN_W + 1];
    __shared__ float sy[IN_H][IN_W + 1];
    __shared__ float hz[5][IN_H][LT_W + 1];
    __shared__ float red[2][GS_BLOCK / GS_WAVE];
    const TileId t = tile_of_block();
    stage_plane(pred, pred_hwc, t.c, H, W, t.x0, t.y0, clamp, sx);
    stage_plane(gt, 0, t.c, H, W, t.x0, t.y0, 0, sy);
    __syncthreads();
    if (threadIdx.x < IN_H * SEGS) {   // horizontal pass: 4 adjacent outputs of one row, 5 window sums each
        const int r = threadIdx.x / SEGS, c0 = (threadIdx.x % SEGS) * SEG;
        float a[SEG + HALO], b[SEG + HALO];
#pragma unroll
        for (int j = 0; j < SEG + HALO; ++j) { a[j] = sx[r][c0 + j]; b[j] = sy[r][c0 + j]; }
        float acc[5][SEG];
#pragma unroll
        for (int o = 0; o < SEG; ++o)
#pragma unroll
            for (int m = 0; m < 5; ++m) acc[m][o] = 0.f;
#pragma unroll
        for (int j = 0; j < SEG + HALO; ++j) {
            const float aa = a[j] * a[j], bb = b[j] * b[j], ab = a[j] * b[j];
#pragma unroll
            for (int o = 0; o < SEG; ++o) {
                const int k = j - o;   // tap index of input j for output o
                if (k >= 0 && k < WIN) {
                    const float w = kWin[k];
                    acc[0][o] = fmaf(w, a[j], acc[0][o]); acc[1][o] = fmaf(w, b[j], acc[1][o]);
                    acc[2][o] = fmaf(w, aa, acc[2][o]); acc[3][o] = fmaf(w, bb, acc[3][o]);
                    acc[4][o] = fmaf(w, ab, acc[4][o]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int o = 0; o < SEG; ++o) hz[m][r][c0 + o] = acc[m][o];
    }
    __syncthreads();
    float l1 = 0.f, ssim_sum = 0.f;
    {   // vertical pass: 2 vertically adjacent outputs per thread
        const int col = threadIdx.x % LT_W, r0 = (threadIdx.x / LT_W) * VROWS;
        float out[5][VROWS];
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            float v[VROWS + HALO];
#pragma unroll
            for (int j = 0; j < VROWS + HALO; ++j) v[j] = hz[m][r0 + j][col];
#pragma unroll
            for (int o = 0; o < VROWS; ++o) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < WIN; ++k) s = fmaf(kWin[k], v[o + k], s);
                out[m][o] = s;
            }
        }
        const size_t plane = (size_t)H * W;
        const int gx = t.x0 + col;
#pragma unroll
        for (int o = 0; o < VROWS; ++o) {
            const int gy = t.y0 + r0 + o;
            if (gy >= H || gx >= W) continue;
            l1 += fabsf(sx[r0 + o][col] - sy[r0 + o][col]);
            float A = 0.f, B = 0.f, Cm = 0.f;
            if (gy < H - HALO && gx < W - HALO) {
                const float mx = out[0][o], my = out[1][o], sxx = out[2][o], syy = out[3][o], sxy = out[4][o];
                const float mxx = mx * mx, myy = my * my, mxy = mx * my;
                // v_rcp_f32 (1 ulp) instead of five IEEE divisions: far inside the fp32 noise of var = E[x^2] - mu^2
                const float i1 = __builtin_amdgcn_rcpf(mxx + myy + C1);
                const float i2 = __builtin_amdgcn_rcpf((sxx - mxx) + (syy - myy) + C2);
                const float l = (2.f * mxy + C1) * i1, cs = (2.f * (sxy - mxy) + C2) * i2;
                ssim_sum += l * cs;
                A = cs * (2.f * my - 2.f * mx * l) * i1 + l * (2.f * mx * cs - 2.f * my) * i2;
                B = -l * cs * i2;
                Cm = 2.f * l * i2;
            }
            if (dmap) {
                const size_t p = (size_t)gy * W + gx;
                dmap[(0 * 3 + t.c) * plane + p] = A;
                dmap[(1 * 3 + t.c) * plane + p] = B;
                dmap[(2 * 3 + t.c) * plane + p] = Cm;
            }
        }
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
    __shared__ float sm[3][IN_H][IN_W + 1];    // A, B, C of this channel, with the 10-pixel halo up/left
    __shared__ float hz[3][IN_H][LT_W + 1];
    const TileId t = tile_of_block();
    const float gt_total = g_total ? *g_total : 0.f;
    const float w_l1 = (gt_total * (1.f - lambda) + (g_l1 ? *g_l1 : 0.f)) / (3.f * (float)H * (float)W);
    const float w_ss = -(gt_total * lambda + (g_dssim ? *g_dssim : 0.f)) /
                       (3.f * (float)(H - HALO) * (float)(W - HALO));
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int m = 0; m < 3; ++m)
        stage_plane(dmap + (size_t)(m * 3 + t.c) * plane, 0, 0, H, W, t.x0 - HALO, t.y0 - HALO, 0, sm[m]);
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * IN_H * SEGS; i += GS_BLOCK) {   // horizontal pass, 4 outputs per item
        const int m = i / (IN_H * SEGS), rem = i - m * (IN_H * SEGS), r = rem / SEGS, c0 = (rem % SEGS) * SEG;
        float v[SEG + HALO];
#pragma unroll
        for (int j = 0; j < SEG + HALO; ++j) v[j] = sm[m][r][c0 + j];
#pragma unroll
        for (int o = 0; o < SEG; ++o) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) s = fmaf(kWin[k], v[o + k], s);
            hz[m][r][c0 + o] = s;
        }
    }
    __syncthreads();
    const int col = threadIdx.x % LT_W, r0 = (threadIdx.x / LT_W) * VROWS;
    float conv[3][VROWS];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        float v[VROWS + HALO];
#pragma unroll
        for (int j = 0; j < VROWS + HALO; ++j) v[j] = hz[m][r0 + j][col];
#pragma unroll
        for (int o = 0; o < VROWS; ++o) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; ++k) s = fmaf(kWin[k], v[o + k], s);
            conv[m][o] = s;
        }
    }
    const int gx = t.x0 + col;
#pragma unroll
    for (int o = 0; o < VROWS; ++o) {
        const int gy = t.y0 + r0 + o;
        if (gy >= H || gx >= W) continue;
        const size_t pi = pixel_index(pred_hwc, t.c, gy, gx, H, W);
        const float raw = pred[pi], y = gt[pixel_index(0, t.c, gy, gx, H, W)];
        const float x = clamp01(raw, clamp);
        const float d = x - y;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        float g = w_l1 * sgn + w_ss * (conv[0][o] + 2.f * x * conv[1][o] + y * conv[2][o]);
        if (clamp && !(raw >= 0.f && raw <= 1.f)) g = 0.f;   // torch.clamp passes the gradient on [min, max]
        grad[pi] = g;
    }
}

}  // namespace

extern "C" {

long long gs_loss_workspace_floats(int height, int width) {
    return 2LL * 3 * gs_div_up(width, LT_W) * gs_div_up(height, LT_H);
}

int gs_loss_forward(const float *prediction, int prediction_is_hwc, int clamp01_prediction, const float *target,
                    int height, int width, float lambda, float *ssim_grad_maps, float *workspace, float *losses,
                    void *stream) {
    GS_REQUIRE(height >= WIN && width >= WIN, "gs_loss_forward: the image must be at least 11x11");
    GS_REQUIRE(prediction && target && workspace && losses, "gs_loss_forward: null pointer");
    const dim3 grid(3 * gs_div_up(width, LT_W), gs_div_up(height, LT_H));
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
    const dim3 grid(3 * gs_div_up(width, LT_W), gs_div_up(height, LT_H));
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
// traffic at N = 1e6).  Here: a 1-byte-per-row count of the live rows, then ONE pass over the 12 scale bytes of
// every row that produces the partial sums of the value and adds weight * dR/ds in place into columns 4..6 of the
// existing feature gradient, then a fixed-order reduction of the partial sums.
namespace {

constexpr int REG_BLOCKS = 1024;

constexpr int COUNT_BLOCKS = GS_BLOCK;   // partial live counts, summed by one workgroup-wide reduction in the consumers

__device__ __forceinline__ int block_sum_int(int v, int *red) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, GS_WAVE);
    if (gs_lane() == 0) red[threadIdx.x / GS_WAVE] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) t += red[i];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(GS_BLOCK) void count_live_kernel(const int8_t *__restrict__ invalid, int n,
                                                              int *__restrict__ partial_counts) {
    __shared__ int red[GS_BLOCK / GS_WAVE];
    int cnt = 0;
    for (int i = blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += COUNT_BLOCKS * GS_BLOCK) cnt += invalid[i] == 0 ? 1 : 0;
    const int t = block_sum_int(cnt, red);
    if (threadIdx.x == 0) partial_counts[blockIdx.x] = t;
}

__global__ __launch_bounds__(GS_BLOCK) void scale_reg_kernel(const float *__restrict__ feat,
                                                             const int8_t *__restrict__ invalid, int n,
                                                             const int *__restrict__ live_count, float weight,
                                                             const float *__restrict__ upstream,
                                                             float *__restrict__ grad_feat,
                                                             float *__restrict__ partials) {
    __shared__ float red[GS_BLOCK / GS_WAVE];
    __shared__ int redi[GS_BLOCK / GS_WAVE];
    const int live = block_sum_int(live_count[threadIdx.x], redi);
    const float scale = grad_feat ? weight * (upstream ? *upstream : 1.f) / (float)live : 0.f;
    float sum = 0.f;
    for (int i = blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += gridDim.x * GS_BLOCK) {
        if (invalid[i] != 0) continue;
        const float *s = feat + (size_t)GS_FEATURE_DIM * i + 4;
        const float a = expf(s[0]), b = expf(s[1]), c = expf(s[2]);
        const float nrm = sqrtf(a * a + b * b + c * c);
        sum += nrm;
        if (grad_feat) {
            float *g = grad_feat + (size_t)GS_FEATURE_DIM * i + 4;
            const float inv = scale / nrm;
            g[0] = fmaf(a * a, inv, g[0]);
            g[1] = fmaf(b * b, inv, g[1]);
            g[2] = fmaf(c * c, inv, g[2]);
        }
    }
    sum = gs_wave_sum_to_lane63(sum);
    if (gs_lane() == GS_WAVE - 1) red[threadIdx.x / GS_WAVE] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) t += red[i];
        partials[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(GS_BLOCK) void scale_reg_finalize_kernel(const float *__restrict__ partials, int n_blocks,
                                                                       const int *__restrict__ live_count,
                                                                       float *__restrict__ out) {
    __shared__ double red[GS_BLOCK];
    __shared__ int redi[GS_BLOCK / GS_WAVE];
    const int live = block_sum_int(live_count[threadIdx.x], redi);
    double a = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += GS_BLOCK) a += partials[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = GS_BLOCK / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {   // 0/0 = nan, like torch's mean over an empty selection
        out[0] = (float)(red[0] / (double)live);
        out[1] = (float)live;
    }
}

}  // namespace

extern "C" {

long long gs_scale_regulariser_workspace_floats(void) { return REG_BLOCKS + COUNT_BLOCKS; }

int gs_scale_regulariser(const float *features, const int8_t *point_invalid_mask, int n_points, float weight,
                         const float *upstream, float *grad_features, float *workspace, float *value_and_count,
                         void *stream) {
    GS_REQUIRE(n_points >= 0 && workspace && value_and_count, "gs_scale_regulariser: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int *live = reinterpret_cast<int *>(workspace + REG_BLOCKS);
    hipLaunchKernelGGL(count_live_kernel, dim3(COUNT_BLOCKS), dim3(GS_BLOCK), 0, s, point_invalid_mask, n_points, live);
    GS_CHECK_LAUNCH();
    int blocks = gs_div_up(n_points, GS_BLOCK);
    blocks = blocks < REG_BLOCKS ? blocks : REG_BLOCKS;
    if (blocks > 0) {
        hipLaunchKernelGGL(scale_reg_kernel, dim3(blocks), dim3(GS_BLOCK), 0, s, features, point_invalid_mask, n_points,
                           live, weight, upstream, grad_features, workspace);
        GS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(scale_reg_finalize_kernel, dim3(1), dim3(GS_BLOCK), 0, s, workspace, blocks, live,
                       value_and_count);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
