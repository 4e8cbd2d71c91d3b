// gs_point_backward.hip -- per-visible-point chain rule of the backward pass (gfx950).
// Replaces the per-point loop of gaussian_point_rasterisation_backward (RAS:707-772) and fuses
// the torch post-processing that follows it in the reference: SH band clearing
// (RAS:1167-1182) and the gradient factors (RAS:1105-1125).
//
// Jacobians (GP3:132-159 position, GP3:237-331 covariance, GP3:351-373 colour) are contracted
// with the upstream gradient FIRST instead of being materialised as 4x9 / 9x9 matrices:
//   Sigma' = U Sigma U^T, Sigma = M M^T, M = R(q) diag(exp s), U = J W
//   dL/dSigma = U^T g U,   dL/dM = (dL/dSigma + dL/dSigma^T) M = 2 (U^T g U) M   (g symmetric)
//   dL/ds_c = (sum_a dL/dM[a][c] R[a][c]) exp(s_c),   dL/dq = sum_ab dL/dM[a][b] dM[a][b]/dq
// which is the same linear map as GP3:270-330 evaluated in a cheaper association order.
#include "gs_common.h"
#include "gs_slots.h"

namespace {

__device__ __forceinline__ void rotmat_from_q(const float q[4], float R[9]) {  // GP3:31-48
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (yy + zz); R[1] = 2.f * (xy - wz); R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz); R[4] = 1.f - 2.f * (xx + zz); R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy); R[7] = 2.f * (yz + wx); R[8] = 1.f - 2.f * (xx + yy);
}

__device__ __forceinline__ void sh_basis(const float d[3], float Y[16]) {  // SPH:10-32
    float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float x = d[0] / n, y = d[1] / n, z = d[2] / n;
    Y[0] = 0.28209479177387814f;
    Y[1] = -0.48860251190291987f * y;
    Y[2] = 0.48860251190291987f * z;
    Y[3] = -0.48860251190291987f * x;
    Y[4] = 1.0925484305920792f * x * y;
    Y[5] = -1.0925484305920792f * y * z;
    Y[6] = 0.94617469575755997f * z * z - 0.31539156525251999f;
    Y[7] = -1.0925484305920792f * x * z;
    Y[8] = 0.54627421529603959f * x * x - 0.54627421529603959f * y * y;
    Y[9] = 0.59004358992664352f * y * (-3.0f * x * x + y * y);
    Y[10] = 2.8906114426405538f * x * y * z;
    Y[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z * z);
    Y[12] = 0.3731763325901154f * z * (5.0f * z * z - 3.0f);
    Y[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z * z);
    Y[14] = 1.4453057213202769f * z * (x * x - y * y);
    Y[15] = 0.59004358992664352f * x * (-x * x + 3.0f * y * y);
}

struct Factors { float q, s, alpha, color, color_hi; int keep; /* SH coefficients kept per channel */ };
constexpr int ZERO_ROUNDS = 8;   // rows per thread of the workgroups that zero the rows of invisible points
// Where a Gaussian's accumulator record comes from when it is not read from `acc`: the slot records of gs_blend_backward,
// summed here (gs_slots.h) -- the fused form of gs_reduce_partials + gs_point_backward: one launch less and the 48 B x M
// accumulator array is neither written nor read back.  MEASURED SLOWER at the headline size (round 5, flags fetched as
// aligned dwords: 0.158 ms against 0.053 + 0.093 ms for the two kernels; round 3: 0.186 against 0.069 + 0.101): the slot
// gather lives on waves in flight, and this kernel runs two waves per SIMD (61 KB of row staging per workgroup) where
// gs_reduce_partials runs six.  Kept as an option of the entry point (small frames, where a launch costs more than the
// gather), not the operator's default.
struct SlotSource {
    const int32_t *slot_offsets, *ntiles_full;
    const uint8_t *slot_flags;
    const float4 *partials;
    int tw, th;
};
#ifndef GS_PB_CHUNK
#define GS_PB_CHUNK 6   // slot records in flight per lane: the kernel runs two waves per SIMD (61 KB of row staging per
                        // workgroup), so memory-level parallelism has to come from the lane, and registers are plentiful
#endif

__global__ __launch_bounds__(GS_BLOCK) void point_backward_kernel(
    const float *__restrict__ xyz, const float *__restrict__ feat, const int32_t *__restrict__ obj,
    const float *__restrict__ Kmat, const float *__restrict__ q_cp, const float *__restrict__ t_cp,
    const float *__restrict__ t_pc, const int32_t *__restrict__ ids, int m, const float4 *__restrict__ acc,
    const float4 *__restrict__ attrs, const int32_t *__restrict__ ntiles_owned, Factors fac,
    float *__restrict__ grad_xyz, float *__restrict__ grad_feat, float *__restrict__ grad_xyz_vis,
    float *__restrict__ grad_feat_vis, float *__restrict__ hook_compact, SlotSource slots,
    const int8_t *__restrict__ visible_mask, int n_points, int n_zero_blocks) {
    extern __shared__ __attribute__((aligned(16))) float4 s_rows[];  // [4 waves][64 rows][GS_ROW_F4]
    if ((int)blockIdx.x < n_zero_blocks) {
        // The leading workgroups zero the rows of the dense gradients that belong to points NOT in the frustum (RAS:1051-1053
        // zero-initialises everything; the visible rows are fully written by the other workgroups): formerly a launch of its
        // own for a few per cent of the rows.
        float4 *gf4 = reinterpret_cast<float4 *>(grad_feat);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < ZERO_ROUNDS; ++r) {
            const int row = (int)blockIdx.x * (GS_BLOCK * ZERO_ROUNDS) + r * GS_BLOCK + threadIdx.x;
            if (row < n_points && visible_mask[row] == 0) {
#pragma unroll
                for (int c = 0; c < 14; ++c) gf4[(size_t)row * 14 + c] = z;
                grad_xyz[3 * (size_t)row] = 0.f; grad_xyz[3 * (size_t)row + 1] = 0.f; grad_xyz[3 * (size_t)row + 2] = 0.f;
            }
        }
        return;
    }
    const int i = ((int)blockIdx.x - n_zero_blocks) * GS_BLOCK + threadIdx.x;
    const int id = i < m ? ids[i] : -1;
    float4 *wave_rows = s_rows + (threadIdx.x >> 6) * (GS_WAVE * GS_ROW_F4);
    float4 *my_row = wave_rows + gs_lane() * GS_ROW_F4;   // this lane's gradient row, assembled in LDS
    const int ic = i < m ? i : 0, idc = i < m ? id : 0;
    // Only q, s and the opacity logit of the 224-B feature row are read (2 x 16 B): the colour chain needs
    // sigmoid'(SH . Y) = rgb (1 - rgb), and rgb was stored by the forward pass (row 2 of the packed record).
    const float4 *frow = reinterpret_cast<const float4 *>(feat) + 14 * (size_t)idc;
    const float4 f0 = frow[0], f1 = frow[1];
    const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    float4 A0, A1, A2;
    if (acc != nullptr) {
        A0 = acc[3 * (size_t)ic]; A1 = acc[3 * (size_t)ic + 1]; A2 = acc[3 * (size_t)ic + 2];
    } else {   // fused slot reduction (wave-convergent: every lane takes part in the sums of a heavy Gaussian)
        SlotSum a;
        gs_sum_slots_of_lane<GS_PB_CHUNK, 2>(i < m, ic, slots.slot_offsets, slots.ntiles_full, slots.slot_flags, slots.partials,
                                          ntiles_owned, attrs, slots.tw, slots.th, a);
        A0 = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
        A1 = make_float4(a.v[4], a.v[5], a.v[6], a.v[7]);
        A2 = make_float4(a.v[8], a.v[9], __builtin_bit_cast(float, a.npix), 0.f);
    }
    const float g_uv[2] = {A0.x, A0.y};
    const float g00 = A0.z, g01 = A0.w, g11 = A1.x;
    const float g_rgb[3] = {A1.y, A1.z, A1.w};
    const float g_logit = A2.x;

    float K[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) K[k] = Kmat[k];
    const int o = obj[idc];
    float W[9];
    rotmat_from_q(q_cp + 4 * o, W);
    const float t[3] = {t_cp[3 * o], t_cp[3 * o + 1], t_cp[3 * o + 2]};
    const float p[3] = {xyz[3 * (size_t)idc], xyz[3 * (size_t)idc + 1], xyz[3 * (size_t)idc + 2]};
    float c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) c[r] = ((W[3 * r] * p[0] + W[3 * r + 1] * p[1]) + W[3 * r + 2] * p[2]) + t[r];
    const float iz = 1.f / c[2], iz2 = iz * iz;

    // position: GP3:132-159 (full K rows 0,1), grad_xyz = g_uv @ (d_uv_d_camera @ W)
    const float dc[6] = {K[0] * iz, K[1] * iz, (-K[0] * c[0] - K[1] * c[1]) * iz2,
                         K[3] * iz, K[4] * iz, (-K[3] * c[0] - K[4] * c[1]) * iz2};
    float gc[3];  // dL/d camera-space position
#pragma unroll
    for (int k = 0; k < 3; ++k) gc[k] = g_uv[0] * dc[k] + g_uv[1] * dc[3 + k];
    float gx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) gx[k] = gc[0] * W[k] + gc[1] * W[3 + k] + gc[2] * W[6 + k];

    // covariance: U = J W with J of GP3:84-87
    const float J[6] = {K[0] * iz, 0.f, -(K[0] * c[0]) * iz2, 0.f, K[4] * iz, -(K[4] * c[1]) * iz2};
    float U[6];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) U[a * 3 + b] = J[a * 3] * W[b] + J[a * 3 + 1] * W[3 + b] + J[a * 3 + 2] * W[6 + b];
    // dL/dSigma = U^T g U  (3x3, symmetric)
    float gU[6];  // g @ U (2x3)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        gU[b] = g00 * U[b] + g01 * U[3 + b];
        gU[3 + b] = g01 * U[b] + g11 * U[3 + b];
    }
    float dS[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) dS[a * 3 + b] = U[a] * gU[b] + U[3 + a] * gU[3 + b];
    float R[9];
    rotmat_from_q(f, R);
    const float es[3] = {expf(f[4]), expf(f[5]), expf(f[6])};
    float M[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) M[a * 3 + b] = R[a * 3 + b] * es[b];
    float dM[9];  // dL/dM = 2 dS M
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            dM[a * 3 + b] = 2.f * (dS[a * 3] * M[b] + dS[a * 3 + 1] * M[3 + b] + dS[a * 3 + 2] * M[6 + b]);
    float gs[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) gs[b] = (dM[b] * R[b] + dM[3 + b] * R[3 + b] + dM[6 + b] * R[6 + b]) * es[b];
    // dM/dq rows of GP3:319-329 (M[a][b] = R[a][b] s_b), contracted with dM
    const float qx = f[0], qy = f[1], qz = f[2], qw = f[3], sx = es[0], sy = es[1], sz = es[2];
    float gq[4];
    gq[0] = dM[1] * (2 * sy * qy) + dM[2] * (2 * sz * qz) + dM[3] * (2 * sx * qy) + dM[4] * (-4 * sy * qx) +
            dM[5] * (-2 * sz * qw) + dM[6] * (2 * sx * qz) + dM[7] * (2 * sy * qw) + dM[8] * (-4 * sz * qx);
    gq[1] = dM[0] * (-4 * sx * qy) + dM[1] * (2 * sy * qx) + dM[2] * (2 * sz * qw) + dM[3] * (2 * sx * qx) +
            dM[5] * (2 * sz * qz) + dM[6] * (-2 * sx * qw) + dM[7] * (2 * sy * qz) + dM[8] * (-4 * sz * qy);
    gq[2] = dM[0] * (-4 * sx * qz) + dM[1] * (-2 * sy * qw) + dM[2] * (2 * sz * qx) + dM[3] * (2 * sx * qw) +
            dM[4] * (-4 * sy * qz) + dM[5] * (2 * sz * qy) + dM[6] * (2 * sx * qx) + dM[7] * (2 * sy * qy);
    gq[3] = dM[1] * (-2 * sy * qz) + dM[2] * (2 * sz * qy) + dM[3] * (2 * sx * qz) + dM[5] * (-2 * sz * qx) +
            dM[6] * (-2 * sx * qy) + dM[7] * (2 * sy * qx);
    my_row[0] = make_float4(gq[0] * fac.q, gq[1] * fac.q, gq[2] * fac.q, gq[3] * fac.q);
    my_row[1] = make_float4(gs[0] * fac.s, gs[1] * fac.s, gs[2] * fac.s, g_logit * fac.alpha);

    // colour: RAS:749-756; ray origin = t_pointcloud_camera of the object (RAS:731)
    const float dir[3] = {p[0] - t_pc[3 * o], p[1] - t_pc[3 * o + 1], p[2] - t_pc[3 * o + 2]};
    float Y[16];
    sh_basis(dir, Y);
    // sigmoid(SH . Y) per channel: stored by the forward pass for every Gaussian that emitted a key on this GPU;
    // a Gaussian that only reached tiles of OTHER ranks (tile-row sharding) gets its colour gradient from their
    // accumulators after the all-reduce -- only then is the SH row read and the colour re-evaluated here
    float sg[3] = {0.f, 0.f, 0.f};
    const bool any_colour_grad = g_rgb[0] != 0.f || g_rgb[1] != 0.f || g_rgb[2] != 0.f;
    if (ntiles_owned == nullptr || ntiles_owned[ic] > 0) {
        const float4 rgb = attrs[4 * (size_t)ic + 2];
        sg[0] = rgb.x; sg[1] = rgb.y; sg[2] = rgb.z;
    } else if (any_colour_grad) {
        // the same source as the forward evaluation (gs_common.h), from the same inputs: bit-identical to what the
        // ranks that did emit this Gaussian stored
        float Wf[9];
        gs_rotmat_from_q(q_cp[4 * o], q_cp[4 * o + 1], q_cp[4 * o + 2], q_cp[4 * o + 3], Wf);
        const float *coeffs = reinterpret_cast<const float *>(frow);
        gs_view_colour(Wf, t, p, [&](int ch, int k) { return coeffs[8 + 16 * ch + k]; }, sg);
    }
    // (Writing the row out in two 112-B halves through a half-sized staging area -- 8 KB instead of 15 KB of LDS per wave,
    // twice the occupancy -- was measured SLOWER: 0.140 vs 0.103 ms; partial-line stores cost more than the occupancy buys.)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        // UTL:356-359; a zero upstream gradient gives exact zeros whatever the stored colour holds
        const float scale = g_rgb[ch] != 0.f ? g_rgb[ch] * (sg[ch] * (1.f - sg[ch])) : 0.f;
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float x = scale * Y[k] * (k == 0 ? fac.color : fac.color_hi);
            v[k] = k < fac.keep ? x : 0.f;  // RAS:1167-1182
        }
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
            my_row[2 + 4 * ch + k4] = make_float4(v[4 * k4], v[4 * k4 + 1], v[4 * k4 + 2], v[4 * k4 + 3]);
    }
    // coalesced AoS scatter of the gradient rows out of the LDS staging area
    gs_rows_lds_to_global(reinterpret_cast<float4 *>(grad_feat), id, wave_rows);
    if (grad_feat_vis) gs_rows_lds_to_global(reinterpret_cast<float4 *>(grad_feat_vis), i < m ? i : -1, wave_rows);
    if (i < m) {
#pragma unroll
        for (int k = 0; k < 3; ++k) grad_xyz[3 * (size_t)id + k] = gx[k];
        if (grad_xyz_vis) {
#pragma unroll
            for (int k = 0; k < 3; ++k) grad_xyz_vis[3 * (size_t)i + k] = gx[k];
        }
        if (hook_compact) {   // the M-compact hook fields that are plain columns of acc / attrs (RAS:1130-1139), SoA planes
            const float4 rec0 = attrs[4 * (size_t)i];   // u, v, depth, opacity
            float *h = hook_compact;
            const size_t M = (size_t)m;
            h[2 * (size_t)i] = A0.x; h[2 * (size_t)i + 1] = A0.y;               // grad_viewspace      [M,2]
            h[2 * M + i] = A2.y;                                                // magnitude           [M]
            h[3 * M + i] = A2.z;                                                // num_affected_pixels [M] (int32 bits)
            h[4 * M + i] = rec0.z;                                              // point_depth         [M]
            h[5 * M + 2 * (size_t)i] = rec0.x; h[5 * M + 2 * (size_t)i + 1] = rec0.y;   // point_uv    [M,2]
        }
    }
}

}  // namespace

extern "C" int gs_point_backward(const float *xyz, const float *features, const int32_t *object_id,
                                 const float *intrinsics, const float *q_cp, const float *t_cp, const float *t_pc,
                                 const int32_t *ids, const int8_t *visible_mask, int n_visible, int n_points,
                                 const float *acc, const float *attrs, const int32_t *num_owned_tiles,
                                 int color_max_sh_band, float grad_q_factor, float grad_s_factor,
                                 float grad_alpha_factor, float grad_color_factor,
                                 float grad_high_order_color_factor, float *grad_xyz, float *grad_features,
                                 float *grad_xyz_visible, float *grad_features_visible, float *hook_compact,
                                 const int32_t *slot_offsets, const int32_t *num_overlap_tiles,
                                 const uint8_t *slot_flags, const float *partials, int width, int height,
                                 void *stream) {
    GS_REQUIRE(n_visible >= 0 && n_points >= n_visible, "sizes");
    GS_REQUIRE(n_visible == 0 || acc != nullptr ||
                   (slot_offsets != nullptr && num_overlap_tiles != nullptr && slot_flags != nullptr && partials != nullptr &&
                    width > 0 && height > 0 && width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0),
               "gs_point_backward: either acc or the slot records of gs_blend_backward (+ image size)");
    GS_REQUIRE(n_visible == 0 || attrs != nullptr, "gs_point_backward: attrs (the packed records of the forward pass) is required");
    GS_REQUIRE((reinterpret_cast<uintptr_t>(slot_flags) & 3u) == 0, "slot_flags must be 4-byte aligned (it is read as dwords)");
    hipStream_t s = (hipStream_t)stream;
    // rows of points outside the frustum: zeroed by the leading workgroups of the per-point kernel when the mask is known,
    // else both arrays are cleared first
    const bool zero_in_kernel = n_points > 0 && visible_mask != nullptr && n_visible > 0;
    if (!zero_in_kernel && n_points > 0) {
        GS_CHECK_HIP(hipMemsetAsync(grad_xyz, 0, sizeof(float) * 3 * (size_t)n_points, s));
        GS_CHECK_HIP(hipMemsetAsync(grad_features, 0, sizeof(float) * GS_FEATURE_DIM * (size_t)n_points, s));
    }
    if (n_visible == 0) return 0;
    Factors fac;
    fac.q = grad_q_factor; fac.s = grad_s_factor; fac.alpha = grad_alpha_factor;
    fac.color = grad_color_factor; fac.color_hi = grad_high_order_color_factor;
    fac.keep = color_max_sh_band <= 0 ? 1 : color_max_sh_band == 1 ? 4 : color_max_sh_band == 2 ? 9 : 16;
    const int nblk = gs_div_up(n_visible, GS_BLOCK);
    const int nzero = zero_in_kernel ? gs_div_up(n_points, GS_BLOCK * ZERO_ROUNDS) : 0;
    hipLaunchKernelGGL(point_backward_kernel, dim3(nblk + nzero), dim3(GS_BLOCK),
                       sizeof(float4) * GS_BLOCK * GS_ROW_F4, s, xyz,
                       features, object_id, intrinsics, q_cp, t_cp, t_pc, ids, n_visible,
                       reinterpret_cast<const float4 *>(acc), reinterpret_cast<const float4 *>(attrs), num_owned_tiles,
                       fac, grad_xyz, grad_features, grad_xyz_visible, grad_features_visible, hook_compact,
                       SlotSource{slot_offsets, num_overlap_tiles, slot_flags, reinterpret_cast<const float4 *>(partials),
                                  width / GS_TILE_WIDTH, height / GS_TILE_HEIGHT},
                       visible_mask, n_points, nzero);
    GS_CHECK_LAUNCH();
    return 0;
}
