// gs_controller.hip -- the two device kernels of the adaptive controller (K9, K10), gfx950.
//   gs_ellipsoid_offsets : focal vector of every Gaussian's ellipsoid      (ADC:10-25, GP3:375-388)
//   gs_sample_from_points: one draw from N(mu, R S S^T R^T) per Gaussian   (ADC:27-42, GP3:90-94,390-406)
// The reference draws its uniforms with ti.random() inside the kernel; here the caller supplies them
// (float[n][4] in (0,1]), which makes the kernel a pure function that can be checked against the oracle.
#include "gs_common.h"

namespace {

__device__ __forceinline__ void rotmat_from_q(const float *q, float R[9]) {  // GP3:31-48
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (yy + zz); R[1] = 2.f * (xy - wz); R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz); R[4] = 1.f - 2.f * (xx + zz); R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy); R[7] = 2.f * (yz + wx); R[8] = 1.f - 2.f * (xx + yy);
}

__global__ void ellipsoid_offsets_kernel(const float *__restrict__ feat, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *f = feat + (size_t)GS_FEATURE_DIM * i;
    const float sx = f[4], sy = f[5], sz = f[6];  // log-scales
    // GP3:377-381: the longest axis (compared on the log-scales), x unless y or z is strictly the largest
    int axis = 0;
    if (sx < sy && sy > sz) axis = 1;
    else if (sx < sz && sy < sz) axis = 2;
    float R[9];
    rotmat_from_q(f, R);
    const float ex = expf(sx), ey = expf(sy), ez = expf(sz);
    const float rc = fmaxf(fmaxf(ex, ey), ez), ra = fminf(fminf(ex, ey), ez);
    const float len = sqrtf(rc * rc - ra * ra);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[3 * (size_t)i + k] = len * R[3 * k + axis];
}

__global__ void sample_from_points_kernel(const float *__restrict__ xyz, const float *__restrict__ feat,
                                          const float *__restrict__ uniforms, int n, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *f = feat + (size_t)GS_FEATURE_DIM * i;
    const float4 u = reinterpret_cast<const float4 *>(uniforms)[i];
    // GP3:90-94 Box-Muller; z3 uses the cosine branch of the second pair (GP3:397)
    const float two_pi = 2.f * 3.141592653589f;
    const float r1 = sqrtf(-2.f * logf(u.x)), r2 = sqrtf(-2.f * logf(u.z));
    const float z1 = r1 * cosf(two_pi * u.y), z2 = r1 * sinf(two_pi * u.y), z3 = r2 * cosf(two_pi * u.w);
    float R[9];
    rotmat_from_q(f, R);
    const float b0 = expf(f[4]) * z1, b1 = expf(f[5]) * z2, b2 = expf(f[6]) * z3;  // S @ z
#pragma unroll
    for (int k = 0; k < 3; ++k)
        out[3 * (size_t)i + k] = xyz[3 * (size_t)i + k] + (R[3 * k] * b0 + R[3 * k + 1] * b1 + R[3 * k + 2] * b2);
}

}  // namespace

extern "C" {

int gs_ellipsoid_offsets(const float *features, int n, float *offsets, void *stream) {
    GS_REQUIRE(n >= 0, "n");
    if (n == 0) return 0;
    hipLaunchKernelGGL(ellipsoid_offsets_kernel, dim3(gs_div_up(n, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       features, n, offsets);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_sample_from_points(const float *xyz, const float *features, const float *uniforms, int n, float *samples,
                          void *stream) {
    GS_REQUIRE(n >= 0, "n");
    if (n == 0) return 0;
    hipLaunchKernelGGL(sample_from_points_kernel, dim3(gs_div_up(n, GS_BLOCK)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, xyz, features, uniforms, n, samples);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"

// Per-iteration statistics of the adaptive controller (GaussianPointAdaptiveController.py:130-146): six indexed
// "+=" over the M visible Gaussians, which eager PyTorch runs as ~20 gather/scatter kernels (~0.15 ms at M = 1e6) on
// every training iteration.  The visible ids are unique, so plain read-modify-write is exact; one pass, ~50 MB.
namespace {

__global__ void controller_accumulate_kernel(const int32_t *__restrict__ ids, const int32_t *__restrict__ pixels,
                                             const float *__restrict__ magnitude, const float *__restrict__ grad_xyz,
                                             int m, int32_t *__restrict__ num_in_camera,
                                             int32_t *__restrict__ num_pixels, float *__restrict__ view_grad,
                                             float *__restrict__ view_grad_avg, float *__restrict__ pos_grad,
                                             float *__restrict__ pos_grad_norm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int id = ids[i];
    const int px = pixels[i];
    const float mag = magnitude[i];
    const float gx = grad_xyz[3 * (size_t)i], gy = grad_xyz[3 * (size_t)i + 1], gz = grad_xyz[3 * (size_t)i + 2];
    num_in_camera[id] += 1;
    num_pixels[id] += px;
    view_grad[id] += mag;
    const float avg = mag / (float)px;                 // 0/0 -> NaN -> 0 (ADC:139-140)
    view_grad_avg[id] += (avg != avg) ? 0.f : avg;
    pos_grad[3 * (size_t)id] += gx;
    pos_grad[3 * (size_t)id + 1] += gy;
    pos_grad[3 * (size_t)id + 2] += gz;
    pos_grad_norm[id] += sqrtf(gx * gx + gy * gy + gz * gz);
}

}  // namespace

extern "C" int gs_controller_accumulate(const int32_t *ids, const int32_t *num_affected_pixels,
                                        const float *magnitude_grad_viewspace, const float *grad_point_in_camera,
                                        int n_visible, int32_t *accumulated_num_in_camera,
                                        int32_t *accumulated_num_pixels, float *accumulated_view_space_gradients,
                                        float *accumulated_view_space_gradients_avg,
                                        float *accumulated_position_gradients,
                                        float *accumulated_position_gradients_norm, void *stream) {
    GS_REQUIRE(n_visible >= 0, "n_visible");
    if (n_visible == 0) return 0;
    hipLaunchKernelGGL(controller_accumulate_kernel, dim3(gs_div_up(n_visible, GS_BLOCK)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, ids, num_affected_pixels, magnitude_grad_viewspace, grad_point_in_camera,
                       n_visible, accumulated_num_in_camera, accumulated_num_pixels, accumulated_view_space_gradients,
                       accumulated_view_space_gradients_avg, accumulated_position_gradients,
                       accumulated_position_gradients_norm);
    GS_CHECK_LAUNCH();
    return 0;
}
