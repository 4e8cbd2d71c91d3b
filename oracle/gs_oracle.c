/*
 * gs_oracle.c -- CPU restatement of the reference rasteriser hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed
 * "reference CPU path" (bench.py cpu_baseline, kind "port").  Nothing in the
 * product package may import, link or call it; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() do, and only as the checker.
 *
 * PARITY STATUS: PINNED AGAINST THE REFERENCE'S OWN CODE, EXECUTED.  Taichi is
 * absent from this environment (and the reference's blend kernels are
 * CUDA-only), so the reference cannot run natively; instead its unmodified
 * sources are executed from /root/reference under a small NumPy emulation of
 * the Taichi subset they use (tests/golden/taichi_emulation.py: fp32 scalars,
 * pass-by-value matrices, the two tile kernels run block by block on 256 OS
 * threads with real barriers).  tests/golden/make_reference_operator_vectors.py
 * runs the reference's whole operator -- seven kernels, torch glue, autograd
 * Function, backward hook -- on five tiny tie-free scenes and commits inputs
 * and outputs; tests/test_reference_operator.py holds this oracle to them:
 * image L-inf 1.2e-7..1.8e-7, gradients 2e-7..1e-6 relative L2, visible ids,
 * tile counts, per-pixel counts and affected-pixel counts identical (fp32
 * build).  The same vectors gate the HIP path on the GPU.  In addition:
 *  - the reference's pure-PyTorch comparator, SH basis and SE(3) helpers are
 *    executed behind a Taichi stub (tests/golden/make_reference_vectors.py,
 *    tests/test_reference_vectors.py);
 *  - every known-answer vector of the reference's own tests is restated in
 *    tests/test_oracle_pins.py (tile ranges, single-Gaussian alpha/gradients,
 *    2x2 covariance vs scipy, quaternion->R vs scipy, SE(3) inverse vs numpy).
 * What stays unpinned: the order of tied sort keys (the reference's torch.sort
 * leaves it undefined, RAS:947; this file uses the stable order) and whatever
 * Taichi's code generator does differently from IEEE fp32 without contraction
 * (fast-math, FMA) -- neither can be observed without Taichi on a GPU.
 *
 * Every function cites the reference file:line it follows.  Abbreviations:
 *   RAS = taichi_3d_gaussian_splatting/GaussianPointCloudRasterisation.py
 *   GP3 = taichi_3d_gaussian_splatting/GaussianPoint3D.py
 *   SPH = taichi_3d_gaussian_splatting/SphericalHarmonics.py
 *   UTL = taichi_3d_gaussian_splatting/utils.py
 *
 * Semantics: sequential per pixel (the block-shared staging of the reference
 * is a performance device only), stable sort on (tile, quantised depth) keys.
 * Arithmetic: `real` = float (default) or double (-DGS_F64, the "spec" build
 * used to flag fragile pixels).  Compile with -ffp-contract=off so that the
 * fp32 evaluation order written here is the one executed.
 *
 * The scatter-add accumulators of the backward pixel loop (fp32 atomics with
 * undefined order in the reference, RAS:674-696) are summed in double and
 * rounded once, which makes the oracle deterministic under OpenMP.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef GS_F64
typedef double real;
#define R_EXP exp
#define R_EXP_SCALE exp
#define R_SQRT sqrt
#define R_FLOOR floor
#define R_FABS fabs
#else
typedef float real;
/* ONE DEFINITION OF exp ON EVERY SIDE (round 5): evaluated in double and rounded once, i.e. the correctly rounded fp32
 * exponential (up to double rounding, probability 2^-29) -- here, in the emulated run of the reference's sources that
 * produced tests/golden/reference_operator_*_exp_cr.npz (GS_EMU_EXP=cr) and in the HIP library wherever an exponential
 * decides something discrete: the scale activation (GP3:175-178: covariance -> radius -> tile box -> counts, keys, slots),
 * the opacity sigmoid (RAS:299-300: an input of every alpha >= 1/255 decision) and the exact re-evaluation of a Gaussian
 * weight next to a threshold (csrc/gs_common.h, gs_alpha_reference_*).  Two libms' expf differ in the last bit on a few
 * per cent of the inputs (glibc's is correctly rounded on 99.93 %); a radius one ulp apart moves a tile-box edge across a
 * tile boundary once in a few thousand frames (fuzz case 6102), an alpha one ulp apart flips a skip decision once in a few
 * frames of a million Gaussians. */
#define R_EXP(x) ((float)exp((double)(x)))
#define R_EXP_SCALE(x) ((float)exp((double)(x)))
#define R_SQRT sqrtf
#define R_FLOOR floorf
#define R_FABS fabsf
#endif

#define TILE_W 16
#define TILE_H 16
#define BOUNDARY_TILES 3 /* RAS:26-28 */
#define RC(x) ((real)(x))

int gs_oracle_sizeof_real(void) { return (int)sizeof(real); }
int gs_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- helpers */

/* GP3:31-48 rotation_matrix_from_quaternion, q = (x,y,z,w), not normalised */
static void rotmat_from_q(const real q[4], real R[9]) {
    real x = q[0], y = q[1], z = q[2], w = q[3];
    real xx = x * x, yy = y * y, zz = z * z;
    real xy = x * y, xz = x * z, yz = y * z;
    real wx = w * x, wy = w * y, wz = w * z;
    R[0] = RC(1) - RC(2) * (yy + zz); R[1] = RC(2) * (xy - wz); R[2] = RC(2) * (xz + wy);
    R[3] = RC(2) * (xy + wz); R[4] = RC(1) - RC(2) * (xx + zz); R[5] = RC(2) * (yz - wx);
    R[6] = RC(2) * (xz - wy); R[7] = RC(2) * (yz + wx); R[8] = RC(1) - RC(2) * (xx + yy);
}

/* GP3:14-27 project_point_to_camera: c = T @ (p,1); uv = (K @ c)/c.z, full 3x3 K */
static void project_point(const real R[9], const real t[3], const real K[9],
                          const real p[3], real uv[2], real c[3]) {
    for (int i = 0; i < 3; ++i)
        c[i] = ((R[3 * i + 0] * p[0] + R[3 * i + 1] * p[1]) + R[3 * i + 2] * p[2]) + t[i] * RC(1);
    real u1 = (K[0] * c[0] + K[1] * c[1]) + K[2] * c[2];
    real v1 = (K[3] * c[0] + K[4] * c[1]) + K[5] * c[2];
    uv[0] = u1 / c[2];
    uv[1] = v1 / c[2];
}

/* small dense helpers, C = A(m x k) @ B(k x n), sums left to right like Taichi's unrolled matmul */
static void matmul(const real *A, const real *B, real *C, int m, int k, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            real s = A[i * k] * B[j];
            for (int l = 1; l < k; ++l) s = s + A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void transpose(const real *A, real *At, int m, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) At[j * m + i] = A[i * n + j];
}

/* UTL:396-432 inverse_SE3_qt_torch: q_inv = conj(q) (not renormalised);
 * t_inv = -rot(normalise(q_inv), t) with the Hamilton products of UTL:402-412 */
static void quat_mul(const real a[4], const real b[4], real o[4]) {
    real x0 = a[0], y0 = a[1], z0 = a[2], w0 = a[3];
    real x1 = b[0], y1 = b[1], z1 = b[2], w1 = b[3];
    o[0] = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1;
    o[1] = w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1;
    o[2] = w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1;
    o[3] = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
}
void gs_oracle_inverse_se3_qt(const real *q, const real *t, real *q_inv, real *t_inv, int n) {
    for (int i = 0; i < n; ++i) {
        real qi[4] = {-q[4 * i], -q[4 * i + 1], -q[4 * i + 2], q[4 * i + 3]};
        memcpy(q_inv + 4 * i, qi, sizeof qi);
        real nrm = R_SQRT(((qi[0] * qi[0] + qi[1] * qi[1]) + qi[2] * qi[2]) + qi[3] * qi[3]);
        real qn[4] = {qi[0] / nrm, qi[1] / nrm, qi[2] / nrm, qi[3] / nrm};
        real v[4] = {t[3 * i], t[3 * i + 1], t[3 * i + 2], RC(0)};
        real qc[4] = {-qn[0], -qn[1], -qn[2], qn[3]};
        real tmp[4], out[4];
        quat_mul(qn, v, tmp);
        quat_mul(tmp, qc, out);
        t_inv[3 * i] = -out[0]; t_inv[3 * i + 1] = -out[1]; t_inv[3 * i + 2] = -out[2];
    }
}

/* exported for the pin tests (GP3:31-48) */
void gs_oracle_rotation_matrix_from_quaternion(const real *q, real *R) { rotmat_from_q(q, R); }

/* ------------------------------------------------------------ K1: filter */
/* RAS:31-78 filter_point_in_camera */
void gs_oracle_filter(const real *xyz, const int8_t *invalid, const int32_t *obj,
                      const real *K, const real *q_cp, const real *t_cp, int n,
                      real near_plane, real far_plane, int width, int height,
                      int8_t *mask) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        if (invalid[i] == 1) { mask[i] = 0; continue; }
        real R[9], uv[2], c[3];
        rotmat_from_q(q_cp + 4 * obj[i], R);
        project_point(R, t_cp + 3 * obj[i], K, xyz + 3 * i, uv, c);
        int ok = c[2] > near_plane && c[2] < far_plane &&
                 uv[0] >= RC(-TILE_W * BOUNDARY_TILES) && uv[0] < RC(width + TILE_W * BOUNDARY_TILES) &&
                 uv[1] >= RC(-TILE_H * BOUNDARY_TILES) && uv[1] < RC(height + TILE_H * BOUNDARY_TILES);
        mask[i] = ok ? 1 : 0;
    }
}

/* ------------------------------------------- covariance and its Jacobians */
/* GP3:65-87 get_projective_transform_jacobian */
static void proj_jacobian(const real K[9], const real c[3], real J[6]) {
    real fx = K[0], fy = K[4], x = c[0], y = c[1], z = c[2];
    J[0] = fx / z; J[1] = RC(0); J[2] = -(fx * x) / (z * z);
    J[3] = RC(0); J[4] = fy / z; J[5] = -(fy * y) / (z * z);
}

/* GP3:161-191 project_to_camera_covariance: cov = J W Sigma W^T J^T, left to right */
static void project_covariance(const real q[4], const real s[3], const real W[9],
                               const real K[9], const real c[3], real cov[4]) {
    real J[6], R[9], S[9] = {0}, Rt[9], Wt[9], Jt[6];
    proj_jacobian(K, c, J);
    rotmat_from_q(q, R);
    S[0] = R_EXP_SCALE(s[0]); S[4] = R_EXP_SCALE(s[1]); S[8] = R_EXP_SCALE(s[2]);
    real RS[9], RSS[9], Sigma[9];
    matmul(R, S, RS, 3, 3, 3);
    matmul(RS, S, RSS, 3, 3, 3); /* S^T == S */
    transpose(R, Rt, 3, 3);
    matmul(RSS, Rt, Sigma, 3, 3, 3);
    real JW[6], JWS[6], JWSW[6];
    matmul(J, W, JW, 2, 3, 3);
    matmul(JW, Sigma, JWS, 2, 3, 3);
    transpose(W, Wt, 3, 3);
    matmul(JWS, Wt, JWSW, 2, 3, 3);
    transpose(J, Jt, 2, 3);
    matmul(JWSW, Jt, cov, 2, 3, 2);
}
void gs_oracle_project_covariance(const real *q, const real *s, const real *W, const real *K,
                                  const real *c, real *cov) {
    project_covariance(q, s, W, K, c, cov);
}

/* SPH:10-32 get_spherical_harmonic_from_xyz */
static void sh_basis(const real d_in[3], real Y[16]) {
    real n = R_SQRT((d_in[0] * d_in[0] + d_in[1] * d_in[1]) + d_in[2] * d_in[2]);
    real x = d_in[0] / n, y = d_in[1] / n, z = d_in[2] / n;
    Y[0] = RC(0.28209479177387814);
    Y[1] = RC(-0.48860251190291987) * y;
    Y[2] = RC(0.48860251190291987) * z;
    Y[3] = RC(-0.48860251190291987) * x;
    Y[4] = RC(1.0925484305920792) * x * y;
    Y[5] = RC(-1.0925484305920792) * y * z;
    Y[6] = RC(0.94617469575755997) * z * z - RC(0.31539156525251999);
    Y[7] = RC(-1.0925484305920792) * x * z;
    Y[8] = RC(0.54627421529603959) * x * x - RC(0.54627421529603959) * y * y;
    Y[9] = RC(0.59004358992664352) * y * (RC(-3.0) * x * x + y * y);
    Y[10] = RC(2.8906114426405538) * x * y * z;
    Y[11] = RC(0.45704579946446572) * y * (RC(1.0) - RC(5.0) * z * z);
    Y[12] = RC(0.3731763325901154) * z * (RC(5.0) * z * z - RC(3.0));
    Y[13] = RC(0.45704579946446572) * x * (RC(1.0) - RC(5.0) * z * z);
    Y[14] = RC(1.4453057213202769) * z * (x * x - y * y);
    Y[15] = RC(0.59004358992664352) * x * (-x * x + RC(3.0) * y * y);
}
void gs_oracle_sh_basis(const real *d, real *Y) { sh_basis(d, Y); }

static real sigmoid(real x) { return RC(1) / (RC(1) + R_EXP(-x)); } /* UTL:351-353 */

/* --------------------------------------------- K2: per-visible-point pass */
/* RAS:239-315 generate_point_attributes_in_camera_plane (+ RAS:196-205 in-place
 * q normalisation, RAS:208-236 row layout, UTL:257-272 conic, GP3:333-349 colour) */
void gs_oracle_preprocess(const real *xyz, real *feat /* [N,56], q cols rewritten */,
                          const int32_t *obj, const real *K, const real *q_cp,
                          const real *t_cp, const int32_t *ids, int m,
                          real *uv_out, real *xyz_cam, real *conic, real *alpha_out,
                          real *rgb, real *radii) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) {
        int id = ids[i];
        real *f = feat + (size_t)56 * id;
        /* RAS:196-205: q <- q/|q| written back */
        real nrm = R_SQRT(((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]) + f[3] * f[3]);
        f[0] = f[0] / nrm; f[1] = f[1] / nrm; f[2] = f[2] / nrm; f[3] = f[3] / nrm;
        const real *p = xyz + 3 * id;
        real W[9], uv[2], c[3], cov[4];
        const real *t = t_cp + 3 * obj[id];
        rotmat_from_q(q_cp + 4 * obj[id], W);
        /* UTL:495-510 taichi_inverse_SE3: ray origin = (-R^T) t */
        real ro[3];
        for (int k = 0; k < 3; ++k)
            ro[k] = ((-W[0 + k]) * t[0] + (-W[3 + k]) * t[1]) + (-W[6 + k]) * t[2];
        project_point(W, t, K, p, uv, c);
        project_covariance(f, f + 4, W, K, c, cov);
        /* UTL:257-272 get_point_conic_and_rescale (operates on a by-value copy) */
        real det0 = cov[0] * cov[3] - cov[1] * cov[2];
        real a = cov[0] + RC(0.3), d = cov[3] + RC(0.3);
        real det = a * d - cov[1] * cov[2];
        real ratio = det0 / det;
        real rescale = R_SQRT(ratio > RC(0) ? ratio : RC(0));
        real inv = RC(1.0) / det;
        uv_out[2 * i] = uv[0]; uv_out[2 * i + 1] = uv[1];
        xyz_cam[3 * i] = c[0]; xyz_cam[3 * i + 1] = c[1]; xyz_cam[3 * i + 2] = c[2];
        conic[4 * i + 0] = inv * d;
        conic[4 * i + 1] = inv * (-cov[1]);
        conic[4 * i + 2] = inv * a;
        conic[4 * i + 3] = rescale;
        alpha_out[i] = RC(1.) / (RC(1.) + R_EXP(-f[7])); /* RAS:299-300 */
        real dir[3] = {p[0] - ro[0], p[1] - ro[1], p[2] - ro[2]}, Y[16];
        sh_basis(dir, Y);
        for (int ch = 0; ch < 3; ++ch) {
            const real *cf = f + 8 + 16 * ch;
            real s = cf[0] * Y[0];
            for (int k = 1; k < 16; ++k) s = s + cf[k] * Y[k];
            rgb[3 * i + ch] = sigmoid(s);
        }
        /* RAS:311-315 radius from the UN-filtered covariance */
        real dd = cov[0] - cov[3];
        real lam = (cov[0] + cov[3] + R_SQRT(dd * dd + RC(4.0) * cov[1] * cov[2])) / RC(2.0);
        radii[i] = R_SQRT(lam) * RC(3.0);
    }
}

/* --------------------------------------------------- K3/K4: tile binning */
/* RAS:81-103 get_bounding_box_by_point_and_radii */
static void tile_box(real u, real v, real r, int width, int height, int box[4]) {
    r = r > RC(1.0) ? r : RC(1.0);
    real min_u = (u - r) > RC(0.0) ? (u - r) : RC(0.0), max_u = u + r;
    real min_v = (v - r) > RC(0.0) ? (v - r) : RC(0.0), max_v = v + r;
    int tw = width / TILE_W, th = height / TILE_H;
    int t0u = (int)R_FLOOR(min_u / RC(TILE_W)); if (t0u > tw) t0u = tw;
    int t1u = (int)R_FLOOR(max_u / RC(TILE_W)) + 1;
    if (t1u < t0u + 1) t1u = t0u + 1; if (t1u > tw) t1u = tw;
    int t0v = (int)R_FLOOR(min_v / RC(TILE_H)); if (t0v > th) t0v = th;
    int t1v = (int)R_FLOOR(max_v / RC(TILE_H)) + 1;
    if (t1v < t0v + 1) t1v = t0v + 1; if (t1v > th) t1v = th;
    box[0] = t0u; box[1] = t1u; box[2] = t0v; box[3] = t1v;
}
/* RAS:106-128 generate_num_overlap_tiles */
void gs_oracle_num_overlap_tiles(const real *uv, const real *radii, int m, int width, int height,
                                 int32_t *count) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) {
        int b[4];
        tile_box(uv[2 * i], uv[2 * i + 1], radii[i], width, height, b);
        count[i] = (b[1] - b[0]) * (b[3] - b[2]);
    }
}
/* RAS:131-172 generate_point_sort_key_by_num_overlap_tiles; `offsets` is the
 * exclusive scan of the counts (RAS:913-922) */
void gs_oracle_make_keys(const real *uv, const real *xyz_cam, const real *radii,
                         const int64_t *offsets, int m, int width, int height, real depth_scale,
                         int64_t *keys, int32_t *payload) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) {
        int b[4];
        tile_box(uv[2 * i], uv[2 * i + 1], radii[i], width, height, b);
        int32_t dq = (int32_t)(xyz_cam[3 * i + 2] * depth_scale); /* trunc toward 0 */
        for (int tu = b[0]; tu < b[1]; ++tu)
            for (int tv = b[2]; tv < b[3]; ++tv) {
                int64_t k = offsets[i] + (int64_t)(b[3] - b[2]) * (tu - b[0]) + (tv - b[2]);
                int32_t tile = tu + tv * (width / TILE_W);
                keys[k] = (int64_t)dq + (((int64_t)tile) << 32);
                payload[k] = i;
            }
    }
}

/* RAS:947-950 sort + gather.  torch.sort is not index-stable; the contract
 * fixed for this project is the STABLE order (ties keep generation order).
 * LSD radix sort, 8 passes x 8 bits over the biased 64-bit key. */
void gs_oracle_sort_pairs(int64_t *keys, int32_t *payload, int64_t n) {
    if (n <= 1) return;
    uint64_t *k0 = (uint64_t *)malloc(sizeof(uint64_t) * n), *k1 = (uint64_t *)malloc(sizeof(uint64_t) * n);
    int32_t *p0 = (int32_t *)malloc(sizeof(int32_t) * n), *p1 = (int32_t *)malloc(sizeof(int32_t) * n);
    for (int64_t i = 0; i < n; ++i) { k0[i] = (uint64_t)keys[i] ^ 0x8000000000000000ull; p0[i] = payload[i]; }
    for (int pass = 0; pass < 8; ++pass) {
        int64_t hist[257] = {0};
        int sh = 8 * pass;
        for (int64_t i = 0; i < n; ++i) hist[((k0[i] >> sh) & 255) + 1]++;
        if (hist[((k0[0] >> sh) & 255) + 1] == n) continue; /* all equal digit */
        for (int d = 0; d < 256; ++d) hist[d + 1] += hist[d];
        for (int64_t i = 0; i < n; ++i) {
            int64_t dst = hist[(k0[i] >> sh) & 255]++;
            k1[dst] = k0[i]; p1[dst] = p0[i];
        }
        uint64_t *tk = k0; k0 = k1; k1 = tk;
        int32_t *tp = p0; p0 = p1; p1 = tp;
    }
    for (int64_t i = 0; i < n; ++i) { keys[i] = (int64_t)(k0[i] ^ 0x8000000000000000ull); payload[i] = p0[i]; }
    free(k0); free(k1); free(p0); free(p1);
}

/* RAS:175-193 find_tile_start_and_end; start/end must be pre-zeroed (RAS:954-957) */
void gs_oracle_tile_ranges(const int64_t *keys, int64_t n, int32_t *start, int32_t *end) {
    if (n <= 0) return;
    for (int64_t i = 0; i + 1 < n; ++i) {
        int32_t t = (int32_t)(keys[i] >> 32), tn = (int32_t)(keys[i + 1] >> 32);
        if (t != tn) { start[tn] = (int32_t)(i + 1); end[t] = (int32_t)(i + 1); }
    }
    end[(int32_t)(keys[n - 1] >> 32)] = (int32_t)n;
}

/* ----------------------------------------------------- K6: forward blend */
/* RAS:318-485 gaussian_point_rasterisation, weight UTL:275-284.
 * `margin` (optional, may be NULL): per pixel, the smallest distance of any
 * evaluated alpha to the 1/255 skip threshold and of any T' to the 1e-4 stop
 * threshold -- used by the f64 spec build to flag fragile pixels. */
void gs_oracle_blend_forward(int height, int width, const int32_t *tile_start, const int32_t *tile_end,
                             const int32_t *payload, const real *uv, const real *xyz_cam,
                             const real *conic, const real *alpha_pt, const real *rgb,
                             real *image, real *depth, real *acc_alpha, int32_t *last_eff,
                             int32_t *count, int rgb_only, real *margin) {
    int tw = width / TILE_W;
    const real eps_alpha = (real)(1. / 255.), stop_T = (real)0.0001, clamp = (real)0.99;
#pragma omp parallel for schedule(dynamic, 4)
    for (int pix = 0; pix < height * width; ++pix) {
        /* RAS:348-358: the loop index is tile-major */
        int tile = pix / (TILE_W * TILE_H), in_tile = pix % (TILE_W * TILE_H);
        int tu = tile % tw, tv = tile / tw;
        int pu = tu * TILE_W + in_tile % TILE_W, pv = tv * TILE_H + in_tile / TILE_W;
        int start = tile_start[tile], end = tile_end[tile];
        real T = RC(1.0), C[3] = {0, 0, 0}, D = RC(0.), Wd = RC(0.);
        int last = start, cnt = 0;
        real mg = RC(1e30);
        real px = (real)pu + RC(0.5), py = (real)pv + RC(0.5);
        for (int j = start; j < end; ++j) {
            int o = payload[j];
            real dx = px - uv[2 * o], dy = py - uv[2 * o + 1];
            const real *cn = conic + 4 * o;
            /* UTL:275-284 */
            real e = RC(-0.5) * (dx * dx * cn[0] + dy * dy * cn[2]) - dx * dy * cn[1];
            real g = R_EXP(e) * cn[3];
            real a = g * alpha_pt[o];
            if (margin && R_FABS(a - eps_alpha) < mg) mg = R_FABS(a - eps_alpha);
            if (a < eps_alpha) continue;
            a = a < clamp ? a : clamp;
            real Tn = T * (RC(1) - a);
            if (margin && R_FABS(Tn - stop_T) < mg) mg = R_FABS(Tn - stop_T);
            if (Tn < stop_T) break; /* RAS:458-460 saturated: this Gaussian is NOT blended */
            last = j + 1;
            C[0] += rgb[3 * o] * a * T; C[1] += rgb[3 * o + 1] * a * T; C[2] += rgb[3 * o + 2] * a * T;
            if (!rgb_only) { D += xyz_cam[3 * o + 2] * a * T; Wd += a * T; cnt += 1; }
            T = Tn;
        }
        size_t p = (size_t)pv * width + pu;
        image[3 * p] = C[0]; image[3 * p + 1] = C[1]; image[3 * p + 2] = C[2];
        if (!rgb_only) {
            depth[p] = D / (Wd > RC(1e-6) ? Wd : RC(1e-6));
            acc_alpha[p] = RC(1.) - T;
            last_eff[p] = last;
            count[p] = cnt;
        }
        if (margin) margin[p] = mg;
    }
}

/* ------------------------------------------- K7: backward per-pixel pass */
/* RAS:531-705, gradients UTL:331-348.  Accumulators (all M-indexed here; the
 * reference indexes grad_uv / logit / magnitude by point id, which is the
 * same set of rows gathered by point_id_in_camera_list, RAS:1128-1140):
 *   acc[o*10 + 0..1] dL/duv, 2..4 dL/dcov (00,01,11), 5..7 dL/drgb,
 *   8 dL/dlogit, 9 sum |dL/duv| ; npix[o] = number of affected pixels. */
void gs_oracle_blend_backward(int height, int width, const int32_t *tile_start, const int32_t *tile_end,
                              const int32_t *payload, const real *uv, const real *conic,
                              const real *alpha_pt, const real *rgb, const real *grad_image,
                              const real *acc_alpha, const int32_t *last_eff, int m,
                              real *acc_out /* [M,10] */, int32_t *npix /* [M] */,
                              real *mag_image /* [H,W,2] */) {
    /* The reference scatters eleven atomics per contributing (pixel, Gaussian) pair (RAS:674-696).  On a CPU those
     * atomics would be the whole run time, so -- as a fair CPU implementation would -- a tile first sums its pairs
     * into a private record per list entry (one thread owns a tile, its pixels run in order), in double, and the
     * records of a Gaussian are added up afterwards in ascending list position.  No atomics, deterministic. */
    int tw = width / TILE_W, th = height / TILE_H;
    const real eps_alpha = (real)(1. / 255.), clamp = (real)0.99;
    long long n_keys = 0;
    for (int t = 0; t < tw * th; ++t)
        if (tile_end[t] > n_keys) n_keys = tile_end[t];
    double *part = (double *)calloc((size_t)(n_keys > 0 ? n_keys : 1) * 10, sizeof(double));
    int32_t *part_n = (int32_t *)calloc((size_t)(n_keys > 0 ? n_keys : 1), sizeof(int32_t));
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < tw * th; ++tile) {
        int tu = tile % tw, tv = tile / tw;
        int start = tile_start[tile], end = tile_end[tile];
        for (int in_tile = 0; in_tile < TILE_W * TILE_H; ++in_tile) {
            int pu = tu * TILE_W + in_tile % TILE_W, pv = tv * TILE_H + in_tile / TILE_W;
            size_t p = (size_t)pv * width + pu;
            int last = last_eff[p];
            real T = RC(1.0) - acc_alpha[p];
            real w[3] = {0, 0, 0};
            real G[3] = {grad_image[3 * p], grad_image[3 * p + 1], grad_image[3 * p + 2]};
            real mag[2] = {0, 0};
            real px = (real)pu + RC(0.5), py = (real)pv + RC(0.5);
            for (int j = (last < end ? last : end) - 1; j >= start; --j) { /* entries at or beyond `last` are skipped, RAS:618 */
                int o = payload[j];
                real dx = px - uv[2 * o], dy = py - uv[2 * o + 1];
                const real *cn = conic + 4 * o;
                /* UTL:331-348: m = inv_cov @ d, exponent = -0.5 * d.m */
                real m0 = cn[0] * dx + cn[1] * dy, m1 = cn[1] * dx + cn[2] * dy;
                real e = RC(-0.5) * (dx * m0 + dy * m1);
                real g = R_EXP(e) * cn[3];
                real a_pt = alpha_pt[o];
                real pa = g * a_pt;
                if (pa >= eps_alpha) {
                    real a = pa < clamp ? pa : clamp;
                    T = T / (RC(1.) - a);
                    real aT = a * T;
                    real gc[3] = {aT * G[0], aT * G[1], aT * G[2]};
                    const real *c = rgb + 3 * o;
                    real one_m = RC(1.) - a;
                    real dLda = ((c[0] * T - w[0] / one_m) * G[0] + (c[1] * T - w[1] / one_m) * G[1]) +
                                (c[2] * T - w[2] / one_m) * G[2];
                    w[0] += c[0] * a * T; w[1] += c[1] * a * T; w[2] += c[2] * a * T;
                    real dlogit = dLda * g * (RC(1.) - a_pt) * a_pt;
                    real dLdg = dLda * a_pt;
                    real v0 = dLdg * (g * m0), v1 = dLdg * (g * m1); /* dg/dmu = g * m */
                    mag[0] += R_FABS(v0); mag[1] += R_FABS(v1);
                    /* dg/dcov = 0.5 g (m m^T) */
                    real c00 = dLdg * (RC(0.5) * g * (m0 * m0));
                    real c01 = dLdg * (RC(0.5) * g * (m0 * m1));
                    real c11 = dLdg * (RC(0.5) * g * (m1 * m1));
                    real nv = R_SQRT(v0 * v0 + v1 * v1);
                    double *A = part + (size_t)10 * j;
                    A[0] += v0; A[1] += v1; A[2] += c00; A[3] += c01; A[4] += c11;
                    A[5] += gc[0]; A[6] += gc[1]; A[7] += gc[2]; A[8] += dlogit; A[9] += nv;
                    part_n[j] += 1;
                }
            }
            mag_image[2 * p] = mag[0]; mag_image[2 * p + 1] = mag[1];
        }
    }
    /* per-Gaussian sums in ascending list position; the ten components are independent (one thread each) */
    double *acc = (double *)calloc((size_t)(m > 0 ? m : 1) * 10, sizeof(double));
    memset(npix, 0, sizeof(int32_t) * (size_t)m);
#pragma omp parallel for schedule(static, 1)
    for (int k = 0; k < 11; ++k) {
        if (k == 10) {
            for (long long j = 0; j < n_keys; ++j) npix[payload[j]] += part_n[j];
        } else {
            for (long long j = 0; j < n_keys; ++j)
                if (part_n[j]) acc[(size_t)10 * payload[j] + k] += part[(size_t)10 * j + k];
        }
    }
    for (size_t i = 0; i < (size_t)m * 10; ++i) acc_out[i] = (real)acc[i];
    free(acc); free(part); free(part_n);
}

/* ------------------------------------------- K8: backward per-point pass */
/* RAS:707-772; Jacobians GP3:132-159 (position), GP3:237-331 (covariance),
 * GP3:351-373 + SPH:47-53 (colour).  Writes (not accumulates) rows of the
 * dense gradients; column 7 of grad_feat is written from acc[...,8]. */
void gs_oracle_point_backward(const real *xyz, const real *feat, const int32_t *obj, const real *K,
                              const real *q_cp, const real *t_cp, const real *t_pc /* ray origins, RAS:731 */,
                              const int32_t *ids, int m, const real *xyz_cam, const real *acc /* [M,10] */,
                              real *grad_xyz /* [N,3] pre-zeroed */, real *grad_feat /* [N,56] pre-zeroed */) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) {
        int id = ids[i];
        const real *f = feat + (size_t)56 * id, *p = xyz + 3 * id;
        const real *A = acc + (size_t)10 * i;
        real W[9];
        rotmat_from_q(q_cp + 4 * obj[id], W);
        const real *t = t_cp + 3 * obj[id];
        /* GP3:132-159: t = T @ (p,1) recomputed, full K rows 0 and 1 */
        real tc[3];
        for (int r = 0; r < 3; ++r)
            tc[r] = ((W[3 * r] * p[0] + W[3 * r + 1] * p[1]) + W[3 * r + 2] * p[2]) + t[r] * RC(1);
        real dc[6] = {K[0] / tc[2], K[1] / tc[2], (-K[0] * tc[0] - K[1] * tc[1]) / (tc[2] * tc[2]),
                      K[3] / tc[2], K[4] / tc[2], (-K[3] * tc[0] - K[4] * tc[1]) / (tc[2] * tc[2])};
        real duv_dp[6];
        matmul(dc, W, duv_dp, 2, 3, 3);
        real guv[2] = {A[0], A[1]};
        for (int k = 0; k < 3; ++k) grad_xyz[3 * id + k] = guv[0] * duv_dp[k] + guv[1] * duv_dp[3 + k];

        /* GP3:237-331 project_to_camera_covariance_jacobian with the SAVED xyz_cam */
        real J[6], R[9], U[6], M[9];
        proj_jacobian(K, xyz_cam + 3 * i, J);
        rotmat_from_q(f, R);
        real es[3] = {R_EXP(f[4]), R_EXP(f[5]), R_EXP(f[6])};
        real S[9] = {es[0], 0, 0, 0, es[1], 0, 0, 0, es[2]};
        matmul(R, S, M, 3, 3, 3);
        matmul(J, W, U, 2, 3, 3);
        real dSp_dS[36];
        {
            real u00 = U[0], u01 = U[1], u02 = U[2], u10 = U[3], u11 = U[4], u12 = U[5];
            real r0[9] = {u00 * u00, u00 * u01, u00 * u02, u00 * u01, u01 * u01, u01 * u02, u00 * u02, u01 * u02, u02 * u02};
            real r1[9] = {u00 * u10, u00 * u11, u00 * u12, u01 * u10, u01 * u11, u01 * u12, u02 * u10, u02 * u11, u02 * u12};
            real r2[9] = {u00 * u10, u01 * u10, u02 * u10, u00 * u11, u01 * u11, u02 * u11, u00 * u12, u01 * u12, u02 * u12};
            real r3[9] = {u10 * u10, u10 * u11, u10 * u12, u10 * u11, u11 * u11, u11 * u12, u10 * u12, u11 * u12, u12 * u12};
            memcpy(dSp_dS, r0, sizeof r0); memcpy(dSp_dS + 9, r1, sizeof r1);
            memcpy(dSp_dS + 18, r2, sizeof r2); memcpy(dSp_dS + 27, r3, sizeof r3);
        }
        real dS_dM[81] = {
            2 * M[0], 2 * M[1], 2 * M[2], 0, 0, 0, 0, 0, 0,
            M[3], M[4], M[5], M[0], M[1], M[2], 0, 0, 0,
            M[6], M[7], M[8], 0, 0, 0, M[0], M[1], M[2],
            M[3], M[4], M[5], M[0], M[1], M[2], 0, 0, 0,
            0, 0, 0, 2 * M[3], 2 * M[4], 2 * M[5], 0, 0, 0,
            0, 0, 0, M[6], M[7], M[8], M[3], M[4], M[5],
            M[6], M[7], M[8], 0, 0, 0, M[0], M[1], M[2],
            0, 0, 0, M[6], M[7], M[8], M[3], M[4], M[5],
            0, 0, 0, 0, 0, 0, 2 * M[6], 2 * M[7], 2 * M[8]};
        real dSp_dM[36];
        matmul(dSp_dS, dS_dM, dSp_dM, 4, 9, 9);
        real dM_dS[27] = {R[0], 0, 0, 0, R[1], 0, 0, 0, R[2],
                          R[3], 0, 0, 0, R[4], 0, 0, 0, R[5],
                          R[6], 0, 0, 0, R[7], 0, 0, 0, R[8]};
        real tmp43[12], dSp_ds[12];
        matmul(dSp_dM, dM_dS, tmp43, 4, 9, 3);
        matmul(tmp43, S, dSp_ds, 4, 3, 3); /* d_S_d_s = diag(exp s) */
        real qx = f[0], qy = f[1], qz = f[2], qw = f[3], sx = es[0], sy = es[1], sz = es[2];
        real dM_dq[36] = {
            0, -4 * sx * qy, -4 * sx * qz, 0,
            2 * sy * qy, 2 * sy * qx, -2 * sy * qw, -2 * sy * qz,
            2 * sz * qz, 2 * sz * qw, 2 * sz * qx, 2 * sz * qy,
            2 * sx * qy, 2 * sx * qx, 2 * sx * qw, 2 * sx * qz,
            -4 * sy * qx, 0, -4 * sy * qz, 0,
            -2 * sz * qw, 2 * sz * qz, 2 * sz * qy, -2 * sz * qx,
            2 * sx * qz, -2 * sx * qw, 2 * sx * qx, -2 * sx * qy,
            2 * sy * qw, 2 * sy * qz, 2 * sy * qy, 2 * sy * qx,
            -4 * sz * qx, -4 * sz * qy, 0, 0};
        real dSp_dq[16];
        matmul(dSp_dM, dM_dq, dSp_dq, 4, 9, 4);
        real g4[4] = {A[2], A[3], A[3], A[4]}; /* RAS:716-721 */
        real *gf = grad_feat + (size_t)56 * id;
        for (int k = 0; k < 4; ++k)
            gf[k] = ((g4[0] * dSp_dq[k] + g4[1] * dSp_dq[4 + k]) + g4[2] * dSp_dq[8 + k]) + g4[3] * dSp_dq[12 + k];
        for (int k = 0; k < 3; ++k)
            gf[4 + k] = ((g4[0] * dSp_ds[k] + g4[1] * dSp_ds[3 + k]) + g4[2] * dSp_ds[6 + k]) + g4[3] * dSp_ds[9 + k];
        gf[7] = A[8];
        /* colour: RAS:749-756, ray origin = t_pointcloud_camera of the object (RAS:731) */
        const real *ro = t_pc + 3 * obj[id];
        real dir[3] = {p[0] - ro[0], p[1] - ro[1], p[2] - ro[2]}, Y[16];
        sh_basis(dir, Y);
        for (int ch = 0; ch < 3; ++ch) {
            const real *cf = f + 8 + 16 * ch;
            real s = cf[0] * Y[0];
            for (int k = 1; k < 16; ++k) s = s + cf[k] * Y[k];
            real sg = sigmoid(s);
            real jac = sg * (RC(1) - sg); /* UTL:356-359 */
            for (int k = 0; k < 16; ++k) gf[8 + 16 * ch + k] = A[5 + ch] * (jac * Y[k]);
        }
    }
}

/* ------------------------------------------------ adaptive-controller kernels (K9, K10) */
/* GP3:375-388 get_ellipsoid_foci_vector (called by ADC:10-25) */
void gs_oracle_ellipsoid_offsets(const real *feat, int n, real *out) {
    for (int i = 0; i < n; ++i) {
        const real *f = feat + (size_t)56 * i;
        real sx = f[4], sy = f[5], sz = f[6];
        int axis = 0;
        if (sx < sy && sy > sz) axis = 1;
        else if (sx < sz && sy < sz) axis = 2;
        real R[9];
        rotmat_from_q(f, R);
        real ex = R_EXP(sx), ey = R_EXP(sy), ez = R_EXP(sz);
        real rc = ex > ey ? (ex > ez ? ex : ez) : (ey > ez ? ey : ez);
        real ra = ex < ey ? (ex < ez ? ex : ez) : (ey < ez ? ey : ez);
        real len = R_SQRT(rc * rc - ra * ra);
        for (int k = 0; k < 3; ++k) out[3 * i + k] = len * R[3 * k + axis];
    }
}
/* GP3:390-406 sample (Box-Muller GP3:90-94) with the uniforms supplied by the caller (ADC:27-42) */
void gs_oracle_sample_from_points(const real *xyz, const real *feat, const real *u, int n, real *out) {
    const real two_pi = RC(2) * RC(3.141592653589);
    for (int i = 0; i < n; ++i) {
        const real *f = feat + (size_t)56 * i;
        real r1 = R_SQRT(RC(-2) * (real)log((double)u[4 * i])), r2 = R_SQRT(RC(-2) * (real)log((double)u[4 * i + 2]));
        real z1 = r1 * (real)cos((double)(two_pi * u[4 * i + 1])), z2 = r1 * (real)sin((double)(two_pi * u[4 * i + 1]));
        real z3 = r2 * (real)cos((double)(two_pi * u[4 * i + 3]));
        real R[9];
        rotmat_from_q(f, R);
        real b[3] = {R_EXP(f[4]) * z1, R_EXP(f[5]) * z2, R_EXP(f[6]) * z3};
        for (int k = 0; k < 3; ++k)
            out[3 * i + k] = xyz[3 * i + k] + ((R[3 * k] * b[0] + R[3 * k + 1] * b[1]) + R[3 * k + 2] * b[2]);
    }
}

/* ------------------------------------------------ trainer loss */
/* LossFunction.py:20-35 driven by GaussianPointTrainer.py:167-176: x = clamp(pred,0,1) (TRN:168),
 * L1 = mean|x-y| (LOS:31), SSIM = pytorch_msssim.ssim(x, y, data_range=1, size_average=True) (LOS:32-33; the
 * dependency is not vendored: requirements.txt:4, unpinned; algorithm restated from its published definition --
 * 11-tap Gaussian sigma 1.5, 'valid' separable filter per channel, K1=0.01, K2=0.03, mean over all outputs),
 * L = (1-lambda) L1 + lambda (1-SSIM) (LOS:34-35).  pred float[H][W][3] (hwc) or [3][H][W]; gt [3][H][W].
 * out[3] = {L, L1, 1-SSIM}; grad (same layout as pred, may be NULL) = g_total dL/dpred + g_l1 dL1/dpred +
 * g_dssim d(1-SSIM)/dpred, derived by hand (reverse sweep through the two separable filters); the clamp passes
 * gradients on the closed interval like torch.clamp.  Always evaluated in double. */
void gs_oracle_l1_ssim(const real *pred, int hwc, int clamp, const real *gt, int H, int W, double lambda,
                       double g_total, double g_l1, double g_dssim, double *out, real *grad) {
    const int Ho = H - 10, Wo = W - 10;
    double win[11], wsum = 0;
    for (int k = 0; k < 11; ++k) { win[k] = exp(-((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); wsum += win[k]; }
    for (int k = 0; k < 11; ++k) win[k] /= wsum;
    const double c1 = 0.01 * 0.01, c2 = 0.03 * 0.03;
    const size_t P = (size_t)H * W;
    double *x = malloc(P * sizeof(double)), *y = malloc(P * sizeof(double));
    double *h = malloc(5 * (size_t)H * Wo * sizeof(double));     /* after the horizontal pass */
    double *dm = calloc(3 * P, sizeof(double));                   /* A, B, C on the output grid, zero elsewhere */
    double *hb = malloc(3 * (size_t)H * Wo * sizeof(double));
    double l1_sum = 0, ssim_sum = 0;
    const double w_l1 = (g_total * (1.0 - lambda) + g_l1) / (3.0 * H * W);
    const double w_ss = -(g_total * lambda + g_dssim) / (3.0 * Ho * (double)Wo);
    for (int c = 0; c < 3; ++c) {
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                double v = (double)(hwc ? pred[((size_t)i * W + j) * 3 + c] : pred[((size_t)c * H + i) * W + j]);
                if (clamp) v = v < 0 ? 0 : (v > 1 ? 1 : v);
                x[(size_t)i * W + j] = v;
                y[(size_t)i * W + j] = (double)gt[((size_t)c * H + i) * W + j];
                l1_sum += fabs(v - y[(size_t)i * W + j]);
            }
#pragma omp parallel for schedule(static)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < Wo; ++j) {
                double s[5] = {0, 0, 0, 0, 0};
                for (int k = 0; k < 11; ++k) {
                    double a = x[(size_t)i * W + j + k], b = y[(size_t)i * W + j + k];
                    s[0] += win[k] * a; s[1] += win[k] * b; s[2] += win[k] * a * a; s[3] += win[k] * b * b;
                    s[4] += win[k] * a * b;
                }
                for (int m = 0; m < 5; ++m) h[((size_t)m * H + i) * Wo + j] = s[m];
            }
        memset(dm, 0, 3 * P * sizeof(double));
        for (int i = 0; i < Ho; ++i)
            for (int j = 0; j < Wo; ++j) {
                double s[5] = {0, 0, 0, 0, 0};
                for (int k = 0; k < 11; ++k)
                    for (int m = 0; m < 5; ++m) s[m] += win[k] * h[((size_t)m * H + i + k) * Wo + j];
                double mx = s[0], my = s[1];
                double d1 = mx * mx + my * my + c1, d2 = (s[2] - mx * mx) + (s[3] - my * my) + c2;
                double l = (2 * mx * my + c1) / d1, cs = (2 * (s[4] - mx * my) + c2) / d2;
                ssim_sum += l * cs;
                dm[0 * P + (size_t)i * W + j] = cs * (2 * my - 2 * mx * l) / d1 + l * (2 * mx * cs - 2 * my) / d2;
                dm[1 * P + (size_t)i * W + j] = -l * cs / d2;
                dm[2 * P + (size_t)i * W + j] = 2 * l / d2;
            }
        if (grad) {
            /* transpose of the 'valid' filter: full correlation of the output-grid maps with the window */
#pragma omp parallel for schedule(static)
            for (int i = 0; i < Ho; ++i)
                for (int j = 0; j < W; ++j)
                    for (int m = 0; m < 3; ++m) {
                        double s = 0;
                        for (int k = 0; k < 11; ++k) {
                            int jj = j - k;
                            if (jj >= 0 && jj < Wo) s += win[k] * dm[m * P + (size_t)i * W + jj];
                        }
                        hb[((size_t)m * Ho + i) * W + j] = s;
                    }
#pragma omp parallel for schedule(static)
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) {
                    double t[3] = {0, 0, 0};
                    for (int k = 0; k < 11; ++k) {
                        int ii = i - k;
                        if (ii >= 0 && ii < Ho)
                            for (int m = 0; m < 3; ++m) t[m] += win[k] * hb[((size_t)m * Ho + ii) * W + j];
                    }
                    double xv = x[(size_t)i * W + j], yv = y[(size_t)i * W + j], d = xv - yv;
                    double g = w_l1 * (d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0)) + w_ss * (t[0] + 2 * xv * t[1] + yv * t[2]);
                    size_t o = hwc ? ((size_t)i * W + j) * 3 + c : ((size_t)c * H + i) * W + j;
                    if (clamp && !((double)pred[o] >= 0 && (double)pred[o] <= 1)) g = 0;
                    grad[o] = (real)g;
                }
        }
    }
    double l1 = l1_sum / (3.0 * H * W), dssim = 1.0 - ssim_sum / (3.0 * Ho * (double)Wo);
    out[0] = (1.0 - lambda) * l1 + lambda * dssim; out[1] = l1; out[2] = dssim;
    free(x); free(y); free(h); free(dm); free(hb);
}
