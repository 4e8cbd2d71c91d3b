// gs_blend.hip -- per-tile alpha blending, forward and backward, for gfx950 (wave64).
//
// Workgroup = one 16x16 tile = 4 wavefronts; lane = pixel (a wave covers 4 rows x 16 columns).
// The tile's depth-sorted Gaussian list is staged through LDS in batches of 256 packed 48-B
// records (3 x float4 per Gaussian, gathered with 16-B loads).
//  forward : front-to-back blend (RAS:318-485, weight UTL:275-284); whole-tile early exit with a
//            workgroup vote (the reference could not express it, RAS:387-394).
//  backward: back-to-front traversal (RAS:531-705, gradients UTL:331-348) starting at the
//            tile's last effective entry; the 10 per-Gaussian partial sums are reduced across
//            the 64 lanes with DPP row shifts/broadcasts and ONE lane issues the hardware fp32
//            atomics -- one atomic set per (wave, Gaussian) instead of one per (pixel, Gaussian).
#include "gs_common.h"

// Automatic FMA contraction is off in this file and every fused multiply-add is written explicitly:
// the unrolled copies of the inner loops then execute the same instruction sequence for a Gaussian
// whatever its position in the tile list, so results do not depend on list positions (e.g. the exact
// tile cull, which only removes non-contributing entries, leaves every output bit-identical).
#pragma clang fp contract(off)

namespace {

constexpr float EPS_ALPHA = (float)(1.0 / 255.0);  // RAS:451
constexpr float CLAMP_ALPHA = 0.99f;               // RAS:453
constexpr float STOP_T = 0.0001f;                  // RAS:458
constexpr int GROUP = 4;  // list entries evaluated together in the blend loops (GS_BLOCK % GROUP == 0)

struct TileCoord { int tile_u, tile_v, tile_id; };

// blockIdx -> owned tile.  Consecutive workgroups are dispatched round-robin over the 8 XCDs
// (block b -> XCD b%8); the remap gives every XCD a contiguous run of tiles so that neighbouring
// tiles (which share Gaussians) hit the same 4-MiB L2.  Speed only, never correctness.
__device__ __forceinline__ TileCoord owned_tile(int tw, int row_begin, int row_step) {
    const int nb = gridDim.x;
    int b = blockIdx.x;
    const int chunk = nb / 8;
    if (b < chunk * 8) b = (b % 8) * chunk + b / 8;
    TileCoord t;
    t.tile_u = b % tw;
    t.tile_v = row_begin + (b / tw) * row_step;
    t.tile_id = t.tile_u + t.tile_v * tw;
    return t;
}

// ------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(GS_BLOCK) void blend_forward_kernel(
    const int32_t *__restrict__ tile_start, const int32_t *__restrict__ tile_end,
    const int32_t *__restrict__ payload, const float4 *__restrict__ attrs, int width, int height, int row_begin,
    int row_step, float *__restrict__ image, float *__restrict__ depth, float *__restrict__ acc_alpha,
    int32_t *__restrict__ last_effective, int32_t *__restrict__ valid_count) {
    __shared__ float4 s_a[GS_BLOCK], s_b[GS_BLOCK], s_c[GS_BLOCK];
    const int tw = width / GS_TILE_WIDTH;
    const TileCoord tc = owned_tile(tw, row_begin, row_step);
    const int tid = threadIdx.x;
    const int pu = tc.tile_u * GS_TILE_WIDTH + (tid & 15);
    const int pv = tc.tile_v * GS_TILE_HEIGHT + (tid >> 4);
    const int start = tile_start[tc.tile_id], end = tile_end[tc.tile_id];
    const float px = (float)pu + 0.5f, py = (float)pv + 0.5f;

    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, D = 0.f, Wd = 0.f;
    int last = start, cnt = 0;
    bool done = false;

    for (int base = start; base < end; base += GS_BLOCK) {
        // barrier (protects the LDS batch) + whole-tile early exit vote
        if (__syncthreads_and(done ? 1 : 0)) break;
        const int j = base + tid;
        if (j < end) {
            const float4 *g = attrs + 3 * (size_t)payload[j];
            s_a[tid] = g[0];
            s_b[tid] = g[1];
            s_c[tid] = g[2];
        } else {  // padding record: opacity 0 -> alpha 0, never blended
            s_a[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            s_b[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        const int n = min(GS_BLOCK, end - base);
        // Entries are evaluated in groups of GROUP: the LDS reads and the exp of a group are independent
        // and overlap (the per-pixel blend recurrence is the only serial part), which hides their latency.
        for (int k = 0; k < n; k += GROUP) {
            if (__ballot(!done) == 0ull) break;  // every pixel of this wave is saturated
            float alpha[GROUP], z[GROUP];
#pragma unroll
            for (int i = 0; i < GROUP; ++i) {
                const float4 a = s_a[k + i], b = s_b[k + i];
                const float dx = px - a.x, dy = py - a.y;
                // UTL:275-284
                const float e = fmaf(-0.5f, fmaf(dy * dy, b.z, dx * dx * b.x), -(dx * dy) * b.y);
                alpha[i] = __expf(e) * b.w * a.w;
                z[i] = a.z;
            }
#pragma unroll
            for (int i = 0; i < GROUP; ++i) {
                const bool ok = !done && alpha[i] >= EPS_ALPHA;  // RAS:451
                if (__ballot(ok) == 0ull) continue;              // wave-uniform skip
                if (ok) {
                    const float al = fminf(alpha[i], CLAMP_ALPHA);
                    const float Tn = T * (1.f - al);
                    if (Tn < STOP_T) {  // RAS:458-460: saturated, this Gaussian is not blended
                        done = true;
                    } else {
                        const float4 c = s_c[k + i];
                        const float wgt = al * T;
                        last = base + k + i + 1;
                        Cr = fmaf(c.x, wgt, Cr); Cg = fmaf(c.y, wgt, Cg); Cb = fmaf(c.z, wgt, Cb);
                        D = fmaf(z[i], wgt, D);
                        Wd += wgt;
                        cnt += 1;
                        T = Tn;
                    }
                }
            }
        }
    }
    const size_t p = (size_t)pv * width + pu;
    image[3 * p] = Cr; image[3 * p + 1] = Cg; image[3 * p + 2] = Cb;
    depth[p] = D / fmaxf(Wd, 1e-6f);  // RAS:479-480
    acc_alpha[p] = 1.f - T;
    last_effective[p] = last;
    valid_count[p] = cnt;
}

// ------------------------------------------------------------------------------- backward
// Per batch of 256 list entries the 4 waves of the tile combine their partial sums in LDS
// (ds_add_f32), then thread k flushes entry k with ONE set of hardware atomics per (tile, Gaussian).
__global__ __launch_bounds__(GS_BLOCK) void blend_backward_kernel(
    const int32_t *__restrict__ tile_start, const int32_t *__restrict__ tile_end,
    const int32_t *__restrict__ payload, const float4 *__restrict__ attrs, const float *__restrict__ grad_image,
    const float *__restrict__ acc_alpha, const int32_t *__restrict__ last_effective, int width, int height,
    int row_begin, int row_step, float *__restrict__ acc, float *__restrict__ magnitude_image) {
    __shared__ float4 s_a[GS_BLOCK], s_b[GS_BLOCK], s_c[GS_BLOCK];
    __shared__ int s_o[GS_BLOCK];
    __shared__ float s_acc[GS_BLOCK][GS_ACC_STRIDE];  // [entry][value]; slot 10 = pixel count (int bits)
    __shared__ int s_max[GS_BLOCK / GS_WAVE];
    const int tw = width / GS_TILE_WIDTH;
    const TileCoord tc = owned_tile(tw, row_begin, row_step);
    const int tid = threadIdx.x, lane = tid & 63;
    const int pu = tc.tile_u * GS_TILE_WIDTH + (tid & 15);
    const int pv = tc.tile_v * GS_TILE_HEIGHT + (tid >> 4);
    const size_t p = (size_t)pv * width + pu;
    const int start = tile_start[tc.tile_id];
    const float px = (float)pu + 0.5f, py = (float)pv + 0.5f;

    const int last = last_effective[p];
    float T = 1.0f - acc_alpha[p];
    float wr = 0.f, wg = 0.f, wb = 0.f;
    const float Gr = grad_image[3 * p], Gg = grad_image[3 * p + 1], Gb = grad_image[3 * p + 2];
    float mag_u = 0.f, mag_v = 0.f;

    // no pixel of the tile touches an entry at or beyond the tile-wide max of `last`
    int mx = last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, GS_WAVE));
    if (lane == 0) s_max[tid >> 6] = mx;
    __syncthreads();
    const int end = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));

    for (int top = end; top > start; top -= GS_BLOCK) {
        __syncthreads();  // previous batch fully flushed before its LDS is reused
        const int j = top - 1 - tid;
        if (j >= start) {
            const int o = payload[j];
            const float4 *g = attrs + 3 * (size_t)o;
            s_a[tid] = g[0];
            s_b[tid] = g[1];
            s_c[tid] = g[2];
            s_o[tid] = o;
        } else {  // padding record: opacity 0 -> never a hit
            s_a[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            s_b[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        {
            float4 *z = reinterpret_cast<float4 *>(&s_acc[tid][0]);
            z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        const int n = min(GS_BLOCK, top - start);
        for (int k = 0; k < n; k += GROUP) {
            // group evaluation: LDS reads + exp of GROUP entries are independent and overlap
            float g[GROUP], m0[GROUP], m1[GROUP], op[GROUP];
            bool hit[GROUP];
#pragma unroll
            for (int i = 0; i < GROUP; ++i) {
                const float4 a = s_a[k + i], b = s_b[k + i];
                const float dx = px - a.x, dy = py - a.y;
                // UTL:331-348: m = conic @ d, exponent = -0.5 d.m
                m0[i] = fmaf(b.y, dy, b.x * dx);
                m1[i] = fmaf(b.z, dy, b.y * dx);
                g[i] = __expf(-0.5f * fmaf(dy, m1[i], dx * m0[i])) * b.w;
                op[i] = a.w;
                hit[i] = (top - 1 - (k + i) < last) && (g[i] * a.w >= EPS_ALPHA);
            }
#pragma unroll
            for (int i = 0; i < GROUP; ++i) {
                const unsigned long long hits = __ballot(hit[i]);
                if (hits == 0ull) continue;  // wave-uniform skip: nobody in this wave touches the Gaussian
                float v0 = 0.f, v1 = 0.f, c00 = 0.f, c01 = 0.f, c11 = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gl = 0.f,
                      nv = 0.f;
                if (hit[i]) {
                    const float4 c = s_c[k + i];
                    const float alpha = fminf(g[i] * op[i], CLAMP_ALPHA);
                    const float inv1m = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * inv1m;  // RAS:643, T_i = T_{i+1} / (1 - alpha_i)
                    const float aT = alpha * T;
                    gr = aT * Gr; gg = aT * Gg; gb = aT * Gb;
                    const float dLda = fmaf(fmaf(c.z, T, -(wb * inv1m)), Gb,
                                            fmaf(fmaf(c.y, T, -(wg * inv1m)), Gg, fmaf(c.x, T, -(wr * inv1m)) * Gr));
                    wr = fmaf(c.x, aT, wr); wg = fmaf(c.y, aT, wg); wb = fmaf(c.z, aT, wb);
                    gl = dLda * g[i] * (1.f - op[i]) * op[i];
                    const float dLdg = dLda * op[i];
                    const float gm0 = g[i] * m0[i], gm1 = g[i] * m1[i];
                    v0 = dLdg * gm0; v1 = dLdg * gm1;
                    mag_u += fabsf(v0); mag_v += fabsf(v1);
                    const float h = 0.5f * dLdg;
                    c00 = h * gm0 * m0[i]; c01 = h * gm0 * m1[i]; c11 = h * gm1 * m1[i];
                    nv = __builtin_amdgcn_sqrtf(fmaf(v1, v1, v0 * v0));  // v_sqrt_f32, 1 ulp
                }
                // reduce-scatter of the 10 partial sums over the 64 lanes (gs_common.h); row totals land in
                // lane 15 of each row:  t0: rows = (v0, c00, v1, c01)  t1: (c11, gg, gr, gb)  t2: (gl, gl, nv, nv)
                float t0, t1, t2;
                gs_wave_reduce10(v0, v1, c00, c01, c11, gr, gg, gb, gl, nv, t0, t1, t2);
                if ((lane & 15) == 15) {
                    const int row = lane >> 4;
                    const int slot = ((row & 1) << 1) | (row >> 1);  // rows (0,1,2,3) -> slots (0,2,1,3)
                    float *A = &s_acc[k + i][0];
                    atomicAdd(A + slot, t0);
                    atomicAdd(A + 4 + slot, t1);
                    if ((row & 1) == 0) atomicAdd(A + 8 + (row >> 1), t2);
                    if (row == 3) atomicAdd(reinterpret_cast<int *>(A + 10), (int)__popcll(hits));
                }
            }
        }
        __syncthreads();
        // flush: thread k owns entry k of the batch -> one global atomic set per (tile, Gaussian)
        if (tid < n) {
            const int npix = __builtin_bit_cast(int, s_acc[tid][10]);
            if (npix > 0) {
                float *A = acc + (size_t)GS_ACC_STRIDE * s_o[tid];
#pragma unroll
                for (int v = 0; v < 10; ++v) atomicAdd(A + v, s_acc[tid][v]);
                atomicAdd(reinterpret_cast<int *>(A + 10), npix);
            }
        }
    }
    magnitude_image[2 * p] = mag_u;
    magnitude_image[2 * p + 1] = mag_v;
}

}  // namespace

extern "C" {

static int owned_row_count(int th, int begin, int step) { return begin < th ? (th - 1 - begin) / step + 1 : 0; }

int gs_blend_forward(const int32_t *tile_start, const int32_t *tile_end, const int32_t *payload, const float *attrs,
                     int width, int height, int tile_row_begin, int tile_row_step, float *image, float *depth,
                     float *acc_alpha, int32_t *last_effective, int32_t *valid_count, void *stream) {
    GS_REQUIRE(width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0, "image size must be a multiple of 16");
    GS_REQUIRE(tile_row_step >= 1 && tile_row_begin >= 0, "tile row ownership");
    const int tw = width / GS_TILE_WIDTH, rows = owned_row_count(height / GS_TILE_HEIGHT, tile_row_begin, tile_row_step);
    if (rows == 0 || tw == 0) return 0;
    hipLaunchKernelGGL(blend_forward_kernel, dim3(tw * rows), dim3(GS_BLOCK), 0, (hipStream_t)stream, tile_start,
                       tile_end, payload, reinterpret_cast<const float4 *>(attrs), width, height, tile_row_begin,
                       tile_row_step, image, depth, acc_alpha, last_effective, valid_count);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_blend_backward(const int32_t *tile_start, const int32_t *tile_end, const int32_t *payload, const float *attrs,
                      const float *grad_image, const float *acc_alpha, const int32_t *last_effective, int n_visible,
                      int width, int height, int tile_row_begin, int tile_row_step, float *acc, float *magnitude_image,
                      void *stream) {
    GS_REQUIRE(width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0, "image size must be a multiple of 16");
    GS_REQUIRE(tile_row_step >= 1 && tile_row_begin >= 0, "tile row ownership");
    GS_REQUIRE(n_visible >= 0, "n_visible");
    hipStream_t s = (hipStream_t)stream;
    if (n_visible > 0)
        GS_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(float) * GS_ACC_STRIDE * (size_t)n_visible, s));
    const int tw = width / GS_TILE_WIDTH, rows = owned_row_count(height / GS_TILE_HEIGHT, tile_row_begin, tile_row_step);
    if (rows == 0 || tw == 0) return 0;
    hipLaunchKernelGGL(blend_backward_kernel, dim3(tw * rows), dim3(GS_BLOCK), 0, s, tile_start, tile_end, payload,
                       reinterpret_cast<const float4 *>(attrs), grad_image, acc_alpha, last_effective, width, height,
                       tile_row_begin, tile_row_step, acc, magnitude_image);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
