// gs_common.h -- shared host/device helpers of the gfx950 rasteriser library.
// Wave size is 64 (CDNA4); every wave-level idiom below is written for 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gsplat_hip.h"

#define GS_WAVE 64
#define GS_BLOCK 256

// ------------------------------------------------------------------ error handling (host)
void gs_set_error(const char *fmt, ...);

#define GS_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            gs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return -2;                                                                  \
        }                                                                               \
    } while (0)

#define GS_CHECK_LAUNCH()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            gs_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return -3;                                                                  \
        }                                                                               \
    } while (0)

#define GS_REQUIRE(cond, msg)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            gs_set_error("invalid argument: %s (%s)", msg, #cond);                      \
            return -1;                                                                  \
        }                                                                               \
    } while (0)

static inline int gs_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ wave / block primitives
#ifdef __HIPCC__

// exp(x) correctly rounded to fp32: evaluated in double and rounded once (the oracle and the emulated reference run do the
// same with their libm's double exp; two double evaluations that are each within an ulp of the true value round to the
// same float unless it lies within ~2e-16 (relative) of a rounding boundary: once in ~1e8 calls).  See the scale activation
// in gs_frontend.hip and the exact decisions of the blend kernels.  Written out (k = rint(x / ln 2), r = x - k ln 2 in two
// pieces, Taylor polynomial of degree 13 on |r| <= 0.347: truncation 4e-18, then one ldexp) rather than calling the device
// library's exp: a dozen live registers instead of forty, which matters where it is inlined next to a hot loop.
// (A double literal the optimiser cannot hoist: left to itself it materialises the polynomial's coefficients at the top of
// the kernel -- half of them in VGPR pairs, the scalar file being full -- where they stay live through the hot loops of the
// blend kernels: +27 registers, one wave per SIMD less.  Two s_mov_b32 at the point of use instead.)
template <unsigned long long BITS>
__device__ __forceinline__ double gs_lit_f64() {
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "i"((unsigned)(BITS & 0xffffffffull)), "i"((unsigned)(BITS >> 32)));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
#define GS_LIT_F64(x) gs_lit_f64<__builtin_bit_cast(unsigned long long, (double)(x))>()
__device__ __forceinline__ float gs_exp_cr(float xf) {
    const double x = (double)fminf(fmaxf(xf, -800.0f), 100.0f);   // (NaN passes through fminf / fmaxf as the other operand: -800 -> 0)
    const double kd = __builtin_rint(x * GS_LIT_F64(1.44269504088896340736));
    double r = __builtin_fma(kd, GS_LIT_F64(-6.93147180369123816490e-01), x);
    r = __builtin_fma(kd, GS_LIT_F64(-1.90821492927058770002e-10), r);
    double p = GS_LIT_F64(1.0 / 6227020800.0);
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 479001600.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 39916800.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 3628800.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 362880.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 40320.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 5040.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 720.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 120.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 24.0));
    p = __builtin_fma(p, r, GS_LIT_F64(1.0 / 6.0));
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const float y = (float)__builtin_ldexp(p, (int)kd);
    return xf != xf ? xf : y;
}

// ------------------------------------------------------------------ threshold decisions taken as the reference takes them
// The reference skips a (pixel, Gaussian) pair whose alpha is below 1/255 (RAS:451 in the forward pass, RAS:631 in the
// backward pass) and stops a pixel at the first Gaussian that would take its transmittance below 1e-4 (RAS:458).  Both are
// DISCRETE decisions (a whole Gaussian blended or not): to produce the reference's results they must be taken exactly as
// the reference takes them, on every pixel.  How the blend kernels (gs_blend.hip) get there:
//  1. THE EXPONENT IS THE REFERENCE'S, TO THE LAST BIT.  Each pass evaluates the quadratic form in the operation order of
//     its counterpart -- UTL:281-283 in the forward pass, UTL:336-339 (m = conic @ d first) in the backward pass -- from the
//     same fp32 (dx, dy, A, B, C), contraction off.  (Rounds 1-4 evaluated a pre-scaled conic in the log2 domain, three
//     instructions shorter per pixel pair: its rounding differs from the reference's by 9 u M, M = |A dx^2|/2 + |C dy^2|/2 +
//     |B dx dy| -- the CANCELLATION between the three terms, up to thousands of times the exponent itself for a thin diagonal
//     Gaussian -- which put 1e-4 between the two images on needle scenes and made every bracket below proportional to the
//     conic's condition; the reference-order form leaves nothing that depends on the conic.)
//  2. What remains between the two alphas (u = 2^-24, relative):
//       kernels  : e * log2(e) -- constant and product rounded -> 2 u |e| on alpha;  v_exp_f32 1 ulp (taken as 4 u);
//                  amp = fl(opacity * rescale) u;  exp * amp u                                  -> 2 u |e| + 6 u
//       reference: exp correctly rounded u;  * rescale u (UTL:284);  * opacity u (RAS:447)       -> 3 u
//     |ln alpha_kernel - ln alpha_reference| <= u (2 |e| + 9), |e| = ln(amp / alpha) <= ln(1 / alpha) (amp <= 1), and a third on
//     top (GS_BAND_SAFETY).  At the 1/255 threshold |e| < 5.6 for every positive-definite conic; the band is cut for
//     |e| <= 20 (a conic that rounding has left indefinite can reach the threshold from a positive exponent):
//     GS_ALPHA_BAND = 4/3 * 49 u = 3.9e-6.
//  3. A comparison whose operand lies inside the bracket [threshold (1 - band), threshold / (1 - band)) is settled by the
//     reference's own expression: exp(e) correctly rounded (gs_exp_cr: the definition the oracle and the committed
//     reference-run vectors use), times rescale, times opacity -- wave-uniform rare paths, kept OUT of the hot loops.
//  4. The stop test compares T' = T (1 - a) with 1e-4, T a product of (1 - a_i): the two sides' factors differ (relatively) by
//     delta_i a_i / (1 - a_i) + 4 u (two roundings per side once the inputs differ), delta_i = u (2 ln(1 / a_i) + 9):
//       a ln(1/a) / (1 - a) <= min(1, 6 a) on [1/255, 0.99] (and for a clamped factor)      -> 2 u * 6 a_i
//       a / (1 - a) <= a H_k,  H_k = 1 / (1 - min(amp_k, 0.99)) -- per GAUSSIAN: alpha <= amp    -> 9 u * H_k a_i
//     accumulated PER PIXEL with one packed FMA per hit-path execution: thr += W_k a_i, W_k = gs_stop_weight(amp_k) (stored
//     in the record), thr = upper edge of the pixel's bracket around 1e-4; the rounding term (4 u per blended Gaussian) is
//     added per batch from the list positions walked (gates the rare path) and replaced by the pixel's own count inside it.
//     thr = STOP (1 + 1.1 beta) >= STOP e^beta holds for beta <= 0.19; here beta <= 4/3 u (12 + 9 * 101) sum(a_i) < 1.2e-3
//     (sum a_i <= ln(1 / 1e-6)).  A T' inside [STOP - (thr - STOP) / 1.1, thr) is settled by replaying the pixel's history in
//     the reference's arithmetic (gs_reference_stops_at, gs_blend.hip).
#define GS_U24 5.9604644775390625e-8f
#define GS_BAND_SAFETY (4.0f / 3.0f)
#define GS_ALPHA_BAND (GS_BAND_SAFETY * GS_U24 * (2.0f * 20.0f + 9.0f))
// thr itself is an fp32 number near 1e-4 (ulp 2^-37 = 7.3e-12): every `thr += W_k a` rounds by up to half an ulp -- an
// increment below that is lost altogether -- i.e. by 0.56 u in the units of the rounding term (1.1 STOP u = 6.6e-12), so the
// term charged per blended Gaussian is 5 u, not the 4 u of the transmittance product alone; the per-batch addition and the
// subtraction inside gs_stop_bracket round once each: GS_STOP_THR_SLACK (more than two half-ulps) per batch, never taken
// back.
#define GS_STOP_ROUNDING_BAND (5.0f * GS_U24)   /* per blended Gaussian: 4 u (two roundings per side) + 1 u (thr's own) */
#define GS_STOP_THR_SLACK 8.0e-12f              /* per batch */
__device__ __forceinline__ float gs_stop_weight(float amp, float stop_t) {
    const float H = 1.0f / (1.0f - fminf(amp, 0.99f)) * 1.01f;
    return stop_t * 1.1f * GS_BAND_SAFETY * GS_U24 * (12.0f + 9.0f * H);
}
// THE 1/255 DECISION IN THE EXPONENT'S DOMAIN (round 6).  The reference's alpha is a non-decreasing function of its exponent e
// -- exp correctly rounded, then two roundings of products with positive constants -- so "alpha >= 1/255" is "e >= e*" for
// one float e* per Gaussian, and e* lies within 3 u of L = ln(fl(1/255) / (rescale opacity)) (three relative roundings of at most
// u each, additive in the log domain).  gs_preprocess leaves e_lo = L~ - h in the record (float 14), L~ = -logf(255 amp):
//   |L~ - L| <= 3 u (amp = fl(opacity rescale), the product with 255, fl(1/255) against 1/255) + 2 ulp(L~) (logf: 1 ulp claimed,
//   two charged) = u (3 + 4 |L~|);  with the reference's own 3 u:  h = 4/3 u (6 + 4 |L~|)   (GS_BAND_SAFETY on top, as everywhere).
// e < e_lo => the reference skips the pair, whatever the blend kernels' own alpha would say: they compare the exponent (which
// is the reference's to the last bit) BEFORE evaluating any exponential, and pay for v_exp_f32 only where a pixel may be hit.
// A pixel with e >= e_lo is a hit for certain once the kernels' alpha reaches EPS_HI (the bracket of 2. above); in between
// (e within ~1e-5 of the threshold: a few hundred pixels per full-size frame) the reference's expression decides (3. above).
// amp = 0 or NaN gives NaN: no exponent compares >= NaN, and the reference's alpha (0, or NaN) is never >= 1/255 either.
__device__ __forceinline__ float gs_hit_exponent_lo(float amp) {
#pragma clang fp contract(off)
    const float L = -logf(255.0f * amp);
    return L - GS_BAND_SAFETY * GS_U24 * (6.0f + 4.0f * fabsf(L));
}
// exp(e) * rescale * opacity as the reference rounds it (UTL:284 then RAS:447 / RAS:627), e = the reference's exponent
__device__ __forceinline__ float gs_alpha_reference(float e, float rescale, float opacity) {
#pragma clang fp contract(off)
    return gs_exp_cr(e) * rescale * opacity;
}

// RAS:81-103 get_bounding_box_by_point_and_radii (shared by the front end and the backward flush, which
// must agree bit-for-bit on the box: it defines the slot of a (Gaussian, tile) pair, RAS:163-166)
__device__ __forceinline__ void gs_tile_box(float u, float v, float r, int tw, int th, int &t0u, int &t1u,
                                            int &t0v, int &t1v) {
    r = fmaxf(r, 1.0f);
    float min_u = fmaxf(0.0f, u - r), max_u = u + r;
    float min_v = fmaxf(0.0f, v - r), max_v = v + r;
    t0u = min((int)floorf(min_u / (float)GS_TILE_WIDTH), tw);
    t1u = min(max((int)floorf(max_u / (float)GS_TILE_WIDTH) + 1, t0u + 1), tw);
    t0v = min((int)floorf(min_v / (float)GS_TILE_HEIGHT), th);
    t1v = min(max((int)floorf(max_v / (float)GS_TILE_HEIGHT) + 1, t0v + 1), th);
}


// ------------------------------------------------------------------ exact contribution test (output-identical culling)
// A (pixel rectangle, Gaussian) pair whose alpha stays below the 1/255 skip threshold (RAS:451) on EVERY pixel centre of
// the rectangle never changes any pixel state, so dropping it from a list is output-identical.  alpha = amp * exp(-q/2)
// with q the conic quadratic form (UTL:275-284); q is minimised over the convex hull of the rectangle's pixel centres
// (a superset of the pixels), which lies on the edge(s) of the rectangle facing the centre.  qmax = 2 ln(255 amp) +
// margin; the margin (1e-2 in q, i.e. 0.5 % in alpha) dwarfs every fp32 disagreement between this test and the blend
// kernels (and the error of the approximate reciprocals below: < 1e-5 in q).  The same function serves the front end
// (per 64 x 64-pixel bin: which bins get a sort key) and the blend kernels (per 16 x 16 tile: which entries of the bin's
// list are staged for this tile); the two decisions need not agree -- each one only ever removes pairs that contribute
// nothing.
__device__ __forceinline__ float gs_cull_qmax(float amp) { return 2.0f * logf(255.0f * amp) + 1e-2f; }

// The tile box a bin walk has to look at: the reference's box (a square around the 3-sigma circle, RAS:81-103)
// intersected with the axis-aligned bounding box of the level set q <= qmax -- half-widths sqrt(qmax C / det) and
// sqrt(qmax A / det), taken 0.01 % and one pixel wider than computed, so that no tile with a contributing pixel is lost.
// A needle-shaped Gaussian's square is mostly empty (the test below rejects those bins one by one: 10,000 screen-long
// needles cost +0.7 ms per frame in the two walks); its bounding box is not, unless the needle is diagonal.  qmax = +inf
// (cull off) or a degenerate conic leave the box as it is; qmax < 0 (never visible) empties it.
__device__ __forceinline__ void gs_cull_box(float u, float v, float A, float B, float C, float qmax, int &t0u, int &t1u,
                                            int &t0v, int &t1v) {
#pragma clang fp contract(off)
    const float det = A * C - B * B;
    if (!(qmax < 1e30f) || !(det > 0.f)) return;
    if (qmax < 0.f) { t1u = t0u; t1v = t0v; return; }
    const float s = qmax / det;
    const float wx = sqrtf(s * C) * 1.0001f + 1.0f, wy = sqrtf(s * A) * 1.0001f + 1.0f;
    // tile column t holds pixel centres 16 t + 0.5 .. 16 t + 15.5: it matters only if that span meets [u - wx, u + wx]
    const float lim = 1.0e6f;
    const int c0u = (int)ceilf(fminf(fmaxf((u - wx - ((float)GS_TILE_WIDTH - 0.5f)) / (float)GS_TILE_WIDTH, -lim), lim));
    const int c1u = (int)floorf(fminf(fmaxf((u + wx - 0.5f) / (float)GS_TILE_WIDTH, -lim), lim)) + 1;
    const int c0v = (int)ceilf(fminf(fmaxf((v - wy - ((float)GS_TILE_HEIGHT - 0.5f)) / (float)GS_TILE_HEIGHT, -lim), lim));
    const int c1v = (int)floorf(fminf(fmaxf((v + wy - 0.5f) / (float)GS_TILE_HEIGHT, -lim), lim)) + 1;
    t0u = max(t0u, c0u); t1u = max(min(t1u, c1u), t0u);
    t0v = max(t0v, c0v); t1v = max(min(t1v, c1v), t0v);
}

// Rows of tile column `cu` that can hold a pixel with q <= qmax: the level set cut by the strip of the column's pixel
// centres (one pixel wider on both sides) is convex, so its y-extent is an interval -- the upper boundary
// y(dx) = (-B dx + sqrt(C qmax - det dx^2)) / C is concave and peaks at dx = -B sqrt(qmax / (A det)), the lower one is
// its mirror image.  Conservative by construction (wider strip, interval taken 0.01 % + one pixel wider, negative
// discriminants clamped); callers intersect with the box rows.  Requires det > 0, 0 <= qmax < inf (as gs_cull_box).
__device__ __forceinline__ void gs_cull_rows_in_column(float u, float v, float A, float B, float C, float qmax, int cu,
                                                       int &r0, int &r1) {
#pragma clang fp contract(off)
    const float det = A * C - B * B;
    const float a = ((float)(cu * GS_TILE_WIDTH) + 0.5f - 1.0f) - u, b = ((float)(cu * GS_TILE_WIDTH + GS_TILE_WIDTH) - 0.5f + 1.0f) - u;
    const float peak = B * sqrtf(qmax / (A * det));            // |dx| of the two extreme points
    const float dxu = fminf(fmaxf(-peak, a), b), dxl = fminf(fmaxf(peak, a), b);
    const float inv_c = 1.0f / C;
    const float up = (-B * dxu + sqrtf(fmaxf(C * qmax - det * dxu * dxu, 0.f))) * inv_c;
    const float lo = (-B * dxl - sqrtf(fmaxf(C * qmax - det * dxl * dxl, 0.f))) * inv_c;
    const float slack = 1.0f + 1.0e-4f * (fabsf(up) + fabsf(lo));
    const float lim = 1.0e6f;
    r0 = (int)ceilf(fminf(fmaxf((v + lo - slack - ((float)GS_TILE_HEIGHT - 0.5f)) / (float)GS_TILE_HEIGHT, -lim), lim));
    r1 = (int)floorf(fminf(fmaxf((v + up + slack - 0.5f) / (float)GS_TILE_HEIGHT, -lim), lim)) + 1;
}

// [x0, x1] x [y0, y1]: pixel-CENTRE coordinates of the rectangle's corner pixels.  sx = -B/A, sy = -B/C: the slopes
// of the conic's conjugate diameters (per-Gaussian constants; an approximate reciprocal is enough, see above).
__device__ __forceinline__ bool gs_rect_may_contribute(float ux, float uy, float A, float B, float C, float sx,
                                                       float sy, float qmax, float x0, float x1, float y0, float y1) {
#pragma clang fp contract(off)
    const float dxc = fminf(fmaxf(ux, x0), x1) - ux;  // x offset of the closest point, 0 if inside the span
    const float dyc = fminf(fmaxf(uy, y0), y1) - uy;
    float qmin = 0.f;
    if (dxc != 0.f || dyc != 0.f) {
        qmin = 3.0e38f;
        if (dxc != 0.f) {  // edge x = const facing the centre: minimise over dy along the edge (slope -B/C)
            const float dy = fminf(fmaxf(sy * dxc, y0 - uy), y1 - uy);
            qmin = A * dxc * dxc + 2.f * B * dxc * dy + C * dy * dy;
        }
        if (dyc != 0.f) {
            const float dx = fminf(fmaxf(sx * dyc, x0 - ux), x1 - ux);
            qmin = fminf(qmin, A * dx * dx + 2.f * B * dx * dyc + C * dyc * dyc);
        }
    }
    return !(qmin > qmax);  // NaN-safe: anything unordered is kept
}

// Staging filter of the blend kernels: does list entry (rows 0 and 1 of its record) belong to tile (tile_u, tile_v)?
//   GS_FILTER_BOX : the tile lies inside the Gaussian's tile box (RAS:81-103) -- the reference emits a key for exactly
//                   those tiles (RAS:131-172); needed whenever a list covers more than one tile (bins)
//   GS_FILTER_CULL: the exact contribution test above on the tile's 16 x 16 pixels
__device__ __forceinline__ bool gs_entry_in_tile(const float4 r0, const float4 r1, int tile_u, int tile_v, int tw,
                                                 int th, int filter) {
    bool keep = true;
    if (filter & GS_FILTER_BOX) {
        int t0u, t1u, t0v, t1v;
        gs_tile_box(r0.x, r0.y, r1.w, tw, th, t0u, t1u, t0v, t1v);
        keep = tile_u >= t0u && tile_u < t1u && tile_v >= t0v && tile_v < t1v;
    }
    if (filter & GS_FILTER_CULL) {
        const float x0 = (float)(tile_u * GS_TILE_WIDTH) + 0.5f, y0 = (float)(tile_v * GS_TILE_HEIGHT) + 0.5f;
        keep = keep && gs_rect_may_contribute(r0.x, r0.y, r1.x, r1.y, r1.z, -r1.y * __builtin_amdgcn_rcpf(r1.x),
                                              -r1.y * __builtin_amdgcn_rcpf(r1.z), r0.w, x0,
                                              x0 + (float)(GS_TILE_WIDTH - 1), y0, y0 + (float)(GS_TILE_HEIGHT - 1));
    }
    return keep;
}

__device__ __forceinline__ int gs_lane() { return threadIdx.x & (GS_WAVE - 1); }
// value of lane `l` (wave-uniform index) for every lane
__device__ __forceinline__ float gs_readlane_f(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

// number of set bits of `mask` strictly below the calling lane
__device__ __forceinline__ int gs_mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// inclusive scan of an int across the 64 lanes of a wave
__device__ __forceinline__ int gs_wave_incl_scan(int v) {
    const int lane = gs_lane();
#pragma unroll
    for (int d = 1; d < GS_WAVE; d <<= 1) {
        int o = __shfl_up(v, d, GS_WAVE);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan across a 256-thread block; *total receives the block sum (all threads).
// lds must hold 4 ints.  Contains two barriers.
__device__ __forceinline__ int gs_block_excl_scan(int v, int *total, int *lds) {
    const int w = threadIdx.x >> 6;
    int incl = gs_wave_incl_scan(v);
    if (gs_lane() == GS_WAVE - 1) lds[w] = incl;
    __syncthreads();
    int base = 0, t = 0;
#pragma unroll
    for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) {
        int s = lds[i];
        if (i < w) base += s;
        t += s;
    }
    __syncthreads();
    *total = t;
    return base + incl - v;
}

// DPP move helper (row_shr / row_bcast patterns of the GCN/CDNA DPP unit)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float gs_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, BANK_MASK, true));
}
// Sum over the 64 lanes; the total is valid in lane 63 (read it with gs_readlane63).
__device__ __forceinline__ float gs_wave_sum_to_lane63(float v) {
    v += gs_dpp<0x111, 0xf, 0xf>(v);  // row_shr:1
    v += gs_dpp<0x112, 0xf, 0xf>(v);  // row_shr:2
    v += gs_dpp<0x114, 0xf, 0xe>(v);  // row_shr:4, banks 1-3
    v += gs_dpp<0x118, 0xf, 0xc>(v);  // row_shr:8, banks 2-3
    v += gs_dpp<0x142, 0xa, 0xf>(v);  // row_bcast:15 -> rows 1,3
    v += gs_dpp<0x143, 0xc, 0xf>(v);  // row_bcast:31 -> rows 2,3
    return v;
}
// Sum within each row of 16 lanes; the row total is valid in lane 15 of the row.
__device__ __forceinline__ float gs_row_sum_to_lane15(float v) {
    v += gs_dpp<0x111, 0xf, 0xf>(v);  // row_shr:1
    v += gs_dpp<0x112, 0xf, 0xf>(v);  // row_shr:2
    v += gs_dpp<0x114, 0xf, 0xe>(v);  // row_shr:4, banks 1-3
    v += gs_dpp<0x118, 0xf, 0xc>(v);  // row_shr:8, banks 2-3
    return v;
}
// ------------------------------------------------------------------ forward colour of a Gaussian, shared source
// sigmoid(SH . Y(view direction)) exactly as gs_preprocess stores it in row 2 of the packed record.  It lives here,
// with FMA contraction switched off inside every function, so that the two kernels that evaluate it -- the projection
// kernel and (for Gaussians that emitted no key on this GPU, tile-row sharding) the per-point backward -- execute the
// same instruction sequence and agree to the last bit: ranks that take different paths must still end up with
// identical replicated gradients.
__device__ __forceinline__ void gs_rotmat_from_q(float x, float y, float z, float w, float R[9]) {  // GP3:31-48
#pragma clang fp contract(off)
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (yy + zz); R[1] = 2.f * (xy - wz); R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz); R[4] = 1.f - 2.f * (xx + zz); R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy); R[7] = 2.f * (yz + wx); R[8] = 1.f - 2.f * (xx + yy);
}
__device__ __forceinline__ void gs_sh_basis(const float d[3], float Y[16]) {  // SPH:10-32
#pragma clang fp contract(off)
    float n = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
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
// W = R(q_camera_pointcloud), t = t_camera_pointcloud, p = the point; coeff(ch, k) = SH coefficient k of channel ch.
// Ray origin = (-W^T) t (UTL:495-510), colour = sigmoid(SH . Y) (RAS:280-282,302-310, GP3:333-349).
// The 16-term dot product is summed as FOUR QUARTERS of four sequential terms, combined as (q0 + q1) + (q2 + q3): the
// projection kernel evaluates a quarter per lane (twelve lanes read a Gaussian's 192 B of coefficients with one coalesced
// 16-byte load each, gs_preprocess) and adds the quarters in exactly this tree, so the cooperative and the scalar
// evaluation agree to the last bit.  (The reference sums the sixteen products in sequence: the CPU oracle does too, and
// the two differ by an ulp of the pre-sigmoid sum -- well inside the colour bar of the parity tests.)
__device__ __forceinline__ void gs_view_basis(const float W[9], const float t[3], const float p[3], float Y[16]) {
#pragma clang fp contract(off)
    float ro[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ro[k] = ((-W[k]) * t[0] + (-W[3 + k]) * t[1]) + (-W[6 + k]) * t[2];
    const float dir[3] = {p[0] - ro[0], p[1] - ro[1], p[2] - ro[2]};
    gs_sh_basis(dir, Y);
}
__device__ __forceinline__ float gs_sh_quarter(float c0, float c1, float c2, float c3, float y0, float y1, float y2,
                                               float y3) {
#pragma clang fp contract(off)
    float s = c0 * y0;
    s = s + c1 * y1;
    s = s + c2 * y2;
    s = s + c3 * y3;
    return s;
}
__device__ __forceinline__ float gs_colour_from_sum(float s) {
#pragma clang fp contract(off)
    return 1.f / (1.f + expf(-s));
}
template <typename Coeff>
__device__ __forceinline__ void gs_view_colour(const float W[9], const float t[3], const float p[3], Coeff coeff,
                                               float rgb[3]) {
#pragma clang fp contract(off)
    float Y[16];
    gs_view_basis(W, t, p, Y);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            q[j] = gs_sh_quarter(coeff(ch, 4 * j), coeff(ch, 4 * j + 1), coeff(ch, 4 * j + 2), coeff(ch, 4 * j + 3), Y[4 * j],
                                 Y[4 * j + 1], Y[4 * j + 2], Y[4 * j + 3]);
        rgb[ch] = gs_colour_from_sum((q[0] + q[1]) + (q[2] + q[3]));
    }
}

// ------------------------------------------------------------------ streaming accesses
// The non-temporal hint for the two big streams of a frame that nothing on the device reads again: the 192 B of SH
// coefficients per Gaussian the projection reads (gs_wave_view_colours) and the dense + compact gradient rows the per-point
// backward writes (gs_rows_lds_to_global: 428 MB at the headline size).  Without it they evict what IS read again -- the
// records, the next frame's rows -- from L2 and the memory-side cache.  Measured (round 6, same box, rocprofv3 averages):
// preprocess_kernel 113 -> 94-97 us, point_backward_kernel 130 -> 119-126 us, frame 1.077-1.082 -> 1.050-1.058 ms.
// NOT for data the next kernel reads: the slot records with the hint made reduce_partials_kernel 52 -> 67 us (they come
// back out of the memory-side cache otherwise); accumulator rows, record rows and the per-pixel outputs: no gain or a loss
// (profiles/r06_streaming_hints.md).
typedef float gs_f4_native __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gs_store_stream(float4 *p, const float4 v) {
    const gs_f4_native nv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(nv, reinterpret_cast<gs_f4_native *>(p));
}
__device__ __forceinline__ float4 gs_load_stream(const float4 *p) {
    const gs_f4_native nv = __builtin_nontemporal_load(reinterpret_cast<const gs_f4_native *>(p));
    return make_float4(nv.x, nv.y, nv.z, nv.w);
}

// ------------------------------------------------------------------ coalesced access to 224-B feature rows
// The feature matrix is AoS (56 floats = 14 x 16 B per Gaussian, owned by the caller).  A lane writing its own
// row with 16-B stores makes every store instruction touch 64 different cache lines.  Instead the wave moves its
// 64 rows cooperatively: every lane puts its row into LDS, then 14 consecutive lanes write the 14 float4 of one row
// (4 rows = 56 lanes per instruction, 16 instructions per wave).  (The reading direction was measured 6 % slower in
// gs_preprocess -- it lowers the occupancy of a kernel that lives on memory-level parallelism -- and is not kept.)
// rows = this wave's LDS staging area: 64 rows x GS_ROW_F4 float4 (the 15th float4 pads the row to 240 B,
// which keeps the per-lane ds_read_b128 of one 16-lane group on distinct banks).
#define GS_ROW_F4 15
__device__ __forceinline__ void gs_rows_lds_to_global(float4 *__restrict__ base, int my_row_id,
                                                      const float4 *__restrict__ rows) {
    __builtin_amdgcn_wave_barrier();
    const int lane = gs_lane(), sub = lane / 14, c = lane - 14 * sub;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int rr = it * 4 + sub;
        const int rid = __shfl(my_row_id, rr & 63, GS_WAVE);
        if (lane < 56 && rid >= 0) gs_store_stream(base + (size_t)rid * 14 + c, rows[rr * GS_ROW_F4 + c]);
    }
}

// ------------------------------------------------------------------ 12-value wave reduce-scatter
// Sums twelve per-lane partials over the 64 lanes of a wave in 34 VALU instructions (6 x 12 = 72 with
// the plain DPP ladder): two swap+add levels use gfx950's v_permlane32_swap / v_permlane16_swap to
// halve the number of live registers (12 -> 6 -> 3), then four DPP row steps finish each register (row_ror:8 first, then
// row_shr:1/2/4 over the 8-lane halves: the SAME addition tree as gs_wave_reduce12_pair below, so an entry's sums do
// not depend on which of the two functions reduced it).
// On return the totals sit in lane 15 of each 16-lane row:
//   t0: rows 0..3 = (x0, x2, x1, x3)   t1: rows = (x4, x6, x5, x7)   t2: rows = (x8, x10, x9, x11)
// Hand-scheduled inline asm because (a) ROCm 7.2's clang returns element 0 for BOTH results of the
// __builtin_amdgcn_permlane*_swap builtins and (b) hazards are not visible through an asm statement:
// the leading s_nop covers "VALU write -> permlane read"; every other dependent pair below is separated
// by >= 2 independent instructions (DPP / permlane reads of a just-written VGPR need 2 wait states).
// Swap semantics verified on hardware: v_permlane32_swap exchanges vdst[63:32] with src0[31:0];
// v_permlane16_swap exchanges the odd rows of vdst with the even rows of src0.
__device__ __forceinline__ void gs_wave_reduce12(float x0, float x1, float x2, float x3, float x4, float x5,
                                                 float x6, float x7, float x8, float x9, float x10, float x11,
                                                 float &t0, float &t1, float &t2) {
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %1\n\t"
        "v_permlane32_swap_b32 %2, %3\n\t"
        "v_permlane32_swap_b32 %4, %5\n\t"
        "v_permlane32_swap_b32 %6, %7\n\t"
        "v_permlane32_swap_b32 %8, %9\n\t"
        "v_permlane32_swap_b32 %10, %11\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_add_f32 %2, %2, %3\n\t"
        "v_add_f32 %4, %4, %5\n\t"
        "v_add_f32 %6, %6, %7\n\t"
        "v_add_f32 %8, %8, %9\n\t"
        "v_add_f32 %10, %10, %11\n\t"
        "v_permlane16_swap_b32 %0, %2\n\t"
        "v_permlane16_swap_b32 %4, %6\n\t"
        "v_permlane16_swap_b32 %8, %10\n\t"
        "v_add_f32 %0, %0, %2\n\t"
        "v_add_f32 %4, %4, %6\n\t"
        "v_add_f32 %8, %8, %10\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %8, %8, %8 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %8, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9),
          "+v"(x10), "+v"(x11));
    t0 = x0; t1 = x4; t2 = x8;
}
// Two list entries at once (the backward blend keeps one hit entry pending): 2 x 12 per-lane partials P (entry A) and
// Q (entry B) -> 50 VALU instructions instead of 2 x 34.  Level 1 swaps ACROSS the entries (v_permlane32_swap P[j],
// Q[j]: lanes 0-31 then carry entry A, lanes 32-63 entry B), level 2 pairs values with v_permlane16_swap, level 3
// packs two registers into one with bank-masked row_ror:8 adds, then three row_shr steps finish the 8-lane groups.
// Value 11 of both entries must be 0 (it is never moved).  On return, for lane l with (l & 7) == 7, row r = l >> 4,
// half h = (l >> 3) & 1:   w_k (k = 0, 1, 2) holds the wave total of value 4 k + 2 h + (r & 1) of entry (r >> 1).
// Hazards as in gs_wave_reduce12: every DPP / permlane read of a freshly written VGPR is >= 2 instructions away.
__device__ __forceinline__ void gs_wave_reduce12_pair(float (&P)[12], float (&Q)[12], float &w0, float &w1, float &w2) {
    float zero = 0.f;
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %12\n\t"
        "v_permlane32_swap_b32 %1, %13\n\t"
        "v_permlane32_swap_b32 %2, %14\n\t"
        "v_permlane32_swap_b32 %3, %15\n\t"
        "v_permlane32_swap_b32 %4, %16\n\t"
        "v_permlane32_swap_b32 %5, %17\n\t"
        "v_permlane32_swap_b32 %6, %18\n\t"
        "v_permlane32_swap_b32 %7, %19\n\t"
        "v_permlane32_swap_b32 %8, %20\n\t"
        "v_permlane32_swap_b32 %9, %21\n\t"
        "v_permlane32_swap_b32 %10, %22\n\t"
        "v_add_f32 %0, %0, %12\n\t"
        "v_add_f32 %1, %1, %13\n\t"
        "v_add_f32 %2, %2, %14\n\t"
        "v_add_f32 %3, %3, %15\n\t"
        "v_add_f32 %4, %4, %16\n\t"
        "v_add_f32 %5, %5, %17\n\t"
        "v_add_f32 %6, %6, %18\n\t"
        "v_add_f32 %7, %7, %19\n\t"
        "v_add_f32 %8, %8, %20\n\t"
        "v_add_f32 %9, %9, %21\n\t"
        "v_add_f32 %10, %10, %22\n\t"
        "v_permlane16_swap_b32 %0, %1\n\t"
        "v_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %4, %5\n\t"
        "v_permlane16_swap_b32 %6, %7\n\t"
        "v_permlane16_swap_b32 %8, %9\n\t"
        "v_permlane16_swap_b32 %10, %11\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_add_f32 %2, %2, %3\n\t"
        "v_add_f32 %4, %4, %5\n\t"
        "v_add_f32 %6, %6, %7\n\t"
        "v_add_f32 %8, %8, %9\n\t"
        "v_add_f32 %10, %10, %11\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %8, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %8, %8, %8 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %8, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1"
        : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7]), "+v"(P[8]),
          "+v"(P[9]), "+v"(P[10]), "+v"(zero), "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3]), "+v"(Q[4]), "+v"(Q[5]),
          "+v"(Q[6]), "+v"(Q[7]), "+v"(Q[8]), "+v"(Q[9]), "+v"(Q[10]));
    w0 = P[0]; w1 = P[4]; w2 = P[8];
}
// The same twelve sums with the matrix pipe doing most of the cross-lane work (the pipe is otherwise idle in the blend
// kernels; f32-input MFMA is exact: a k-ordered fmaf chain).  One permlane32-swap level leaves six registers that hold
// value 2k in lanes 0-31 and value 2k+1 in lanes 32-63; v_mfma_f32_16x16x4_f32 with that register as A (A[m][kk] =
// lane 16 kk + m) and a 0/1 selector as B (B[kk][n] = [n == 2k + (kk >> 1)]) adds lanes m, m+16 (or m+32, m+48) into
// column n of a 16 x 16 accumulator; after the six accumulating MFMAs the four accumulator registers of a lane are added
// (rows 4g .. 4g+3 of column n = lane & 15) and a second MFMA with A = 1 sums the four row groups: every lane with
// (lane & 15) == n holds the wave total of value n.  15 VALU instructions + 7 MFMA instead of 30 VALU.
typedef float gs_v4f __attribute__((ext_vector_type(4)));
struct GsMfmaReduceConsts { float sel[6]; float one; };
__device__ __forceinline__ GsMfmaReduceConsts gs_mfma_reduce_consts() {
    GsMfmaReduceConsts c;
    const int lane = gs_lane(), n = lane & 15, hi = lane >> 5;
#pragma unroll
    for (int k = 0; k < 6; ++k) c.sel[k] = (n == 2 * k + hi) ? 1.f : 0.f;
    c.one = 1.f;
    return c;
}
__device__ __forceinline__ float gs_wave_reduce12_mfma(float x0, float x1, float x2, float x3, float x4, float x5,
                                                       float x6, float x7, float x8, float x9, float x10, float x11,
                                                       const GsMfmaReduceConsts &c) {
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %1\n\t"
        "v_permlane32_swap_b32 %2, %3\n\t"
        "v_permlane32_swap_b32 %4, %5\n\t"
        "v_permlane32_swap_b32 %6, %7\n\t"
        "v_permlane32_swap_b32 %8, %9\n\t"
        "v_permlane32_swap_b32 %10, %11\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_add_f32 %2, %2, %3\n\t"
        "v_add_f32 %4, %4, %5\n\t"
        "v_add_f32 %6, %6, %7\n\t"
        "v_add_f32 %8, %8, %9\n\t"
        "v_add_f32 %10, %10, %11\n\t"
        "s_nop 1"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9),
          "+v"(x10), "+v"(x11));
    gs_v4f d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, c.sel[0], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, c.sel[1], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(x4, c.sel[2], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(x6, c.sel[3], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(x8, c.sel[4], d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(x10, c.sel[5], d, 0, 0, 0);
    const float t = (d[0] + d[1]) + (d[2] + d[3]);
    const gs_v4f z = {0.f, 0.f, 0.f, 0.f};
    const gs_v4f e = __builtin_amdgcn_mfma_f32_16x16x4f32(c.one, t, z, 0, 0, 0);
    return e[0];
}
__device__ __forceinline__ float gs_readlane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

#endif  // __HIPCC__
