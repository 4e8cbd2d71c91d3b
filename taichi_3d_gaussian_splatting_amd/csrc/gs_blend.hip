// gs_blend.hip -- per-tile alpha blending, forward and backward, for gfx950 (wave64).
//
// Workgroup = one 16x16 tile = 2 wavefronts; every lane owns two horizontally adjacent pixels (a wave
// covers 8 rows x 16 columns).  Lists are sorted per BIN of (1 << bin_shift)^2 tiles (gs_make_keys); a tile walks its
// bin's depth-sorted list, keeps the entries that belong to it (tile box of RAS:81-103 + exact contribution test,
// gs_entry_in_tile) with an ORDER-PRESERVING compaction (wave ballots + mbcnt) and stages up to 128 packed records
// (16-B gathers) in LDS per blend round -- the sequence a tile blends is exactly the reference's per-tile list.
//  forward : front-to-back blend (RAS:318-485, weight UTL:275-284); 2 waves per tile, two pixels per lane
//            on packed fp32 math; whole-tile early exit with a workgroup vote (the reference could not
//            express it, RAS:387-394).  With a binned layout it also writes out the entries each tile keeps
//            (walked_list): the backward pass then walks plain per-tile lists and never filters a bin list again.
//  backward: back-to-front traversal (RAS:531-705, gradients UTL:331-348) starting at the tile's last
//            effective entry; 2 waves per tile, two pixels per lane; alpha is evaluated by the SAME expression as in
//            the forward kernel (gs_pair_alpha below), so the hit test alpha >= 1/255 decides identically in both
//            passes; the per-Gaussian partial sums (ten gradients + the pixel count) are reduced across the 64 lanes
//            by a permlane-swap + DPP reduce-scatter (two hit entries sharing one reduce-scatter, a
//            round's odd one alone), combined across the two waves in LDS (ds_add_f32), and stored
//            once per (tile, Gaussian) into that pair's private slot -- no global atomics at all (the reference issues
//            eleven per (pixel, Gaussian), RAS:674-696); the slots of a Gaussian are summed in a fixed order by
//            reduce_partials_kernel, so gradients are bitwise reproducible.
//  order   : tiles are handed to the hardware longest list / longest backward walk first (tile_order_kernel), which
//            removes the tail of half-empty CUs of a launch in image order.
//  small   : grids of at most 3,840 tiles run both passes with FOUR waves per tile and one pixel per lane
//            (blend_*_small_kernel): same per-pixel arithmetic, half the issue slots per wave and entry.
#include "gs_common.h"
#include <stdlib.h>
#include "gs_slots.h"

// Automatic FMA contraction is off in this file and every fused multiply-add is written explicitly:
// the unrolled copies of the inner loops then execute the same instruction sequence for a Gaussian
// whatever its position in the tile list, so results do not depend on list positions (e.g. the exact
// tile cull, which only removes non-contributing entries, leaves every output bit-identical).
#pragma clang fp contract(off)

namespace {

constexpr float EPS_ALPHA = (float)(1.0 / 255.0);  // RAS:451
constexpr float CLAMP_ALPHA = 0.99f;               // RAS:453
constexpr float STOP_T = 0.0001f;                  // RAS:458
// MEASUREMENT ARMS (GS_ABLATE_FWD, GS_BWD_REDUCE_ARM = 0, GS_MFMA_REDUCE, GS_STATS) exist in tuning builds only: they are
// honoured when GS_TUNING_BUILD is defined -- which also changes gs_abi_version() (gs_api.hip), so that _lib.load refuses
// such a library unless the measuring tool asks for it (GS_ALLOW_TUNING_LIB=1) -- and are a compile error otherwise: one
// stray -D cannot ship a library that blends nothing or throws its gradients away.
#ifndef GS_TUNING_BUILD
#if defined(GS_ABLATE_FWD) || defined(GS_MFMA_REDUCE) || defined(GS_STATS) || \
    (defined(GS_BWD_REDUCE_ARM) && GS_BWD_REDUCE_ARM == 0) || (defined(GS_BWD_REDUCE_STAGED) && GS_BWD_REDUCE_STAGED == 0) || \
    (defined(GS_BWD_REDUCE_DIRECT) && GS_BWD_REDUCE_DIRECT == 0)
#error "GS_ABLATE_FWD / GS_MFMA_REDUCE / GS_STATS / GS_BWD_REDUCE_* = 0 are measurement arms: build them with -DGS_TUNING_BUILD=1 (tools/build_variants.sh)"
#endif
#endif
#ifndef GS_ABLATE_FWD
#define GS_ABLATE_FWD 0   // 1: forward evaluates alpha but blends nothing (tools/build_variants.sh, measurements only)
#endif
#ifndef GS_GROUP_FWD
#define GS_GROUP_FWD 4
#endif
#ifndef GS_FWD_WHOLE_Q
#define GS_FWD_WHOLE_Q 0   // 1: (amp, depth) arrive with (B, e_lo) in front of the hit test and wait in registers instead of being read
                           // on the hit path.  Measured slower (forward stage 0.260 -> 0.268 ms in groups of four -- 36 B of scratch --
                           // and 0.270 in groups of two, same box, alternating runs)
#endif
#ifndef GS_FWD_UPFRONT
#define GS_FWD_UPFRONT 0   // 1: the four exponents of a group are pinned in front of the first hit test.  Measured slower (forward
                           // stage 0.260 -> 0.269 ms, same box, alternating runs; GS_GROUP_FWD = 2: 0.269 either way): left alone the
                           // compiler sinks each evaluation to its own test, which keeps fewer values live
#endif
#ifndef GS_GROUP_BWD
#define GS_GROUP_BWD 4
#endif
#ifndef GS_MFMA_REDUCE
#define GS_MFMA_REDUCE 0     // 1: the backward's cross-lane sums go through the matrix pipe (gs_wave_reduce12_mfma)
#endif
#ifndef GS_BWD_MIN_WAVES
#define GS_BWD_MIN_WAVES 1   // second argument of __launch_bounds__ (minimum waves per SIMD) of the backward kernel
#endif
// Cross-lane sums of the backward: 1 = per hit entry (gs_wave_reduce12, 34 instructions), 2 = two hit entries at a time
// (gs_wave_reduce12_pair, 50 per pair; one entry stays pending in registers: +15 VGPRs, four waves per SIMD instead of
// five), 0 = none (measurement only: the partials are thrown away, gradients are wrong).  Same bits either way.  Measured
// at the headline size (profiles/r03_exp1_backward_arms.txt): binned lists 0.489 -> 0.468 ms with pairs, per-tile lists
// 0.460 -> 0.457 (the lost wave costs what the shorter reduction saves).  Both kernels pair.  (History: the one-switch
// macro of the variant builds used to be called GS_BWD_REDUCE -- since the frame entry points, the name of a stage bit
// in include/gsplat_hip.h with the value 2, so BOTH kernels have been built with pairs since then and every round-4
// measurement is of that build; the switch is now GS_BWD_REDUCE_ARM and the direct kernel's default says 2.)
// Round 6: 3 = through LDS (the default now: see the REDUCE == 3 arm), 4 = the same with the read-back deferred to the next
// hit entry (measured slower: the kernel is not waiting for the LDS, it is short of LDS cycles).
#ifndef GS_BWD_REDUCE_STAGED
#define GS_BWD_REDUCE_STAGED 3
#endif
#ifndef GS_BWD_REDUCE_DIRECT
#define GS_BWD_REDUCE_DIRECT 3
#endif
// The form for frames whose walk lengths are SKEWED (flag GS_BLEND_SKEWED_WALKS of gs_blend_backward: the caller's call -- the
// operator samples the walk lengths the forward pass records every few frames).  A trained scene -- mean 125, max 1,316 list
// positions walked per tile -- ends in a tail of a few long tiles, and what counts there is the dependent chain per hit
// entry, not the SIMD's throughput: the register reduce-scatter (2) finishes the trained 1920 x 1072 scene's backward blend
// in 0.398 ms where the LDS form (3) needs 0.448, while on evenly loaded frames (headline: mean 292, max 749; cfg 3, cfg 4)
// the LDS form is 5-7 % faster (profiles/r06_backward_arms.md).  (Both forms in ONE kernel behind a wave-uniform branch:
// 125 VGPRs with scratch, slower than either; two launches with a device-side mark: 5 us for the launch that does nothing.)
#ifndef GS_BWD_REDUCE_SKEWED
#define GS_BWD_REDUCE_SKEWED 2
#endif
#ifdef GS_BWD_REDUCE_ARM      // (one switch for both kernels: tools/build_variants.sh)
#undef GS_BWD_REDUCE_STAGED
#undef GS_BWD_REDUCE_DIRECT
#define GS_BWD_REDUCE_STAGED GS_BWD_REDUCE_ARM
#define GS_BWD_REDUCE_DIRECT GS_BWD_REDUCE_ARM
#endif
// Path statistics of the two-wave blend kernels (tuning builds only, -DGS_STATS=1: tools/blend_stats.py): how often the
// hit path runs, how many of its lanes are live, how often the careful paths are entered.  One 64-bit atomic per wave and
// event; gs_blend_read_stats reads (and clears) the counters -- all zero in the product build.
#ifndef GS_STATS
#define GS_STATS 0
#endif
#if GS_STATS
__device__ unsigned long long gs_blend_stats_dev[GS_BLEND_STATS];
#define GS_STAT(i, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&gs_blend_stats_dev[i], (unsigned long long)(n)); } while (0)
#else
#define GS_STAT(i, n) do { } while (0)
#endif
#ifndef GS_FWD_MIN_WAVES
#define GS_FWD_MIN_WAVES 6   // second argument of __launch_bounds__ of the forward kernel: six waves per SIMD (80 registers).  Seven
                             // (72 registers) is met only with 24-32 B of scratch for no gain (0.272 vs 0.270 ms at the headline size)
#endif
constexpr int GS_TR_ROWS = 11, GS_TR_STRIDE = 68;   // backward, REDUCE 3: rows of the LDS transpose and their distance in floats
constexpr int GROUP = GS_GROUP_FWD > GS_GROUP_BWD ? (GS_GROUP_FWD > 4 ? GS_GROUP_FWD : 4) : (GS_GROUP_BWD > 4 ? GS_GROUP_BWD : 4);
                              // padding granularity of a staged batch (BATCH % GROUP == 0; covers both group sizes)
constexpr int GROUP_FWD = GS_GROUP_FWD;  // list entries evaluated together in the forward blend loop
constexpr int GROUP_BWD = GS_GROUP_BWD;  // ... in the backward loop

struct TileCoord { int tile_u, tile_v, tile_id, index; };   // index: n-th owned tile (row-major over the owned rows)

// blockIdx -> owned tile.  Consecutive workgroups are dispatched round-robin over the 8 XCDs
// (block b -> XCD b%8); the remap gives every XCD a contiguous run of tiles so that neighbouring
// tiles (which share Gaussians) hit the same 4-MiB L2.  Speed only, never correctness.
#ifndef GS_XCD_CHUNK
#define GS_XCD_CHUNK 0   // 0: one contiguous run of tiles per XCD; C > 0: runs of C tiles dealt round-robin to the XCDs.
                         // Measured (tools/xcd_sweep.sh): C = 8, 120, 480 equal the default within 1 %; C = 30 (a fixed
                         // quarter of every tile row per XCD) is 25 % slower -- XCD load balance matters, L2 locality less.
#endif
__device__ __forceinline__ TileCoord owned_tile_at(int b, int nb, int tw, int row_begin, int row_step,
                                                   const int32_t *__restrict__ tile_order);
__device__ __forceinline__ TileCoord owned_tile(int tw, int row_begin, int row_step,
                                                const int32_t *__restrict__ tile_order = nullptr) {
    return owned_tile_at(blockIdx.x, gridDim.x, tw, row_begin, row_step, tile_order);
}
// (b-th of nb workgroups: kernels that give a tile several workgroups pass blockIdx / split and gridDim / split)
__device__ __forceinline__ TileCoord owned_tile_at(int b, int nb, int tw, int row_begin, int row_step,
                                                   const int32_t *__restrict__ tile_order) {
    if (tile_order != nullptr) {
        b = tile_order[b];   // dispatch order (longest walks first, tile_order_kernel): a permutation of the owned tiles
    } else if (GS_XCD_CHUNK > 0) {
        const int group = 8 * GS_XCD_CHUNK, full = (nb / group) * group;
        if (b < full) {
            const int x = b % 8, j = b / 8;   // XCD, position in that XCD's dispatch order
            b = ((j / GS_XCD_CHUNK) * 8 + x) * GS_XCD_CHUNK + (j % GS_XCD_CHUNK);
        }
    } else {
        const int chunk = nb / 8;
        if (b < chunk * 8) b = (b % 8) * chunk + b / 8;
    }
    TileCoord t;
    t.index = b;
    t.tile_u = b % tw;
    t.tile_v = row_begin + (b / tw) * row_step;
    t.tile_id = t.tile_u + t.tile_v * tw;
    return t;
}

// ------------------------------------------------------------------------------- the Gaussian weight
// Workgroup = one tile = 2 wave64s; every lane owns TWO horizontally adjacent pixels and does its fp32 arithmetic on float2
// values, which the compiler maps to the packed CDNA instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two pixels
// per VALU issue).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float x) { return (v2f){x, x}; }
__device__ __forceinline__ unsigned long long gs_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
constexpr float GS_LOG2E = 1.4426950408889634f;

// What a staged list entry leaves in LDS for the group loops:
//   P = (u, v, A, C)                     one 16-B broadcast read per entry and wave, with (B, lim) -- 8 B -- in front of the hit test
//   forward:  Q = (B, e_lo, amp, depth)  colour row (r, g, b, W)     W = the Gaussian's stop-bracket weight (gs_stop_weight, float 13)
//   backward: Q = (B, amp, radius, s_hi) colour row (r, g, b, opacity)   (one 16-B read; the rescale factor, which only the
//             exact re-evaluation of an alpha next to 1/255 wants, is fetched from the record itself there)
//   amp = opacity * rescale, formed by gs_preprocess (float 12 of the record); e_lo = the exponent below which the reference
//   skips the pair for certain (float 14: gs_common.h, "the 1/255 decision in the exponent's domain"); s_hi = -2 e_lo, the same
//   bound on the backward pass' quadratic form s = -2 e.  Forward: the second half of Q and the colour row are read on the
//   hit path only.
//   (C, which the packed arithmetic only ever uses as a per-lane scalar, sits in the one position -- the upper half of the
//   second register pair -- from which the compiler will not broadcast without a v_mov)
// THE EXPONENT IS THE REFERENCE'S TO THE LAST BIT (gs_common.h, "threshold decisions"): each pass evaluates the quadratic
// form in the operation order of its counterpart, contraction off (this file) --
//   forward  UTL:281-283   e = -0.5 * (dx*dx*A + dy*dy*C) - dx*dy*B        (the final fma rounds once, as the reference's
//                                                                           subtraction does: -0.5 * t is exact)
//   backward UTL:336-339   m = conic @ d,  e = -0.5 * (dx*m0 + dy*m1)      (m is also what the gradients need, UTL:343-346)
// -- and alpha = 2^(e * log2 e) * amp: four packed instructions more per pixel pair than the pre-scaled log2-domain form of
// rounds 1-4 (two in the backward pass, which gets m for free), for an alpha within u (2 |e| + 9) of the reference's
// whatever the conic, instead of 9 u times the cancellation inside the quadratic form.
// Round 6: the exponent is compared with e_lo FIRST, and the exponential (two v_exp_f32 and two packed products per pixel
// pair: 14 of the 35 ns a visit without a hit cost the SIMD) is evaluated on the hit path only -- a quarter of the (wave,
// entry) visits never get there.
__device__ __forceinline__ v2f gs_weight_from_exponent(v2f e, float amp) {
    const v2f e2 = e * splat(GS_LOG2E);
    return (v2f){__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)} * splat(amp);
}
__device__ __forceinline__ v2f gs_pair_exponent_forward(const float4 P, const float B, const v2f px, const float py) {
    const v2f dx = px - splat(P.x);
    const float dy = py - P.y;
    const v2f t = (dx * dx) * splat(P.z) + splat((dy * dy) * P.w);
    const v2f x = (dx * splat(dy)) * splat(B);
    return fma2(splat(-0.5f), t, -x);
}
__device__ __forceinline__ v2f gs_pair_alpha_forward(const float4 P, const float4 Q, const v2f px, const float py, v2f &e) {
    e = gs_pair_exponent_forward(P, Q.x, px, py);
    return gs_weight_from_exponent(e, Q.z);
}
// the backward pass' quadratic form s = d . (conic @ d) = -2 e (UTL:336-339) and m = conic @ d
__device__ __forceinline__ v2f gs_pair_form_backward(const float4 P, const float B, const v2f px, const float py, v2f &m0, v2f &m1) {
    const v2f dx = px - splat(P.x);
    const float dy = py - P.y;
    m0 = splat(P.z) * dx + splat(B * dy);
    m1 = splat(B) * dx + splat(P.w * dy);
    return dx * m0 + splat(dy) * m1;
}
// 2^(e log2 e) amp with e log2 e formed as s * (-0.5 log2 e): the same bits as (-0.5 s) * log2 e -- halving is exact -- in one
// multiplication
__device__ __forceinline__ v2f gs_pair_alpha_from_form(const v2f s, const float amp) {
    const v2f e2 = s * splat(-0.5f * GS_LOG2E);
    return (v2f){__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)} * splat(amp);
}
__device__ __forceinline__ v2f gs_pair_alpha_backward(const float4 P, const float4 Q, const v2f px, const float py, v2f &e,
                                                      v2f &m0, v2f &m1) {
    const v2f s = gs_pair_form_backward(P, Q.x, px, py, m0, m1);
    e = splat(-0.5f) * s;   // (exact: only the careful path looks at it)
    return gs_pair_alpha_from_form(s, Q.y);
}
// (one pixel per lane: the four-waves-per-tile kernels and the replay; the same operations, component by component)
__device__ __forceinline__ float gs_exponent_forward(float dx, float dy, float A, float B, float C) {
    const float t = (dx * dx) * A + (dy * dy) * C;
    return __builtin_fmaf(-0.5f, t, -((dx * dy) * B));
}
__device__ __forceinline__ float gs_pixel_alpha_forward(const float4 P, const float4 Q, float px, float py, float &e) {
    e = gs_exponent_forward(px - P.x, py - P.y, P.z, Q.x, P.w);
    return __builtin_amdgcn_exp2f(e * GS_LOG2E) * Q.z;
}
__device__ __forceinline__ float gs_pixel_alpha_backward(const float4 P, const float4 Q, float px, float py, float &e, float &m0,
                                                         float &m1) {
    const float dx = px - P.x, dy = py - P.y;
    m0 = P.z * dx + Q.x * dy;
    m1 = Q.x * dx + P.w * dy;
    const float s = dx * m0 + dy * m1;
    e = -0.5f * s;
    return __builtin_amdgcn_exp2f(s * (-0.5f * GS_LOG2E)) * Q.y;   // (= (-0.5 s) * log2 e to the last bit, see the pair form)
}
// the bracket around the 1/255 skip threshold (gs_common.h): below EPS_LO / from EPS_HI on, the kernels' alpha decides as the
// reference's does
constexpr float EPS_LO = EPS_ALPHA * (1.0f - GS_ALPHA_BAND), EPS_HI = EPS_ALPHA / (1.0f - GS_ALPHA_BAND);
// rows of a record as the forward / backward group loops want them (see above)
__device__ __forceinline__ void gs_stage_forward(const float4 r0, const float4 r1, const float4 r2, const float4 r3, float4 &P,
                                                 float4 &Q, float4 &colour, float2 &rescale_opacity) {
    P = make_float4(r0.x, r0.y, r1.x, r1.z);
    Q = make_float4(r1.y, r3.z, r3.x, r0.z);
    colour = make_float4(r2.x, r2.y, r2.z, r3.y);
    rescale_opacity = make_float2(r3.w, r2.w);
}
__device__ __forceinline__ void gs_stage_backward(const float4 r0, const float4 r1, const float4 r2, const float4 r3, float4 &P,
                                                  float4 &Q, float4 &colour) {
    P = make_float4(r0.x, r0.y, r1.x, r1.z);
    Q = make_float4(r1.y, r3.x, r1.w, -2.0f * r3.z);   // (B, amp, radius, s_hi = -2 e_lo: exact)
    colour = r2;
}
// The pixel's bracket around T' = 1e-4 from thr (its upper edge as the group loop keeps it: per-Gaussian weights + 4 u per
// list position walked) and the pixel's own number of blended Gaussians (what the rounding term really is; -1: unknown)
__device__ __forceinline__ void gs_stop_bracket(float thr, int walked, int blended, float &lo, float &hi) {
    hi = blended >= 0 ? thr - STOP_T * 1.1f * GS_STOP_ROUNDING_BAND * (float)(walked - blended) : thr;
    lo = fmaxf(STOP_T - (hi - STOP_T) * (1.0f / 1.1f), 0.f);
}
// Does the REFERENCE's forward pass stop pixel (pxr, pyr) at or before list position j_cur?  The whole wave replays the
// pixel's history from the start of the list in the reference's arithmetic (RAS:440-470: the reference's exponent, the
// correctly rounded exponential, * rescale * opacity, T *= 1 - min(alpha, 0.99)): 64 list entries per step, one per lane,
// then the transmittance recurrence over the hits in list order.  BOXED: the list covers a bin of several tiles -- entries
// whose tile box (RAS:81-103) does not hold this tile are skipped, as the staging skips them (entries the exact cull removed
// there fail the 1/255 test here).  Called about a thousand times per full-size frame: when a T' lands inside the pixel's
// bracket around 1e-4 (and the brackets hold: the stop, if any, is at j_cur).  870 replays per headline frame cost the
// forward kernel 35 us before the list entries were prefetched and the tile's own written-out list was used where it exists.
template <bool BOXED, bool OWN_LIST>
__device__ __forceinline__ bool gs_reference_stops_at(const int32_t *list, const float4 *__restrict__ attrs, int first,
                                                      int j_cur, float pxr, float pyr, int tile_u, int tile_v, int tw, int th) {
    // OWN_LIST: `list` is the tile's own kept entries as this workgroup has written them out so far (walked_list: read with
    // workgroup-scope loads, behind the barrier that followed the stores)
    auto entry_at = [&](int j) {
        return OWN_LIST ? __hip_atomic_load(list + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : list[j];
    };
    const int lane = gs_lane();
    float T = 1.0f;
    // (the next step's list entries are fetched while this step's records are on their way: the replay is a chain of
    // dependent gathers on the critical path of its tile)
    int o_next = first + lane <= j_cur ? entry_at(first + lane) : 0;
    for (int base = first; base <= j_cur; base += GS_WAVE) {
        const int j = base + lane, o = o_next;
        bool valid = j <= j_cur;
        if (j + GS_WAVE <= j_cur) o_next = entry_at(j + GS_WAVE);
        float a = 0.f;
        if (valid) {
            const float4 *g = attrs + 4 * (size_t)o;
            const float4 r0 = g[0], r1 = g[1];
            if (BOXED) valid = gs_entry_in_tile(r0, r1, tile_u, tile_v, tw, th, GS_FILTER_BOX);
            if (valid)
                a = gs_alpha_reference(gs_exponent_forward(pxr - r0.x, pyr - r0.y, r1.x, r1.y, r1.z),
                                       reinterpret_cast<const float *>(g)[15], reinterpret_cast<const float *>(g)[11]);
        }
        unsigned long long m = gs_ballot(valid && a >= EPS_ALPHA);
        while (m != 0ull) {
            const int l = __builtin_ctzll(m);
            m &= m - 1ull;
            const float Tn = T * (1.0f - fminf(gs_readlane_f(a, l), CLAMP_ALPHA));
            if (Tn < STOP_T) return true;   // RAS:458-460: the reference saturates the pixel here
            T = Tn;
        }
    }
    return false;
}

constexpr int BLEND_THREADS = 128;
constexpr int BATCH = 128;               // staged (kept) entries per blend round
constexpr int FILL_PER_THREAD = 1;       // list entries examined per thread and fill step (2: +10 VGPRs, one wave/SIMD less)
constexpr int FILL = BLEND_THREADS * FILL_PER_THREAD;
constexpr unsigned GS_HASH_MUL = 2654435761u;

// One fill step of the list staging, shared by both kernels.  The workgroup examines FILL consecutive list positions
// (ascending from `pos` when DIR = +1, descending from `pos` when DIR = -1, bounded by `limit`), keeps the entries that
// belong to the tile and appends them IN LIST ORDER behind the `nbuf` entries already staged; at most BATCH - nbuf are
// accepted and the walk resumes at the first entry that did not fit.  Returns through `pos` / `nbuf` (uniform).
// keep(row0, row1) is the membership test; store(slot, j, o, row0..row3) writes a kept entry to the kernel's LDS arrays.
// Two barriers per step.  s_cnt: int[4], s_next: int[1] in LDS.
template <int DIR, typename Keep, typename Store>
__device__ __forceinline__ void gs_fill_step(const int32_t *__restrict__ payload, const float4 *__restrict__ attrs,
                                             int &pos, int limit, int &nbuf, int *s_cnt, int *s_next, Keep keep,
                                             Store store) {
    const int tid = threadIdx.x, w = tid >> 6;
    int j[FILL_PER_THREAD], o[FILL_PER_THREAD];
    float4 r[FILL_PER_THREAD][4];
    bool kept[FILL_PER_THREAD];
    unsigned long long bal[FILL_PER_THREAD];
#pragma unroll
    for (int h = 0; h < FILL_PER_THREAD; ++h) {
        j[h] = pos + DIR * (h * BLEND_THREADS + tid);
        const bool valid = DIR > 0 ? j[h] < limit : j[h] >= limit;
        o[h] = valid ? payload[j[h]] : 0;
        kept[h] = valid;
    }
#pragma unroll
    for (int h = 0; h < FILL_PER_THREAD; ++h) {
        const float4 *g = attrs + 4 * (size_t)o[h];
        if (kept[h]) { r[h][0] = g[0]; r[h][1] = g[1]; r[h][2] = g[2]; r[h][3] = g[3]; }
    }
#pragma unroll
    for (int h = 0; h < FILL_PER_THREAD; ++h) {
        kept[h] = kept[h] && keep(r[h][0], r[h][1]);
        bal[h] = gs_ballot(kept[h]);
        if ((tid & 63) == 0) s_cnt[2 * h + w] = __popcll(bal[h]);
    }
    __syncthreads();
    int before = nbuf;   // staged entries ahead of this half-step's first wave
#pragma unroll
    for (int h = 0; h < FILL_PER_THREAD; ++h) {
        const int c0 = s_cnt[2 * h], c1 = s_cnt[2 * h + 1];
        const int slot = before + (w ? c0 : 0) + gs_mbcnt(bal[h]);
        if (kept[h]) {
            if (slot < BATCH) store(slot, j[h], o[h], r[h][0], r[h][1], r[h][2], r[h][3]);
            else if (slot == BATCH) *s_next = j[h];   // first entry that does not fit: the walk resumes here
        }
        before += c0 + c1;
    }
    __syncthreads();
    // s_next is written (by exactly one thread) and read only when the step overflowed the batch -- and then the fill
    // loop ends, so the next write is at least one more barrier away from this read
    pos = before > BATCH ? *s_next : pos + DIR * FILL;
    nbuf = min(before, BATCH);
}

// Per-tile lists (bin_shift 0: the key generator already emitted exactly the tile's entries): the next <= BATCH
// positions are staged as they are -- no test, no compaction, one barrier (the caller's).
template <int DIR, typename Store>
__device__ __forceinline__ void gs_direct_step(const int32_t *__restrict__ payload, const float4 *__restrict__ attrs,
                                               int &pos, int limit, int &nbuf, Store store) {
    const int tid = threadIdx.x;
    const int j = pos + DIR * tid;
    const bool valid = DIR > 0 ? j < limit : j >= limit;
    const int o = valid ? payload[j] : 0;
    float4 r0, r1, r2, r3;   // (scalars, not an array: an array handed to `store` by pointer ends up in scratch memory)
    if (valid) {
        const float4 *g = attrs + 4 * (size_t)o;
        r0 = g[0]; r1 = g[1]; r2 = g[2]; r3 = g[3];
    }
    if (valid) store(tid, j, o, r0, r1, r2, r3);
    nbuf = min(BATCH, DIR > 0 ? limit - pos : pos - limit + 1);
    pos += DIR * BATCH;
}

// ------------------------------------------------------------------------------- dispatch order
// The hardware hands workgroups to the CUs in blockIdx order as slots free up.  Tiles differ in work by an order of
// magnitude (list positions walked: mean 158, max 383 per-tile / 292 and 749 binned at the headline scene), so with
// tiles in image order the launch ends with a long tail of half-empty CUs: VALU busy 80 %, 3.4 of 5 waves resident on
// average.  Longest-first (LPT) order fixes most of it (backward 0.46 -> 0.40 ms measured with an order computed by
// torch).  This kernel is that order on the device: a counting sort of the owned tiles by 1024 classes of their work
// estimate, descending (LDS atomics; the order inside a class is arbitrary and irrelevant -- a tile's results do not
// depend on when it runs).  work[e] given: the estimate of tile e (the forward kernel records the walk
// length the backward will see); work == nullptr: the length of the tile's (bin's) sorted list.
// Eight workgroups, one per residue of the tile number modulo 8: workgroup r sorts the tiles e = r (mod 8) among themselves
// and stores its p-th heaviest at order[r + 8 p] -- block b of the blend launch (which the hardware places on XCD b mod 8)
// simply takes order[b], i.e. the (b / 8)-th heaviest tile of list b mod 8: eight descending lists interleaved, no
// communication between the workgroups, and each XCD works through its own list heaviest first.  (One workgroup sorting
// all 8,040 tiles took 10-13 us per call, most of it fixed latency of a lone workgroup.)
constexpr int ORDER_WGS = 8, ORDER_THREADS = 256, ORDER_CLASSES = 1024, ORDER_ITEMS = 4;
constexpr int ORDER_WAVES = ORDER_THREADS / GS_WAVE, ORDER_CLASSES_PER_THREAD = ORDER_CLASSES / ORDER_THREADS;
// Workgroups past the eight sorting ones (the backward pass' launch, gs_blend_backward_split) zero `zero_bytes` bytes at
// `zero`, 16 KB each: the slot flags of the backward pass are cleared beside the sort instead of by a fill launch of their
// own in front of it (5.4 us per frame at the headline size; the eight sorting workgroups leave 248 CUs idle).
constexpr int ORDER_ZERO_PER_THREAD = 4;                                              // 16-byte stores per thread
constexpr int ORDER_ZERO_BYTES_PER_WG = ORDER_THREADS * ORDER_ZERO_PER_THREAD * 16;   // 16 KB
__global__ __launch_bounds__(ORDER_THREADS) void tile_order_kernel(
    const int32_t *__restrict__ work, const int32_t *__restrict__ bin_start, const int32_t *__restrict__ bin_end, int n,
    int tw, int row_begin, int row_step, int bin_shift, int32_t *__restrict__ order, uint4 *__restrict__ zero,
    long long zero_bytes) {
    if (blockIdx.x >= ORDER_WGS) {
        const long long first = ((long long)(blockIdx.x - ORDER_WGS) * ORDER_ZERO_BYTES_PER_WG) / 16 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < ORDER_ZERO_PER_THREAD; ++k) {
            const long long q = first + (long long)k * ORDER_THREADS;
            if (q * 16 < zero_bytes) zero[q] = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    // one histogram PER WAVE: the tiles of a frame crowd into few classes, and a shared histogram turns every LDS atomic
    // into a many-way same-address conflict
    __shared__ int s_hist[ORDER_WAVES][ORDER_CLASSES];
    __shared__ int s_red[ORDER_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = blockIdx.x;
    const int n_local = n > r ? (n - r + ORDER_WGS - 1) / ORDER_WGS : 0;   // tiles r, r + 8, ... below n
    const int bins_u = (tw + (1 << bin_shift) - 1) >> bin_shift;
    auto work_of = [&](int j) {   // j-th tile of this workgroup's list
        const int e = r + ORDER_WGS * j;
        if (work != nullptr) return max(work[e], 0);
        const int tu = e % tw, tv = row_begin + (e / tw) * row_step;
        const int bin = (tu >> bin_shift) + (tv >> bin_shift) * bins_u;
        return max(bin_end[bin] - bin_start[bin], 0);
    };
    // up to ORDER_ITEMS x 256 tiles per list (8 x 1024 = 8192 tiles: a 1920 x 1088 frame) live in registers: one read of the
    // estimates; larger frames re-read them in the later passes
    int w[ORDER_ITEMS];
    int mx = 1;
#pragma unroll
    for (int k = 0; k < ORDER_ITEMS; ++k) {
        const int j = k * ORDER_THREADS + tid;
        w[k] = j < n_local ? work_of(j) : -1;
        mx = max(mx, w[k]);
    }
    for (int j = ORDER_ITEMS * ORDER_THREADS + tid; j < n_local; j += ORDER_THREADS) mx = max(mx, work_of(j));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, GS_WAVE));
    if (lane == 0) s_red[wv] = mx;
#pragma unroll
    for (int k = 0; k < ORDER_WAVES; ++k)
#pragma unroll
        for (int c = 0; c < ORDER_CLASSES_PER_THREAD; ++c) s_hist[k][c * ORDER_THREADS + tid] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ORDER_WAVES; ++k) mx = max(mx, s_red[k]);
    // class 0 = heaviest: the counting sort below is then ascending in the class
    const float scale = (float)(ORDER_CLASSES - 1) / (float)mx;
    auto class_of = [&](int x) { return ORDER_CLASSES - 1 - min((int)((float)x * scale), ORDER_CLASSES - 1); };
#pragma unroll
    for (int k = 0; k < ORDER_ITEMS; ++k)
        if (w[k] >= 0) atomicAdd(&s_hist[wv][class_of(w[k])], 1);
    for (int j = ORDER_ITEMS * ORDER_THREADS + tid; j < n_local; j += ORDER_THREADS) atomicAdd(&s_hist[wv][class_of(work_of(j))], 1);
    __syncthreads();
    // thread t owns the consecutive classes 4 t .. 4 t + 3: exclusive prefix of every class over the waves (in place), class
    // totals scanned over the classes
    int first[ORDER_CLASSES_PER_THREAD], mine = 0;
#pragma unroll
    for (int c = 0; c < ORDER_CLASSES_PER_THREAD; ++c) {
        const int cls = ORDER_CLASSES_PER_THREAD * tid + c;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < ORDER_WAVES; ++k) {
            const int v = s_hist[k][cls];
            s_hist[k][cls] = cnt;
            cnt += v;
        }
        first[c] = mine;      // (exclusive within the thread)
        mine += cnt;
    }
    const int incl = gs_wave_incl_scan(mine);
    __syncthreads();                       // (s_red is reused)
    if (lane == GS_WAVE - 1) s_red[wv] = incl;
    __syncthreads();
    int before = incl - mine;
#pragma unroll
    for (int k = 0; k < ORDER_WAVES; ++k) before += k < wv ? s_red[k] : 0;
#pragma unroll
    for (int c = 0; c < ORDER_CLASSES_PER_THREAD; ++c)
#pragma unroll
        for (int k = 0; k < ORDER_WAVES; ++k)
            s_hist[k][ORDER_CLASSES_PER_THREAD * tid + c] += before + first[c];   // -> running cursor of (wave, class)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ORDER_ITEMS; ++k)
        if (w[k] >= 0)
            order[r + ORDER_WGS * atomicAdd(&s_hist[wv][class_of(w[k])], 1)] = r + ORDER_WGS * (k * ORDER_THREADS + tid);
    for (int j = ORDER_ITEMS * ORDER_THREADS + tid; j < n_local; j += ORDER_THREADS)
        order[r + ORDER_WGS * atomicAdd(&s_hist[wv][class_of(work_of(j))], 1)] = r + ORDER_WGS * j;
}

// ------------------------------------------------------------------------------- forward
// STAGED: lists cover a bin of several tiles and are filtered while they are staged (gs_fill_step); otherwise they are
//         the tile's own list (gs_direct_step)
// AUX:   depth and per-pixel count are produced (off with rgb_only, RAS:464-469,478-484)
// STATE: acc_alpha / last_effective are produced (what the backward pass needs; off for inference)
// DEBUG: per-pixel {number of blended Gaussians, wrap-around sum of (payload+1)*GS_HASH_MUL} -> debug_hits (tests)
template <bool STAGED, bool AUX, bool STATE, bool DEBUG>
__global__ __launch_bounds__(BLEND_THREADS, GS_FWD_MIN_WAVES) void blend_forward_kernel(
    const int32_t *__restrict__ bin_start, const int32_t *__restrict__ bin_end,
    const int32_t *__restrict__ payload, const float4 *__restrict__ attrs, int width, int height, int row_begin,
    int row_step, int bin_shift, int filter, float *__restrict__ image, float *__restrict__ depth,
    float *__restrict__ acc_alpha, int32_t *__restrict__ last_effective, int32_t *__restrict__ valid_count,
    uint32_t *__restrict__ debug_hits, const int32_t *__restrict__ tile_order, int32_t *__restrict__ tile_work,
    int32_t *__restrict__ walked_list, int32_t *__restrict__ walked_start) {
    __shared__ float4 s_p[BATCH], s_q[BATCH], s_c[BATCH];  // P, Q and the colour row of the kept records (gs_stage_forward)
    __shared__ float2 s_ro[BATCH];                         // their (rescale, opacity): read by the careful path only
    __shared__ int s_j[BATCH];                             // their list positions (last_effective is one of them + 1)
    __shared__ int s_o[DEBUG ? BATCH : 1];
    __shared__ int s_cnt[2 * FILL_PER_THREAD], s_next[1];
    const int tw = width / GS_TILE_WIDTH, th = height / GS_TILE_HEIGHT;
    const TileCoord tc = owned_tile(tw, row_begin, row_step, tile_order);
    const int tid = threadIdx.x;
    const int pu = tc.tile_u * GS_TILE_WIDTH + 2 * (tid & 7);   // left pixel of the pair
    const int pv = tc.tile_v * GS_TILE_HEIGHT + (tid >> 3);
    const int bins_u = (tw + (1 << bin_shift) - 1) >> bin_shift;
    const int bin = (tc.tile_u >> bin_shift) + (tc.tile_v >> bin_shift) * bins_u;
    const int start = bin_start[bin], end = bin_end[bin];
    const v2f px = {(float)pu + 0.5f, (float)pu + 1.5f};
    const float py = (float)pv + 0.5f;
    // The tile's OWN list, as far as this pass walks it.  A tile of a binned layout filters its bin's list while staging it;
    // the entries it keeps are written out here -- tile t of bin b owns the segment [(start_b << 2 s) + t len_b, + len_b) of
    // walked_list -- so that the backward pass walks a plain per-tile list (no filter, no compaction, no second examination
    // of the bin's entries: it never goes beyond what the forward pass walked).  last_effective then refers to positions
    // in walked_list and walked_start[tile] holds the segment's first position.
    const bool emit = STAGED && STATE && walked_list != nullptr;
    const int bin_mask = (1 << bin_shift) - 1;
    const int wbase = emit ? (start << (2 * bin_shift)) +
                                 ((tc.tile_u & bin_mask) + ((tc.tile_v & bin_mask) << bin_shift)) * (end - start)
                           : start;
    int kept_base = 0;   // entries this tile kept in earlier rounds

    // alive0 / alive1: the lanes whose left / right pixel has not saturated (wave-uniform masks in scalar registers: ANDed with
    // the ballot of the hit test on the scalar unit -- no per-pixel "alive" factor in the vector arithmetic)
    v2f T = splat(1.0f);
    unsigned long long alive0 = ~0ull, alive1 = ~0ull;
    v2f Cr = splat(0.f), Cg = splat(0.f), Cb = splat(0.f), D = splat(0.f), Wd = splat(0.f);
    int last0 = wbase, last1 = wbase, cnt0 = 0, cnt1 = 0;
    unsigned dh0 = 0u, dh1 = 0u, dc0 = 0u, dc1 = 0u;

    auto keep = [&](const float4 r0, const float4 r1) {
        return gs_entry_in_tile(r0, r1, tc.tile_u, tc.tile_v, tw, th, filter);
    };
    auto store = [&](int slot, int j, int o, const float4 r0, const float4 r1, const float4 r2, const float4 r3) {
        float4 P, Q, colour;
        float2 ro;
        gs_stage_forward(r0, r1, r2, r3, P, Q, colour, ro);
        s_p[slot] = P; s_q[slot] = Q; s_c[slot] = colour; s_ro[slot] = ro;
        if (STAGED) s_j[slot] = j;
        if (DEBUG) s_o[slot] = o;
        if (emit) walked_list[wbase + kept_base + slot] = o;
    };
    // upper edge of each pixel's bracket around T' = 1e-4 (gs_common.h, 4.): + 4 u per list position walked (batch by batch),
    // + every blended Gaussian's weight times its alpha
    v2f thr = splat(STOP_T);
    int park0 = GROUP_FWD, park1 = GROUP_FWD;   // entry of the current group at which the pixel is parked (GROUP_FWD: it is not)

    int pos = start;
    while (pos < end) {
        // barrier (protects the staged batch) + whole-tile early exit vote
        if (__syncthreads_and((alive0 | alive1) == 0ull ? 1 : 0)) break;
        int nbuf = 0;
        const int batch_first = pos;   // direct path: staged entry k sits at list position batch_first + k
        if (STAGED)
            while (nbuf < BATCH && pos < end)
                gs_fill_step<+1>(payload, attrs, pos, end, nbuf, s_cnt, s_next, keep, store);
        else
            gs_direct_step<+1>(payload, attrs, pos, end, nbuf, store);
        {   // pad to a multiple of GROUP with inert records (e_lo = +inf: no exponent passes the hit test)
            const int padded = (nbuf + GROUP - 1) & ~(GROUP - 1);
            if (tid < padded - nbuf) {
                s_p[nbuf + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
                s_q[nbuf + tid] = make_float4(0.f, __builtin_inff(), 0.f, 0.f);
            }
        }
        __syncthreads();
        thr = thr + splat(STOP_T * 1.1f * GS_STOP_ROUNDING_BAND * (float)nbuf + GS_STOP_THR_SLACK);
        const int walked = kept_base + nbuf;   // list positions of the tile walked so far (this batch included)
        // The blend update of one entry, shared by the group loop and its careful twin below (al = 0 for a skipped pixel makes
        // the update an exact no-op: T*(1-0) = T, C += c*0)
        auto blend = [&](int e, const float4 c, float z, v2f al, v2f Tn, bool ok0, bool ok1) {
            const v2f wgt = al * T;
            Cr = fma2(splat(c.x), wgt, Cr);
            Cg = fma2(splat(c.y), wgt, Cg);
            Cb = fma2(splat(c.z), wgt, Cb);
            if (AUX) {
                D = fma2(splat(z), wgt, D);
                Wd = Wd + wgt;
                cnt0 += ok0 ? 1 : 0;
                cnt1 += ok1 ? 1 : 0;
            }
            T = Tn;
            if (STATE) {
                const int idx = (emit ? wbase + kept_base + e : STAGED ? s_j[e] : batch_first + e) + 1;
                last0 = ok0 ? idx : last0;
                last1 = ok1 ? idx : last1;
            }
            if (DEBUG) {
                const unsigned hv = (unsigned)(s_o[e] + 1) * GS_HASH_MUL;
                dc0 += ok0 ? 1u : 0u; dh0 += ok0 ? hv : 0u;
                dc1 += ok1 ? 1u : 0u; dh1 += ok1 ? hv : 0u;
            }
        };
        // PARKED PIXELS.  When a comparison of the group loop falls inside a bracket -- an exponent at or above e_lo whose alpha
        // does not reach EPS_HI, a T' inside the pixel's stop bracket: about a thousand (wave, entry) visits per full-size frame
        // -- the decision is not the loop's to take: the pixel is PARKED at that entry (its state untouched, its alive bit cleared
        // so that the rest of the group passes it by) and the loop goes on for the other pixels without leaving its straight-line
        // body.  After the group the wave takes the parked pixels through the entries they missed with every decision taken as
        // the reference takes it (careful_entry: the reference's expression with the same exponent, correctly rounded exp, *
        // rescale * opacity; a replay of the pixel's history in the reference's arithmetic for a T' in the bracket).  Every
        // update of a pixel that is not taking part (al = 0) is an exact no-op, so the catch-up runs the ordinary blend update
        // on the whole wave.  Kept out of the group loop's body so that its temporaries (double-precision exp, the replay's
        // records) do not add to the registers of the hot path, nor its exits to the loop's control flow (an early-exit form
        // cost the forward kernel 20 us).
        // m0 / m1: the pixels taking part; dead0 / dead1: those the reference stops during the catch-up.
        auto careful_entry = [&](int e, bool m0, bool m1, bool &dead0, bool &dead1) {
            v2f ex;
            const float4 P = s_p[e], Q = s_q[e];
            const v2f am = {m0 && !dead0 ? 1.f : 0.f, m1 && !dead1 ? 1.f : 0.f};
            const v2f a = gs_pair_alpha_forward(P, Q, px, py, ex) * am;
            bool ok0 = a.x >= EPS_ALPHA, ok1 = a.y >= EPS_ALPHA;   // RAS:451
            {
                const bool in0 = a.x >= EPS_LO && a.x < EPS_HI, in1 = a.y >= EPS_LO && a.y < EPS_HI;
                if (gs_ballot(in0 || in1) != 0ull) {
                    GS_STAT(GS_STAT_FWD_EXACT_ALPHA, 1);
                    const float2 ro = s_ro[e];
#pragma clang loop unroll(disable)
                    for (int c = 0; c < 2; ++c) {
                        const float exact = gs_alpha_reference(c ? ex.y : ex.x, ro.x, ro.y);
                        if (c ? in1 : in0) { if (c) ok1 = exact * am.y >= EPS_ALPHA; else ok0 = exact * am.x >= EPS_ALPHA; }
                    }
                }
            }
            if (gs_ballot(ok0 || ok1) == 0ull) return;
            v2f al = {ok0 ? __builtin_amdgcn_fmed3f(a.x, 0.f, CLAMP_ALPHA) : 0.f,
                      ok1 ? __builtin_amdgcn_fmed3f(a.y, 0.f, CLAMP_ALPHA) : 0.f};
            const float4 c = s_c[e];
            thr = fma2(splat(c.w), al, thr);   // this Gaussian's share of the pixel's stop bracket (its own factor included)
            v2f Tn = T * (splat(1.f) - al);
            float lo0, hi0, lo1, hi1;
            gs_stop_bracket(thr.x, walked, AUX ? cnt0 + 1 : -1, lo0, hi0);
            gs_stop_bracket(thr.y, walked, AUX ? cnt1 + 1 : -1, lo1, hi1);
            // RAS:458-460: below the pixel's bracket it stops on both sides, inside it the reference's arithmetic decides
            bool sat0 = ok0 && Tn.x < lo0, sat1 = ok1 && Tn.y < lo1;
            {
                const unsigned long long mf0 = gs_ballot(ok0 && !sat0 && Tn.x < hi0), mf1 = gs_ballot(ok1 && !sat1 && Tn.y < hi1);
                if ((mf0 | mf1) != 0ull) {
                    const int j_cur = STAGED ? s_j[e] : batch_first + e;
#pragma clang loop unroll(disable)
                    for (int c = 0; c < 2; ++c)
                        for (unsigned long long m = c ? mf1 : mf0; m != 0ull; m &= m - 1ull) {
                            const int l = __builtin_ctzll(m);
                            GS_STAT(GS_STAT_FWD_REPLAYS, 1);
                            GS_STAT(GS_STAT_FWD_REPLAY_ENTRIES, j_cur - start + 1);
                            const float pxr = gs_readlane_f(c ? px.y : px.x, l), pyr = gs_readlane_f(py, l);
                            // (the tile's own written-out list where there is one: half the entries of the bin's, no box test)
                            const bool stops = emit ? gs_reference_stops_at<false, true>(walked_list, attrs, wbase, wbase + kept_base + e,
                                                                                        pxr, pyr, tc.tile_u, tc.tile_v, tw, th)
                                                    : gs_reference_stops_at<STAGED, false>(payload, attrs, start, j_cur, pxr, pyr,
                                                                                           tc.tile_u, tc.tile_v, tw, th);
                            if ((tid & (GS_WAVE - 1)) == l) { if (c) sat1 = stops; else sat0 = stops; }
                        }
                }
            }
            if (sat0) { al.x = 0.f; dead0 = true; ok0 = false; }
            if (sat1) { al.y = 0.f; dead1 = true; ok1 = false; }
            Tn = T * (splat(1.f) - al);
            blend(e, c, Q.w, al, Tn, ok0, ok1);
        };
        // Entries are evaluated in groups of GROUP: the LDS reads and the quadratic forms of a group are independent
        // and overlap (the per-pixel blend recurrence is the only serial part), which hides their latency.
        for (int k = 0; k < nbuf; k += GROUP_FWD) {
            if ((alive0 | alive1) == 0ull) break;  // every pixel of this wave is saturated
            v2f ex[GROUP_FWD];
            float elo[GROUP_FWD];
#if GS_FWD_WHOLE_Q
            float2 azs[GROUP_FWD];
#endif
#pragma unroll
            for (int i = 0; i < GROUP_FWD; ++i) {
#if GS_FWD_WHOLE_Q
                const float4 q = s_q[k + i];
                azs[i] = make_float2(q.z, q.w);
#else
                const float2 q = *reinterpret_cast<const float2 *>(&s_q[k + i]);   // (B, e_lo)
#endif
                ex[i] = gs_pair_exponent_forward(s_p[k + i], q.x, px, py);
                elo[i] = q.y;
            }
#if GS_FWD_UPFRONT   // the group's exponents exist before the first hit test (left alone the compiler sinks each evaluation to its test)
#pragma unroll
            for (int i = 0; i < GROUP_FWD; ++i) asm volatile("" : "+v"(ex[i].x), "+v"(ex[i].y));
#endif
#if GS_ABLATE_FWD == 1   // tuning only: evaluation without the blend (what the evaluation of every visited entry costs)
#pragma unroll
            for (int i = 0; i < GROUP_FWD; ++i) { Cr = Cr + ex[i]; last0 += (int)elo[i]; }
            continue;
#endif
            bool parked = false;   // wave-uniform: a pixel of this wave was parked in this group
            GS_STAT(GS_STAT_FWD_ENTRIES, GROUP_FWD);
#pragma unroll
            for (int i = 0; i < GROUP_FWD; ++i) {
                // RAS:451 in the exponent's domain: below e_lo the reference skips the pair (live pixels only)
                unsigned long long mok0 = gs_ballot(ex[i].x >= elo[i]) & alive0, mok1 = gs_ballot(ex[i].y >= elo[i]) & alive1;
                if ((mok0 | mok1) == 0ull) continue;                  // wave-uniform skip: no exponential is evaluated
                GS_STAT(GS_STAT_FWD_HIT_ENTRIES, 1);
                GS_STAT(GS_STAT_FWD_HIT_PIXELS, __popcll(mok0) + __popcll(mok1));
                GS_STAT(GS_STAT_FWD_HIT_LANES, __popcll(mok0 | mok1));
                GS_STAT(GS_STAT_FWD_HIT_BLOCKS, (((mok0 | mok1) & 0x0f0f0f0f0f0f0f0full) != 0ull) + (((mok0 | mok1) & 0xf0f0f0f0f0f0f0f0ull) != 0ull));
                const float4 c = s_c[k + i];
#if GS_FWD_WHOLE_Q
                const float2 az = azs[i];
#else
                const float2 az = reinterpret_cast<const float2 *>(&s_q[k + i])[1];   // (amp, depth)
#endif
                const v2f a = gs_weight_from_exponent(ex[i], az.x);
                {
                    const unsigned long long mhi0 = gs_ballot(a.x >= EPS_HI) & mok0, mhi1 = gs_ballot(a.y >= EPS_HI) & mok1;
                    if (((mok0 ^ mhi0) | (mok1 ^ mhi1)) != 0ull) {    // rare: not a hit for certain -> park the pixel here
                        const unsigned long long p0 = mok0 & ~mhi0, p1 = mok1 & ~mhi1;
                        if (__builtin_amdgcn_inverse_ballot_w64(p0)) park0 = i;
                        if (__builtin_amdgcn_inverse_ballot_w64(p1)) park1 = i;
                        alive0 &= ~p0; alive1 &= ~p1;
                        parked = true;
                        mok0 = mhi0; mok1 = mhi1;
                        if ((mok0 | mok1) == 0ull) continue;
                    }
                }
                bool ok0 = __builtin_amdgcn_inverse_ballot_w64(mok0), ok1 = __builtin_amdgcn_inverse_ballot_w64(mok1);
                // alpha = 0 for a skipped pixel makes the update an exact no-op
                v2f al = {ok0 ? __builtin_amdgcn_fmed3f(a.x, 0.f, CLAMP_ALPHA) : 0.f,
                          ok1 ? __builtin_amdgcn_fmed3f(a.y, 0.f, CLAMP_ALPHA) : 0.f};
                thr = fma2(splat(c.w), al, thr);   // this Gaussian's share of the pixel's stop bracket (its own factor included)
                v2f Tn = T * (splat(1.f) - al);
                const bool low0 = Tn.x < thr.x, low1 = Tn.y < thr.y;
                // (masks combined on the scalar unit: a ballot of the AND would be materialised as select + compare)
                if (((mok0 & gs_ballot(low0)) | (mok1 & gs_ballot(low1))) != 0ull) {
                    // rare: RAS:458-460 -- the first Gaussian that would push T below 1e-4 saturates the
                    // pixel and is NOT blended (below the pixel's bracket on both sides; inside it: parked)
                    float lo0, hi0, lo1, hi1;
                    gs_stop_bracket(thr.x, walked, AUX ? cnt0 + 1 : -1, lo0, hi0);
                    gs_stop_bracket(thr.y, walked, AUX ? cnt1 + 1 : -1, lo1, hi1);
                    const bool sat0 = ok0 && Tn.x < lo0, sat1 = ok1 && Tn.y < lo1;
                    const bool fr0 = ok0 && !sat0 && Tn.x < hi0, fr1 = ok1 && !sat1 && Tn.y < hi1;
                    const unsigned long long mfr0 = gs_ballot(fr0), mfr1 = gs_ballot(fr1);
                    if ((mfr0 | mfr1) != 0ull) {
                        if (fr0) { park0 = i; al.x = 0.f; ok0 = false; }
                        if (fr1) { park1 = i; al.y = 0.f; ok1 = false; }
                        alive0 &= ~mfr0; alive1 &= ~mfr1;
                        parked = true;
                    }
                    if (sat0) { al.x = 0.f; ok0 = false; }
                    if (sat1) { al.y = 0.f; ok1 = false; }
                    alive0 &= ~gs_ballot(sat0); alive1 &= ~gs_ballot(sat1);
                    Tn = T * (splat(1.f) - al);
                }
                blend(k + i, c, az.y, al, Tn, ok0, ok1);
            }
            if (parked) {   // the parked pixels catch up on the entries of the group they missed
                bool dead0 = false, dead1 = false;
#pragma clang loop unroll(disable)
                for (int i = 0; i < GROUP_FWD; ++i) {
                    const bool m0 = park0 <= i, m1 = park1 <= i;
                    if (gs_ballot(m0 || m1) == 0ull) continue;
                    GS_STAT(GS_STAT_FWD_CAREFUL_ENTRIES, 1);
                    careful_entry(k + i, m0, m1, dead0, dead1);
                }
                alive0 |= gs_ballot(park0 < GROUP_FWD && !dead0);
                alive1 |= gs_ballot(park1 < GROUP_FWD && !dead1);
                park0 = GROUP_FWD; park1 = GROUP_FWD;
            }
        }
        kept_base += nbuf;
    }
    const size_t p = (size_t)pv * width + pu;
    float *img = image + 3 * p;
    img[0] = Cr.x; img[1] = Cg.x; img[2] = Cb.x; img[3] = Cr.y; img[4] = Cg.y; img[5] = Cb.y;
    if (AUX) {
        depth[p] = D.x / fmaxf(Wd.x, 1e-6f);  // RAS:479-480
        depth[p + 1] = D.y / fmaxf(Wd.y, 1e-6f);
        valid_count[p] = cnt0;
        valid_count[p + 1] = cnt1;
    }
    if (STATE) {
        acc_alpha[p] = 1.f - T.x;
        acc_alpha[p + 1] = 1.f - T.y;
        last_effective[p] = last0;
        last_effective[p + 1] = last1;
        if (tile_work != nullptr) {   // list positions the backward pass will walk for this tile: its dispatch-order estimate
            int mx = max(last0, last1);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, GS_WAVE));
            __syncthreads();          // (all waves are past their last use of s_cnt)
            if ((tid & 63) == 0) s_cnt[tid >> 6] = mx;
            __syncthreads();
            if (tid == 0) tile_work[tc.index] = max(s_cnt[0], s_cnt[1]) - wbase;
        }
        if (emit && tid == 0) walked_start[tc.tile_id] = wbase;
    }
    if (DEBUG) {
        debug_hits[2 * p] = dc0; debug_hits[2 * p + 1] = dh0;
        debug_hits[2 * p + 2] = dc1; debug_hits[2 * p + 3] = dh1;
    }
}

// ------------------------------------------------------------------------------- backward
// Workgroup = one tile = 2 wave64s, every lane owns two horizontally adjacent pixels (as in the forward kernel).
// Doubling the pixels per lane halves the number of cross-lane reductions, LDS record reads and per-entry uniform work
// per pixel; the kernel is bound by VALU issue.  alpha comes from gs_pair_alpha_backward (the reference's backward
// expression, whose m = conic @ d also serves the gradients), w = dL/dalpha * alpha_unclamped (= dL/dg * g,
// UTL:343), and the twelve-value reduce-scatter carries the pixel count as a float (exact below 2^24).  Per round of up
// to 128 staged entries the two waves combine their partial sums in LDS (ds_add_f32), then thread k stores entry k's
// 48-B record into its (Gaussian, tile) slot (plain stores).
// REDUCE_ARM: how a hit entry's partial sums cross the lanes (GS_BWD_REDUCE_* above)
template <bool STAGED, bool DEBUG, int REDUCE_ARM>
__global__ __launch_bounds__(BLEND_THREADS, GS_BWD_MIN_WAVES) void blend_backward_kernel(
    const int32_t *__restrict__ bin_start, const int32_t *__restrict__ payload,
    const float4 *__restrict__ attrs, const float *__restrict__ grad_image, const float *__restrict__ acc_alpha,
    const int32_t *__restrict__ last_effective, int width, int height, int row_begin, int row_step, int bin_shift,
    int filter, const int32_t *__restrict__ slot_offsets, float4 *__restrict__ partials,
    uint8_t *__restrict__ slot_flags, float *__restrict__ magnitude_image, uint32_t *__restrict__ debug_hits,
    const int32_t *__restrict__ tile_order) {
    __shared__ float4 s_p[BATCH], s_q[BATCH], s_c[BATCH];  // P, Q and the colour row of the kept records (gs_stage_backward)
    __shared__ __attribute__((aligned(16))) int s_j[BATCH];
    __shared__ int s_o[BATCH];
    __shared__ float s_acc[BATCH][GS_ACC_STRIDE];  // [entry][value]; value 10 = pixel count (as a float)
    constexpr int REDUCE_ = REDUCE_ARM;
    // REDUCE 3: the cross-lane sums go THROUGH LDS (gs_lds_reduce11 below): per wave eleven rows of 64 partials, 68 floats apart
    __shared__ __attribute__((aligned(16))) float s_tr[REDUCE_ >= 3 ? BLEND_THREADS / GS_WAVE : 1][REDUCE_ == 6 ? 6 * GS_TR_STRIDE : REDUCE_ >= 3 ? GS_TR_ROWS * GS_TR_STRIDE : 4];
    __shared__ int s_max[BLEND_THREADS / GS_WAVE];
    __shared__ int s_cnt[2 * FILL_PER_THREAD], s_next[1];
    const int tw = width / GS_TILE_WIDTH, th = height / GS_TILE_HEIGHT;
    const TileCoord tc = owned_tile(tw, row_begin, row_step, tile_order);
    const int tid = threadIdx.x, lane = tid & 63;
    const int pu = tc.tile_u * GS_TILE_WIDTH + 2 * (tid & 7);  // left pixel of the pair
    const int pv = tc.tile_v * GS_TILE_HEIGHT + (tid >> 3);
    const size_t p = (size_t)pv * width + pu;
    const int bins_u = (tw + (1 << bin_shift) - 1) >> bin_shift;
    const int start = bin_start[(tc.tile_u >> bin_shift) + (tc.tile_v >> bin_shift) * bins_u];
    const v2f px = {(float)pu + 0.5f, (float)pu + 1.5f};
    const float py = (float)pv + 0.5f;

    const int last0 = last_effective[p], last1 = last_effective[p + 1];
    v2f T = {1.0f - acc_alpha[p], 1.0f - acc_alpha[p + 1]};
    // S = sum_{j behind i} (c_j . G) a_j T_j: the reference keeps the colour-space suffix sum w_i (RAS:652-656)
    // and dots it with dL/dimage; only that dot product is ever used, so the scalar is carried instead
    // (same quantity re-associated: 6 instead of 12 flops per hit).
    v2f S = splat(0.f);
    const float *gi = grad_image + 3 * p;
    const v2f Gr = {gi[0], gi[3]}, Gg = {gi[1], gi[4]}, Gb = {gi[2], gi[5]};
    v2f mag_u = splat(0.f), mag_v = splat(0.f);
    unsigned dh0 = 0u, dh1 = 0u, dc0 = 0u, dc1 = 0u;

    // no pixel of the tile touches a list position at or beyond the tile-wide max of `last`
    int mx = max(last0, last1);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, GS_WAVE));
    const int wave_end = mx;  // no pixel of THIS wave touches a position at or beyond wave_end
    if (lane == 0) s_max[tid >> 6] = mx;
    __syncthreads();
    const int end = max(s_max[0], s_max[1]);
    // LDS destination of this lane's reduce-scatter results (lane 15 of row r holds the totals of three values)
    const int row = lane >> 4;
    const int slot = ((row & 1) << 1) | (row >> 1);  // rows (0,1,2,3) -> values (0,2,1,3) of each result register
    const bool row_tail = (lane & 15) == 15;
    constexpr int REDUCE = REDUCE_ARM;
    // destination of this lane's pair-reduce results (gs_wave_reduce12_pair): lanes 7 and 15 of each row
    const bool pair_tail = (lane & 7) == 7, pair_b_sel = (row >> 1) != 0;
    const int pair_vi = 2 * ((lane >> 3) & 1) + (row & 1);
    // (REDUCE 3) lane 4 n + p reads floats [16 p, 16 p + 16) of row n; lanes 44.. read row 10 again and add nothing
    const int tr_read = min(lane >> 2, GS_TR_ROWS - 1) * GS_TR_STRIDE + 16 * (lane & 3);
    const bool tr_owner = (lane & 3) == 0 && lane < 4 * GS_TR_ROWS;
    // (REDUCE 6) lane 4 v + p reads floats [8 p, 8 p + 8) of value v's half row; lanes 44.. read value 10's again and add nothing
    const int tr_v = min(lane >> 2, GS_TR_ROWS - 1);
    const int tr_read_half = (tr_v >> 1) * GS_TR_STRIDE + (tr_v & 1) * 32 + 8 * (lane & 3);
    // (REDUCE 4) sixteen fetched values -> this lane's quarter of row n -> the wave's total of value n -> the entry's row
    auto tr_sum = [&](const float4 a, const float4 b, const float4 c4, const float4 d, int entry, bool live) {
        float t = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) +
                  (((c4.x + c4.y) + (c4.z + c4.w)) + ((d.x + d.y) + (d.z + d.w)));
        t += gs_dpp<0xb1, 0xf, 0xf>(t);   // quad_perm [1,0,3,2]
        t += gs_dpp<0x4e, 0xf, 0xf>(t);   // quad_perm [2,3,0,1]
        if (tr_owner && live) atomicAdd(&s_acc[entry][lane >> 2], t);
    };
    float pend_x[12];
    int pend_k = 0;
    int pend = 0;   // wave-uniform: 1 while a hit entry's partials are waiting in pend_x (REDUCE == 2)
    float sink = 0.f;
#if GS_MFMA_REDUCE
    const GsMfmaReduceConsts mfma_consts = gs_mfma_reduce_consts();
#endif

    auto keep = [&](const float4 r0, const float4 r1) {
        return gs_entry_in_tile(r0, r1, tc.tile_u, tc.tile_v, tw, th, filter);
    };
    auto store = [&](int slot_, int j, int o, const float4 r0, const float4 r1, const float4 r2, const float4 r3) {
        float4 P, Q, colour;
        gs_stage_backward(r0, r1, r2, r3, P, Q, colour);
        s_p[slot_] = P; s_q[slot_] = Q; s_c[slot_] = colour;
        if (STAGED) s_j[slot_] = j;
        s_o[slot_] = o;
    };

    int pos = end - 1;   // next list position to examine, walking down to `start`
    while (pos >= start) {
        __syncthreads();  // previous round fully flushed before its LDS is reused
        int nbuf = 0;
        const int batch_first = pos;   // direct path: staged entry k sits at list position batch_first - k
        if (STAGED)
            while (nbuf < BATCH && pos >= start)
                gs_fill_step<-1>(payload, attrs, pos, start, nbuf, s_cnt, s_next, keep, store);
        else
            gs_direct_step<-1>(payload, attrs, pos, start, nbuf, store);
        {
            const int padded = (nbuf + GROUP - 1) & ~(GROUP - 1);
            if (tid < padded - nbuf) {   // inert padding: s_hi = -inf, no quadratic form passes the hit test
                s_p[nbuf + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
                s_q[nbuf + tid] = make_float4(0.f, 0.f, 0.f, -__builtin_inff());
                s_j[nbuf + tid] = -1;
            }
            float4 *z = reinterpret_cast<float4 *>(&s_acc[tid][0]);
            z[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        for (int k = 0; k < nbuf; k += GROUP_BWD) {
            // descending positions: the whole group lies behind this wave's pixels
            int jg[GROUP_BWD];
            if (STAGED) {
                static_assert(GROUP_BWD % 4 == 0, "list positions are read four at a time");
#pragma unroll
                for (int i = 0; i < GROUP_BWD; i += 4) {
                    const int4 j4 = *reinterpret_cast<const int4 *>(&s_j[k + i]);
                    jg[i] = j4.x; jg[i + 1] = j4.y; jg[i + 2] = j4.z; jg[i + 3] = j4.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < GROUP_BWD; ++i) jg[i] = batch_first - (k + i);
            }
            if (jg[GROUP_BWD - 1] >= wave_end) continue;
            // group evaluation: LDS reads + quadratic forms of GROUP entries are independent and overlap
            v2f alpha[GROUP_BWD], m0[GROUP_BWD], m1[GROUP_BWD];   // alpha[i]: the form s = -2 e until the entry's hit test, then alpha
            float amp[GROUP_BWD], shi[GROUP_BWD];
#pragma unroll
            for (int i = 0; i < GROUP_BWD; ++i) {
                const float4 Q = s_q[k + i];
                alpha[i] = gs_pair_form_backward(s_p[k + i], Q.x, px, py, m0[i], m1[i]);
                amp[i] = Q.y; shi[i] = Q.w;
            }
            // (A) the decisions of the whole group: RAS:631 in the exponent's domain -- s <= s_hi is e >= e_lo, below which the
            // reference skips the pair for certain (gs_common.h) -- and RAS:618 (effective range).  Only an entry that may hit a
            // pixel of this wave has its exponentials evaluated; one whose alpha then does not reach EPS_HI on every such pixel
            // is noted and settled before any entry is processed -- between the evaluation and the hit path, where few
            // registers are live
            // (the decisions are kept as lane MASKS in scalar registers -- gs_ballot / inverse_ballot: as bools they went through
            // the vector unit and back, four instructions per entry)
            unsigned long long ma0[GROUP_BWD], ma1[GROUP_BWD];
            unsigned bracketed = 0u;   // wave-uniform: bit i = entry k + i has an alpha inside the bracket
            GS_STAT(GS_STAT_BWD_ENTRIES, GROUP_BWD);
#pragma unroll
            for (int i = 0; i < GROUP_BWD; ++i) {
                ma0[i] = gs_ballot(alpha[i].x <= shi[i]) & gs_ballot(jg[i] < last0);
                ma1[i] = gs_ballot(alpha[i].y <= shi[i]) & gs_ballot(jg[i] < last1);
                if ((ma0[i] | ma1[i]) != 0ull) {   // (wave-uniform)
                    alpha[i] = gs_pair_alpha_from_form(alpha[i], amp[i]);
                    if (((ma0[i] & ~gs_ballot(alpha[i].x >= EPS_HI)) | (ma1[i] & ~gs_ballot(alpha[i].y >= EPS_HI))) != 0ull)
                        bracketed |= 1u << i;
                }
            }
            if (bracketed != 0u) {
                // rare (about a thousand (wave, entry) visits per full-size frame): the reference's BACKWARD expression decides
                // (same exponent -- UTL:336-339 -- correctly rounded exp, * rescale * opacity).  The reference's two passes round
                // the exponent differently and so may decide a pair differently; each pass of this library follows its counterpart.
#pragma clang loop unroll(disable)
                for (int ii = 0; ii < GROUP_BWD; ++ii) {
                    if (((bracketed >> ii) & 1u) == 0u) continue;
                    GS_STAT(GS_STAT_BWD_BRACKETED, 1);
                    const int e = k + ii;
                    v2f ex, mm0, mm1;
                    const float4 Q = s_q[e];
                    const v2f al = gs_pair_alpha_backward(s_p[e], Q, px, py, ex, mm0, mm1);
                    bool r0 = al.x >= EPS_ALPHA, r1 = al.y >= EPS_ALPHA;
                    const bool in0 = al.x >= EPS_LO && al.x < EPS_HI, in1 = al.y >= EPS_LO && al.y < EPS_HI;
                    if (gs_ballot(in0 || in1) != 0ull) {
                        GS_STAT(GS_STAT_BWD_EXACT_ALPHA, 1);
                        const float opacity = s_c[e].w;
                        const float rescale = reinterpret_cast<const float *>(attrs + 4 * (size_t)s_o[e])[15];
#pragma clang loop unroll(disable)
                        for (int c = 0; c < 2; ++c) {
                            const float exact = gs_alpha_reference(c ? ex.y : ex.x, rescale, opacity);
                            if (c ? in1 : in0) { if (c) r1 = exact >= EPS_ALPHA; else r0 = exact >= EPS_ALPHA; }
                        }
                    }
                    const int jj = STAGED ? s_j[e] : batch_first - e;
                    const unsigned long long mr0 = gs_ballot(r0 && jj < last0), mr1 = gs_ballot(r1 && jj < last1);
#pragma unroll
                    for (int i = 0; i < GROUP_BWD; ++i)
                        if (i == ii) { ma0[i] = mr0; ma1[i] = mr1; }
                }
            }
            // (B) the hit path, entry by entry
#pragma unroll
            for (int i = 0; i < GROUP_BWD; ++i) {
                const unsigned long long mh0 = ma0[i], mh1 = ma1[i];
                if ((mh0 | mh1) == 0ull) continue;  // wave-uniform skip: no pixel of this wave is touched
                const bool hit0 = __builtin_amdgcn_inverse_ballot_w64(mh0), hit1 = __builtin_amdgcn_inverse_ballot_w64(mh1);
#if GS_STATS
                {
                    GS_STAT(GS_STAT_BWD_HIT_ENTRIES, 1);
                    GS_STAT(GS_STAT_BWD_HIT_PIXELS, __popcll(mh0) + __popcll(mh1));
                    GS_STAT(GS_STAT_BWD_HIT_LANES, __popcll(mh0 | mh1));
                    GS_STAT(GS_STAT_BWD_HIT_BLOCKS, (((mh0 | mh1) & 0x0f0f0f0f0f0f0f0full) != 0ull) + (((mh0 | mh1) & 0xf0f0f0f0f0f0f0f0ull) != 0ull));
                }
#endif
                // The twelve per-lane partial sums of this entry (in-lane sums over the lane's two pixels), in the order of
                // the accumulator record: v0, v1 (dL/dmu), c00, c01, c11 (2 dL/dcov), gr, gg, gb (dL/drgb), w, |v|, count, 0.
                // (REDUCE 4) the PREVIOUS hit entry's rows are fetched now and summed behind this entry's arithmetic: nothing waits
                // for the LDS.  (The first hit entry of a round fetches rows that nobody adds.)
                // (the colour row FIRST: LDS reads return in order, and the arithmetic below waits for this one only.  Read as all
                // 16 bytes -- volatile, or the compiler trims the unused opacity off and issues ds_read_b96, which takes the LDS
                // eight cycles where ds_read_b128 takes four; the kernel is short of LDS cycles since its reduction moved there)
                const gs_v4f cv = *reinterpret_cast<const volatile gs_v4f *>(&s_c[k + i]);
                const float4 c = make_float4(cv.x, cv.y, cv.z, cv.w);
                float4 qa, qb, qc, qd;
                if constexpr (REDUCE == 4) {
                    __builtin_amdgcn_wave_barrier();
                    const float4 *src = reinterpret_cast<const float4 *>(&s_tr[tid >> 6][0] + tr_read);
                    qa = src[0]; qb = src[1]; qc = src[2]; qd = src[3];
                    __builtin_amdgcn_wave_barrier();   // (this entry's stores stay below these reads)
                }
                v2f pq[11];   // the eleven packed (two-pixel) partials of this entry
                {
                    // alpha = 0 for a pixel that is not hit makes its whole update an exact no-op
                    // (1/(1-0) = 1, a*T = 0); only dL/dalpha needs an explicit mask.
                    const v2f h = {hit0 ? 1.f : 0.f, hit1 ? 1.f : 0.f};
                    const v2f al = {hit0 ? __builtin_amdgcn_fmed3f(alpha[i].x, 0.f, CLAMP_ALPHA) : 0.f,
                                    hit1 ? __builtin_amdgcn_fmed3f(alpha[i].y, 0.f, CLAMP_ALPHA) : 0.f};
                    const v2f one_m = splat(1.f) - al;
                    const v2f inv1m = {__builtin_amdgcn_rcpf(one_m.x), __builtin_amdgcn_rcpf(one_m.y)};
                    T = T * inv1m;  // RAS:643, T_i = T_{i+1} / (1 - alpha_i)
                    const v2f aT = al * T;
                    const v2f gr = aT * Gr, gg = aT * Gg, gb = aT * Gb;
                    // dL/dalpha = sum_c (c_c T - w_c/(1-alpha)) G_c = T (c.G) - S/(1-alpha)       (RAS:652-657)
                    const v2f cg = fma2(splat(c.z), Gb, fma2(splat(c.y), Gg, splat(c.x) * Gr));
                    const v2f dLda = fma2(T, cg, -(S * inv1m)) * h;
                    S = fma2(cg, aT, S);
                    // w = dL/dg * g = dL/dalpha * opacity * g = dL/dalpha * alpha (unclamped).  dL/dlogit = (1-o) w and the
                    // factor 1/2 of dg/dcov are per-Gaussian constants: applied once per (tile, Gaussian) in the flush.
                    const v2f w = dLda * alpha[i];
                    // UTL:331-348: m = conic @ d (from the evaluation of alpha)
                    const v2f v0 = w * m0[i], v1 = w * m1[i];  // dL/dmu = dL/dg * g * (conic @ d)   (UTL:343)
                    asm("v_add_f32 %0, |%1|, %0" : "+v"(mag_u.x) : "v"(v0.x));
                    asm("v_add_f32 %0, |%1|, %0" : "+v"(mag_u.y) : "v"(v0.y));
                    asm("v_add_f32 %0, |%1|, %0" : "+v"(mag_v.x) : "v"(v1.x));
                    asm("v_add_f32 %0, |%1|, %0" : "+v"(mag_v.y) : "v"(v1.y));
                    const v2f c00 = v0 * m0[i], c01 = v0 * m1[i], c11 = v1 * m1[i];  // 2 dL/dcov (UTL:345-346)
                    const v2f n2 = fma2(v1, v1, v0 * v0);
                    const v2f nv = {__builtin_amdgcn_sqrtf(n2.x), __builtin_amdgcn_sqrtf(n2.y)};  // v_sqrt_f32, 1 ulp
                    if (DEBUG) {
                        const unsigned hv = (unsigned)(s_o[k + i] + 1) * GS_HASH_MUL;
                        dc0 += hit0 ? 1u : 0u; dh0 += hit0 ? hv : 0u;
                        dc1 += hit1 ? 1u : 0u; dh1 += hit1 ? hv : 0u;
                    }
                    pq[0] = v0; pq[1] = v1; pq[2] = c00; pq[3] = c01; pq[4] = c11; pq[5] = gr; pq[6] = gg; pq[7] = gb;
                    pq[8] = w; pq[9] = nv; pq[10] = h;
                }
                // in-lane sums over the lane's two pixels (inline asm: the vectoriser would otherwise shuffle the operands
                // into v_pk_add_f32 pairs with more moves than it saves adds)
                // (value 9 -- the two square roots -- is added by the compiler: a transcendental result needs a wait state before
                // another VALU instruction reads it, and the hazard recogniser does not look inside an asm statement: scheduled
                // straight behind v_sqrt_f32 the asm form read a stale register once in ~30 hit entries, REDUCE 4's schedule)
                auto partials_of_entry = [&](float (&x)[12]) {
#pragma unroll
                    for (int n = 0; n < 11; ++n)
                        if (n != 9) asm("v_add_f32 %0, %1, %2" : "=v"(x[n]) : "v"(pq[n].x), "v"(pq[n].y));
                    x[9] = pq[9].x + pq[9].y;
                    x[11] = 0.f;
                };
                if constexpr (REDUCE == 0) {   // measurement only: what the kernel costs without the cross-lane sums
                    float x[12];
                    partials_of_entry(x);
#pragma unroll
                    for (int n = 0; n < 11; ++n) sink += x[n];
                } else if constexpr (REDUCE == 2) {
                    // Two hit entries share one reduce-scatter: the first one's partials wait in registers (pend_x) until the
                    // wave meets its next hit entry of this round; a leftover is reduced alone before the flush.  Both
                    // reduce-scatters add a value's 64 lanes in the same tree (gs_common.h), so an entry's sums are the same
                    // bits whether it found a partner or not.
                    // (the two arms are kept apart by distinct asm markers: merged, the compiler computes into scratch
                    // registers and copies twelve values into pend_x)
                    if (__builtin_amdgcn_readfirstlane(pend)) {
                        asm volatile("; pair: second entry");
                        float x[12];
                        partials_of_entry(x);
                        float w0, w1, w2;
                        gs_wave_reduce12_pair(pend_x, x, w0, w1, w2);
                        if (pair_tail) {
                            float *A = &s_acc[pair_b_sel ? k + i : pend_k][pair_vi];
                            atomicAdd(A, w0);
                            atomicAdd(A + 4, w1);
                            atomicAdd(A + 8, w2);
                        }
                        pend = 0;
                    } else {
                        asm volatile("; pair: first entry");
                        partials_of_entry(pend_x);
                        pend_k = k + i;
                        pend = 1;
                    }
                } else if constexpr (REDUCE == 3) {
                    // The LDS pipe instead of the vector unit (tools/ubench/valu_exec_mask.hip: a permlane swap costs the SIMD 3.5 ns,
                    // a DPP add 1.9, a plain add 1.1 -- the pair reduce-scatter is 53 ns of VALU time per hit entry, a quarter of the
                    // kernel -- while the LDS pipe idles 80 % of the time): every lane stores its eleven partials as eleven rows of
                    // 64, lane 4 n + p reads the p-th quarter of row n back (four 16-byte reads, conflict-free with rows 68 floats
                    // apart), adds its sixteen values in a fixed tree, two quad-permute adds combine the quarters, and lane 4 n adds
                    // the wave's total of value n to the entry's row (the two waves of the tile meet there: a + b = b + a).  A wave's
                    // LDS operations execute in program order, so the reads see the stores without a barrier.
                    float x[12];
                    partials_of_entry(x);
                    float *rows = &s_tr[tid >> 6][0];
#pragma unroll
                    for (int n = 0; n < GS_TR_ROWS; ++n) rows[n * GS_TR_STRIDE + lane] = x[n];
                    __builtin_amdgcn_wave_barrier();   // (no instruction: the compiler keeps the reads below the stores)
                    const float4 *src = reinterpret_cast<const float4 *>(rows + tr_read);
                    const float4 a = src[0], b = src[1], c4 = src[2], d = src[3];
                    float t = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) +
                              (((c4.x + c4.y) + (c4.z + c4.w)) + ((d.x + d.y) + (d.z + d.w)));
                    t += gs_dpp<0xb1, 0xf, 0xf>(t);   // quad_perm [1,0,3,2]
                    t += gs_dpp<0x4e, 0xf, 0xf>(t);   // quad_perm [2,3,0,1]
                    __builtin_amdgcn_wave_barrier();   // (the next entry's stores stay below these reads)
                    if (tr_owner) atomicAdd(&s_acc[k + i][lane >> 2], t);
                } else if constexpr (REDUCE == 5) {
                    // REDUCE 3 with the read-back confined to the 44 lanes that have a row to read (the other twenty re-read row 10
                    // and threw it away: a third of the read-back's LDS cycles)
                    float x[12];
                    partials_of_entry(x);
                    float *rows = &s_tr[tid >> 6][0];
#pragma unroll
                    for (int n = 0; n < GS_TR_ROWS; ++n) rows[n * GS_TR_STRIDE + lane] = x[n];
                    __builtin_amdgcn_wave_barrier();
                    float t = 0.f;
                    if (lane < 4 * GS_TR_ROWS) {
                        const float4 *src = reinterpret_cast<const float4 *>(rows + tr_read);
                        const float4 a = src[0], b = src[1], c4 = src[2], d = src[3];
                        t = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) +
                            (((c4.x + c4.y) + (c4.z + c4.w)) + ((d.x + d.y) + (d.z + d.w)));
                    }
                    t += gs_dpp<0xb1, 0xf, 0xf>(t);   // quad_perm [1,0,3,2]
                    t += gs_dpp<0x4e, 0xf, 0xf>(t);   // quad_perm [2,3,0,1]
                    __builtin_amdgcn_wave_barrier();
                    if (tr_owner) atomicAdd(&s_acc[k + i][lane >> 2], t);
                } else if constexpr (REDUCE == 6) {
                    // Half the LDS traffic of REDUCE 3: one v_permlane32_swap + one add per PAIR of values leaves the pair's 2 x 32
                    // half-wave sums in one register (lanes 0-31: value 2n, lanes 32-63: value 2n + 1) -- six stores of 64 instead
                    // of eleven, and lane 4 v + p reads 8 floats (two 16-byte reads) of value v's 32 instead of 16 of its 64
                    float x[12];
                    partials_of_entry(x);
                    float *rows = &s_tr[tid >> 6][0];
                    // (one asm block, as gs_wave_reduce12: the hazard recogniser does not look inside -- the leading s_nop covers
                    //  "VALU write -> permlane read", every add is six instructions behind the swap that feeds it)
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
                        "v_add_f32 %10, %10, %11"
                        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                          "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
#pragma unroll
                    for (int n = 0; n < 6; ++n) rows[n * GS_TR_STRIDE + lane] = x[2 * n];   // [value 2n: 32 sums | value 2n + 1: 32 sums]
                    __builtin_amdgcn_wave_barrier();
                    const float4 *src = reinterpret_cast<const float4 *>(rows + tr_read_half);
                    const float4 a = src[0], b = src[1];
                    float t = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
                    t += gs_dpp<0xb1, 0xf, 0xf>(t);   // quad_perm [1,0,3,2]
                    t += gs_dpp<0x4e, 0xf, 0xf>(t);   // quad_perm [2,3,0,1]
                    __builtin_amdgcn_wave_barrier();
                    if (tr_owner) atomicAdd(&s_acc[k + i][lane >> 2], t);
                } else if constexpr (REDUCE == 4) {
                    // REDUCE 3 with the LDS round trip taken off the wave's critical path: an entry's rows are stored here and read
                    // back at the top of the NEXT hit entry's path (above) -- one buffer is enough, the wave's LDS operations execute
                    // in program order: the fetch of the old rows is ahead of these stores -- or by the drain behind the round's loop.
                    float x[12];
                    partials_of_entry(x);
                    // (an empty asm that "rewrites" one fetched value once this entry's partials exist: the compiler cannot start
                    // the sum -- and wait for the fetch -- before the arithmetic above)
                    asm volatile("" : "+v"(qa.x), "+v"(qb.x), "+v"(qc.x), "+v"(qd.x) : "v"(x[0]), "v"(x[10]));
                    tr_sum(qa, qb, qc, qd, pend_k, __builtin_amdgcn_readfirstlane(pend) != 0);
                    float *rows = &s_tr[tid >> 6][0];
#pragma unroll
                    for (int n = 0; n < GS_TR_ROWS; ++n) rows[n * GS_TR_STRIDE + lane] = x[n];
                    __builtin_amdgcn_wave_barrier();
                    pend_k = k + i;
                    pend = 1;
                } else {
                    float x[12];
                    partials_of_entry(x);
                    // the 12-value reduce-scatter over the 64 lanes (gs_common.h); row totals land in lane 15 of each row:
                    // t0 (v0, c00, v1, c01)  t1 (c11, gg, gr, gb)  t2 (w, count, |v|, 0)
#if GS_MFMA_REDUCE
                    // matrix-pipe variant: lane n (< 11) ends up with the wave total of value n
                    const float tot = gs_wave_reduce12_mfma(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8], x[9], x[10],
                                                            0.f, mfma_consts);
                    if (lane < 11) atomicAdd(&s_acc[k + i][lane], tot);
#else
                    float t0, t1, t2;
                    gs_wave_reduce12(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8], x[9], x[10], 0.f, t0, t1, t2);
                    if (row_tail) {
                        float *A = &s_acc[k + i][slot];
                        atomicAdd(A, t0);
                        atomicAdd(A + 4, t1);
                        atomicAdd(A + 8, t2);
                    }
#endif
                }
            }
        }
        if (REDUCE == 4 && __builtin_amdgcn_readfirstlane(pend)) {   // the round's last hit entry is still in its rows
            const float4 *src = reinterpret_cast<const float4 *>(&s_tr[tid >> 6][0] + tr_read);
            const float4 qa = src[0], qb = src[1], qc = src[2], qd = src[3];
            tr_sum(qa, qb, qc, qd, pend_k, true);
            pend = 0;
        }
        if (REDUCE == 2 && __builtin_amdgcn_readfirstlane(pend)) {   // the round's odd hit entry
            float t0, t1, t2;
            gs_wave_reduce12(pend_x[0], pend_x[1], pend_x[2], pend_x[3], pend_x[4], pend_x[5], pend_x[6], pend_x[7],
                             pend_x[8], pend_x[9], pend_x[10], 0.f, t0, t1, t2);
            if (row_tail) {
                float *A = &s_acc[pend_k][slot];
                atomicAdd(A, t0);
                atomicAdd(A + 4, t1);
                atomicAdd(A + 8, t2);
            }
            pend = 0;
        }
        __syncthreads();
        // flush: thread k owns staged entry k -> one 48-B store into the (Gaussian, tile) slot
        if (tid < nbuf) {
            const float4 *Sa = reinterpret_cast<const float4 *>(&s_acc[tid][0]);
            float4 r0 = Sa[0], r1 = Sa[1], r2 = Sa[2];
            if (r2.z > 0.f) {
                const float4 a = s_p[tid];
                int t0u, t1u, t0v, t1v;
                gs_tile_box(a.x, a.y, s_q[tid].z, tw, th, t0u, t1u, t0v, t1v);
                const int dst_slot = slot_offsets[s_o[tid]] + (t1v - t0v) * (tc.tile_u - t0u) + (tc.tile_v - t0v);
                float4 *dst = partials + 3 * (size_t)dst_slot;
                r0.z *= 0.5f; r0.w *= 0.5f; r1.x *= 0.5f;  // dg/dcov = 0.5 g (m m^T)
                r2.x *= (1.f - s_c[tid].w);                // dL/dlogit = (1 - opacity) * sum(w)
                r2.z = __builtin_bit_cast(float, (int)r2.z);  // pixel count: float sum -> int32 bits (layout of `acc`)
                dst[0] = r0;
                dst[1] = r1;
                dst[2] = r2;
                slot_flags[dst_slot] = 1;
            }
        }
    }
    if (REDUCE == 0) mag_u.x += 1e-30f * sink;
    magnitude_image[2 * p] = mag_u.x;
    magnitude_image[2 * p + 1] = mag_v.x;
    magnitude_image[2 * p + 2] = mag_u.y;
    magnitude_image[2 * p + 3] = mag_v.y;
    if (DEBUG) {
        debug_hits[2 * p] = dc0; debug_hits[2 * p + 1] = dh0;
        debug_hits[2 * p + 2] = dc1; debug_hits[2 * p + 3] = dh1;
    }
}

// ------------------------------------------------------------------------------- backward, ONE wave per tile
// Per-tile lists taken as they are (bin_shift 0, no filter: the walked lists the forward pass writes out, or the
// reference's own keys) on grids large enough to fill the chip.  One wave64 owns the whole 16 x 16 tile -- FOUR pixels per
// lane: lane (c, r) holds columns 2c, 2c + 1 of rows r and r + 8 ("set A" and "set B", each a packed pair exactly as a lane
// of the two-wave kernel holds it) -- because what the two-wave kernel pays PER WAVE AND ENTRY is then paid once per tile and
// entry:
//   * the cross-lane sums of a hit entry (a quarter of the two-wave kernel's time): one reduction per (tile, entry) instead
//     of one per half-tile, and no combining of two waves' totals (plain LDS stores instead of ds_add_f32, no zero-fill);
//   * the two 16-byte broadcast reads of the staged record, the colour row, the x-parts of m = conic @ d (the two sets share
//     their columns), every scalar decision and branch of the group loop;
//   * and a lone wave needs no workgroup barrier at all: its LDS operations execute in program order.
// The cross-lane sums go through LDS (the REDUCE 3 arm of the two-wave kernel: stores of eleven rows, four 16-byte reads per
// lane, a fixed tree of fifteen adds, two quad-permute adds), which costs the vector unit 23 ns per hit entry where the
// permlane / DPP reduce-scatter costs 58 (tools/ubench/valu_exec_mask.hip) -- and with one wave per tile the LDS pipe has
// the room (two waves per tile doing the same keep it 80 % busy: profiles/r06_backward_arms.md).
// Per pixel the arithmetic is the two-wave kernel's, operation by operation: decisions, the |grad uv| image and the debug
// hashes are bit-identical; a slot's sums add the same per-pixel terms in another order.
constexpr int WIDE_BATCH = 64;   // staged entries per round (one per lane)
constexpr int WIDE_GROUP = 2;    // entries evaluated together
#ifndef GS_BWD_WIDE_MIN_WAVES
#define GS_BWD_WIDE_MIN_WAVES 4
#endif
#ifndef GS_BWD_WIDE_PRIO
#define GS_BWD_WIDE_PRIO 1   // 0: every tile at the default priority (A/B)
#endif
struct WideSet {   // one packed pixel pair of a lane
    v2f T, S, Gr, Gg, Gb, mag_u, mag_v;
    int last0, last1;
    float py;
    unsigned dh0, dh1, dc0, dc1;
};
template <bool DEBUG>
__global__ __launch_bounds__(GS_WAVE, GS_BWD_WIDE_MIN_WAVES) void blend_backward_wide_kernel(
    const int32_t *__restrict__ tile_start, const int32_t *__restrict__ payload, const float4 *__restrict__ attrs,
    const float *__restrict__ grad_image, const float *__restrict__ acc_alpha, const int32_t *__restrict__ last_effective,
    int width, int height, int row_begin, int row_step, const int32_t *__restrict__ slot_offsets,
    float4 *__restrict__ partials, uint8_t *__restrict__ slot_flags, float *__restrict__ magnitude_image,
    uint32_t *__restrict__ debug_hits, const int32_t *__restrict__ tile_order) {
    __shared__ float4 s_p[WIDE_BATCH], s_q[WIDE_BATCH], s_c[WIDE_BATCH];   // (gs_stage_backward)
    __shared__ int s_o[WIDE_BATCH];
    __shared__ __attribute__((aligned(16))) float s_acc[WIDE_BATCH][GS_ACC_STRIDE];   // [entry][value], written once per hit entry
    __shared__ __attribute__((aligned(16))) float s_tr[GS_TR_ROWS * GS_TR_STRIDE];   // the rows of the LDS transpose
    const int tw = width / GS_TILE_WIDTH, th = height / GS_TILE_HEIGHT;
    const TileCoord tc = owned_tile(tw, row_begin, row_step, tile_order);
    // A tile is ONE wave's serial chain here, and the longest chains (2.4 x the mean at the headline scene) are what the launch
    // lasts if they have to share their SIMD's issue slots evenly with three lighter waves.  Tiles are dispatched longest walk
    // first (tile_order), so the position in the grid is the work estimate: the first eighth of the grid runs at priority 3,
    // the next eighth at 2, the next quarter at 1 -- the long chains finish early and the short ones fill in behind them.
#if GS_BWD_WIDE_PRIO
    if (tile_order != nullptr) {
        const unsigned b = blockIdx.x, n = gridDim.x;
        if (8 * b < n) __builtin_amdgcn_s_setprio(3);
        else if (4 * b < n) __builtin_amdgcn_s_setprio(2);
        else if (2 * b < n) __builtin_amdgcn_s_setprio(1);
    }
#endif
    const int lane = threadIdx.x;
    const int pu = tc.tile_u * GS_TILE_WIDTH + 2 * (lane & 7);   // left pixel of both pairs
    const int pv = tc.tile_v * GS_TILE_HEIGHT + (lane >> 3);     // row of set A; set B: eight rows below
    // (pixel index of set h, recomputed where it is used -- at both ends of the kernel -- rather than kept in registers)
    auto pixel_of = [&](int h) { return (size_t)(pv + 8 * h) * width + pu; };
    const int start = tile_start[tc.tile_id];
    const v2f px = {(float)pu + 0.5f, (float)pu + 1.5f};
    WideSet st[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const size_t p = pixel_of(h);
        st[h].py = (float)(pv + 8 * h) + 0.5f;
        st[h].last0 = last_effective[p]; st[h].last1 = last_effective[p + 1];
        st[h].T = (v2f){1.0f - acc_alpha[p], 1.0f - acc_alpha[p + 1]};
        st[h].S = splat(0.f);
        const float *gi = grad_image + 3 * p;
        st[h].Gr = (v2f){gi[0], gi[3]}; st[h].Gg = (v2f){gi[1], gi[4]}; st[h].Gb = (v2f){gi[2], gi[5]};
        st[h].mag_u = splat(0.f); st[h].mag_v = splat(0.f);
        st[h].dh0 = st[h].dh1 = st[h].dc0 = st[h].dc1 = 0u;
    }
    // no pixel of the tile touches a list position at or beyond the tile-wide max of `last`
    int mx = max(max(st[0].last0, st[0].last1), max(st[1].last0, st[1].last1));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, GS_WAVE));
    const int end = __builtin_amdgcn_readfirstlane(mx);
    // the LDS transpose (see the REDUCE 3 arm of blend_backward_kernel): lane 4 n + p reads floats [16 p, 16 p + 16) of row n
    const int tr_read = min(lane >> 2, GS_TR_ROWS - 1) * GS_TR_STRIDE + 16 * (lane & 3);
    const bool tr_owner = (lane & 3) == 0 && lane < 4 * GS_TR_ROWS;

    int pos = end - 1;   // next list position to stage, walking down to `start`
    int o_next = pos - lane >= start ? payload[pos - lane] : 0;   // (the next round's list entries are fetched a round ahead)
    while (pos >= start) {
        const int nbuf = min(WIDE_BATCH, pos - start + 1);
        const int batch_first = pos;   // staged entry k sits at list position batch_first - k
        {
            const int o = o_next;
            const bool valid = lane < nbuf;
            float4 r0, r1, r2, r3;
            if (valid) {
                const float4 *g = attrs + 4 * (size_t)o;
                r0 = g[0]; r1 = g[1]; r2 = g[2]; r3 = g[3];
            }
            pos -= WIDE_BATCH;
            o_next = pos - lane >= start ? payload[pos - lane] : 0;
            float4 P = make_float4(0.f, 0.f, 0.f, 0.f), Q = P, colour = P;   // (inert padding: amplitude 0 -> alpha 0, never a hit)
            if (valid) gs_stage_backward(r0, r1, r2, r3, P, Q, colour);
            __builtin_amdgcn_wave_barrier();   // (the previous round's flush has read its rows: program order)
            s_p[lane] = P; s_q[lane] = Q; s_c[lane] = colour;
            s_o[lane] = o;
            __builtin_amdgcn_wave_barrier();
        }
        unsigned long long hit_entries = 0ull;   // wave-uniform: staged entries whose row of s_acc holds this round's sums
        for (int k = 0; k < nbuf; k += WIDE_GROUP) {
            // group evaluation: the LDS reads and exponentials of the group's entries are independent and overlap
            v2f alpha[WIDE_GROUP][2], m0[WIDE_GROUP][2], m1[WIDE_GROUP][2];
            unsigned long long ma0[WIDE_GROUP][2], ma1[WIDE_GROUP][2];
            unsigned bracketed = 0u;   // bit 2 i + h: entry k + i has an alpha of set h inside the bracket around 1/255
#pragma unroll
            for (int i = 0; i < WIDE_GROUP; ++i) {
                const float4 P = s_p[k + i], Q = s_q[k + i];
                // UTL:336-339, operation by operation as gs_pair_alpha_backward; the products with dx serve both sets
                const v2f dx = px - splat(P.x);
                const v2f ax = splat(P.z) * dx, bx = splat(Q.x) * dx;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float dy = st[h].py - P.y;
                    m0[i][h] = ax + splat(Q.x * dy);
                    m1[i][h] = bx + splat(P.w * dy);
                    const v2f sq = dx * m0[i][h] + splat(dy) * m1[i][h];
                    const v2f e2 = sq * splat(-0.5f * GS_LOG2E);
                    alpha[i][h] = (v2f){__builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y)} * splat(Q.y);
                    ma0[i][h] = gs_ballot(alpha[i][h].x >= EPS_LO); ma1[i][h] = gs_ballot(alpha[i][h].y >= EPS_LO);
                    if ((ma0[i][h] | ma1[i][h]) != 0ull &&
                        ((ma0[i][h] ^ gs_ballot(alpha[i][h].x >= EPS_HI)) | (ma1[i][h] ^ gs_ballot(alpha[i][h].y >= EPS_HI))) != 0ull)
                        bracketed |= 1u << (2 * i + h);
                }
            }
            if (bracketed != 0u) {   // rare: the reference's BACKWARD expression decides (as in blend_backward_kernel)
#pragma clang loop unroll(disable)
                for (int b = 0; b < 2 * WIDE_GROUP; ++b) {
                    if (((bracketed >> b) & 1u) == 0u) continue;
                    const int e = k + (b >> 1), hsel = b & 1;
                    v2f ex, mm0, mm1;
                    const float4 Q = s_q[e];
                    const float pyh = hsel ? st[1].py : st[0].py;
                    const v2f al = gs_pair_alpha_backward(s_p[e], Q, px, pyh, ex, mm0, mm1);
                    bool r0 = al.x >= EPS_ALPHA, r1 = al.y >= EPS_ALPHA;
                    const bool in0 = al.x >= EPS_LO && al.x < EPS_HI, in1 = al.y >= EPS_LO && al.y < EPS_HI;
                    if (gs_ballot(in0 || in1) != 0ull) {
                        const float opacity = s_c[e].w;
#pragma clang loop unroll(disable)
                        for (int c = 0; c < 2; ++c) {
                            const float exact = gs_alpha_reference(c ? ex.y : ex.x, reinterpret_cast<const float *>(attrs + 4 * (size_t)s_o[e])[15], opacity);
                            if (c ? in1 : in0) { if (c) r1 = exact >= EPS_ALPHA; else r0 = exact >= EPS_ALPHA; }
                        }
                    }
                    const unsigned long long mr0 = gs_ballot(r0), mr1 = gs_ballot(r1);
#pragma unroll
                    for (int i = 0; i < WIDE_GROUP; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            if (2 * i + h == b) { ma0[i][h] = mr0; ma1[i][h] = mr1; }
                }
            }
#pragma unroll
            for (int i = 0; i < WIDE_GROUP; ++i) {
                if ((ma0[i][0] | ma1[i][0] | ma0[i][1] | ma1[i][1]) == 0ull) continue;   // wave-uniform: no pixel of the tile is touched
                const int jj = batch_first - (k + i);
                unsigned long long mh0[2], mh1[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {   // RAS:618 (effective range)
                    mh0[h] = ma0[i][h] & gs_ballot(jj < st[h].last0);
                    mh1[h] = ma1[i][h] & gs_ballot(jj < st[h].last1);
                }
                if ((mh0[0] | mh1[0] | mh0[1] | mh1[1]) == 0ull) continue;
                const float4 c = s_c[k + i];
                // the eleven per-lane partial sums of this entry over the lane's four pixels, in the order of the accumulator
                // record: v0, v1 (dL/dmu), c00, c01, c11 (2 dL/dcov), gr, gg, gb (dL/drgb), w, |v|, count
                float xs[2][GS_TR_ROWS];   // per set: in-lane sums over its pixel pair
                const bool any_a = (mh0[0] | mh1[0]) != 0ull, any_b = (mh0[1] | mh1[1]) != 0ull;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (!(h ? any_b : any_a)) continue;   // wave-uniform: this half of the tile is not touched
                    WideSet &w_ = st[h];
                    const bool hit0 = __builtin_amdgcn_inverse_ballot_w64(mh0[h]), hit1 = __builtin_amdgcn_inverse_ballot_w64(mh1[h]);
                    // (the two-wave kernel's hit path, operation by operation)
                    const v2f hm = {hit0 ? 1.f : 0.f, hit1 ? 1.f : 0.f};
                    const v2f al = {hit0 ? __builtin_amdgcn_fmed3f(alpha[i][h].x, 0.f, CLAMP_ALPHA) : 0.f,
                                    hit1 ? __builtin_amdgcn_fmed3f(alpha[i][h].y, 0.f, CLAMP_ALPHA) : 0.f};
                    const v2f one_m = splat(1.f) - al;
                    const v2f inv1m = {__builtin_amdgcn_rcpf(one_m.x), __builtin_amdgcn_rcpf(one_m.y)};
                    w_.T = w_.T * inv1m;  // RAS:643
                    const v2f aT = al * w_.T;
                    const v2f gr = aT * w_.Gr, gg = aT * w_.Gg, gb = aT * w_.Gb;
                    const v2f cg = fma2(splat(c.z), w_.Gb, fma2(splat(c.y), w_.Gg, splat(c.x) * w_.Gr));
                    const v2f dLda = fma2(w_.T, cg, -(w_.S * inv1m)) * hm;   // RAS:652-657
                    w_.S = fma2(cg, aT, w_.S);
                    const v2f wv = dLda * alpha[i][h];
                    const v2f v0 = wv * m0[i][h], v1 = wv * m1[i][h];   // UTL:343
                    w_.mag_u = w_.mag_u + (v2f){__builtin_fabsf(v0.x), __builtin_fabsf(v0.y)};
                    w_.mag_v = w_.mag_v + (v2f){__builtin_fabsf(v1.x), __builtin_fabsf(v1.y)};
                    const v2f c00 = v0 * m0[i][h], c01 = v0 * m1[i][h], c11 = v1 * m1[i][h];   // UTL:345-346
                    const v2f n2 = fma2(v1, v1, v0 * v0);
                    const v2f nv = {__builtin_amdgcn_sqrtf(n2.x), __builtin_amdgcn_sqrtf(n2.y)};
                    if (DEBUG) {
                        const unsigned hv = (unsigned)(s_o[k + i] + 1) * GS_HASH_MUL;
                        w_.dc0 += hit0 ? 1u : 0u; w_.dh0 += hit0 ? hv : 0u;
                        w_.dc1 += hit1 ? 1u : 0u; w_.dh1 += hit1 ? hv : 0u;
                    }
                    const v2f pq[GS_TR_ROWS] = {v0, v1, c00, c01, c11, gr, gg, gb, wv, nv, hm};
                    // (inline asm: left to itself the vectoriser shuffles the operands into v_pk_add_f32 pairs -- fifteen moves for
                    // five packed adds; the leading s_nop covers the transcendental results of value 9, which the hazard
                    // recogniser cannot see an asm statement read)
#pragma unroll
                    for (int n = 0; n < GS_TR_ROWS; ++n) {
                        if (n == 9) asm("s_nop 0\n\tv_add_f32 %0, %1, %2" : "=v"(xs[h][n]) : "v"(pq[n].x), "v"(pq[n].y));
                        else asm("v_add_f32 %0, %1, %2" : "=v"(xs[h][n]) : "v"(pq[n].x), "v"(pq[n].y));
                    }
                }
                // cross-lane sums through LDS: eleven rows of 64 partials (a lane's four pixels in the fixed order (A.x + A.y) +
                // (B.x + B.y)), lane 4 n + p adds the p-th quarter of row n, two quad-permute adds combine the quarters, lane
                // 4 n stores the tile's total of value n in the entry's row
                if (any_a && any_b) {
#pragma unroll
                    for (int n = 0; n < GS_TR_ROWS; ++n) {
                        float both;
                        asm("v_add_f32 %0, %1, %2" : "=v"(both) : "v"(xs[0][n]), "v"(xs[1][n]));
                        s_tr[n * GS_TR_STRIDE + lane] = both;
                    }
                } else if (any_a) {
#pragma unroll
                    for (int n = 0; n < GS_TR_ROWS; ++n) s_tr[n * GS_TR_STRIDE + lane] = xs[0][n];
                } else {
#pragma unroll
                    for (int n = 0; n < GS_TR_ROWS; ++n) s_tr[n * GS_TR_STRIDE + lane] = xs[1][n];
                }
                __builtin_amdgcn_wave_barrier();
                const float4 *src = reinterpret_cast<const float4 *>(s_tr + tr_read);
                const float4 a = src[0], b4 = src[1], c4 = src[2], d4 = src[3];
                float t = (((a.x + a.y) + (a.z + a.w)) + ((b4.x + b4.y) + (b4.z + b4.w))) +
                          (((c4.x + c4.y) + (c4.z + c4.w)) + ((d4.x + d4.y) + (d4.z + d4.w)));
                t += gs_dpp<0xb1, 0xf, 0xf>(t);   // quad_perm [1,0,3,2]
                t += gs_dpp<0x4e, 0xf, 0xf>(t);   // quad_perm [2,3,0,1]
                __builtin_amdgcn_wave_barrier();
                if (tr_owner) s_acc[k + i][lane >> 2] = t;
                hit_entries |= 1ull << (k + i);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // flush: lane k owns staged entry k -> one 48-B store into the (Gaussian, tile) slot
        if ((hit_entries >> lane) & 1ull) {
            const float4 *Sa = reinterpret_cast<const float4 *>(&s_acc[lane][0]);
            float4 r0 = Sa[0], r1 = Sa[1], r2 = Sa[2];
            if (r2.z > 0.f) {
                const float4 a = s_p[lane];
                int t0u, t1u, t0v, t1v;
                gs_tile_box(a.x, a.y, s_q[lane].z, tw, th, t0u, t1u, t0v, t1v);
                const int dst_slot = slot_offsets[s_o[lane]] + (t1v - t0v) * (tc.tile_u - t0u) + (tc.tile_v - t0v);
                float4 *dst = partials + 3 * (size_t)dst_slot;
                r0.z *= 0.5f; r0.w *= 0.5f; r1.x *= 0.5f;  // dg/dcov = 0.5 g (m m^T)
                r2.x *= (1.f - s_c[lane].w);               // dL/dlogit = (1 - opacity) * sum(w)
                r2.z = __builtin_bit_cast(float, (int)r2.z);  // pixel count: float sum -> int32 bits (layout of `acc`)
                r2.w = 0.f;
                dst[0] = r0;
                dst[1] = r1;
                dst[2] = r2;
                slot_flags[dst_slot] = 1;
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const size_t p = pixel_of(h);
        magnitude_image[2 * p] = st[h].mag_u.x;
        magnitude_image[2 * p + 1] = st[h].mag_v.x;
        magnitude_image[2 * p + 2] = st[h].mag_u.y;
        magnitude_image[2 * p + 3] = st[h].mag_v.y;
        if (DEBUG) {
            debug_hits[2 * p] = st[h].dc0; debug_hits[2 * p + 1] = st[h].dh0;
            debug_hits[2 * p + 2] = st[h].dc1; debug_hits[2 * p + 3] = st[h].dh1;
        }
    }
}

// ------------------------------------------------------------------------------- small grids: four waves per tile
// A frame of a few hundred tiles (the 4x / 2x down-sampled first iterations of every training run, TRN:133-148; a 256 x 256
// view; one rank's band of a sharded frame) cannot fill 256 CUs with two waves per tile: every tile walks its list alone on
// two of its CU's four SIMDs and the launch lasts as long as the longest list (10k Gaussians at 256 x 256: 256 tiles, 96 +
// 146 us in the two kernels above).  These arms give a tile FOUR waves -- one pixel per lane, so a list entry costs a wave
// about half the issue slots of the two-pixel form -- for per-tile lists taken as they are (bin_shift 0, no filter: what
// small frames use).  Per pixel they execute the same IEEE operations in the same order as the two-pixel kernels (packed
// instructions are per-component), so image, depth, counts, state and hit decisions are bit-identical; the backward's slot
// sums add the same per-pixel terms in another order (four waves of 64 pixels instead of two of 128).
constexpr int SMALL_THREADS = 256;
#ifndef GS_SMALL_GRID_TILES
#define GS_SMALL_GRID_TILES 3840   // at most this many owned tiles (15 x the 256 CUs): four waves per tile.  Measured, four
                                   // against two waves per tile, backward / forward: 256 tiles -21 % / -20 %, 1,080 (one of eight
                                   // bands of a 1920 x 1072 frame) -3 % of the rank's frame, 2,040 (one of four) -5 % of the frame,
                                   // 2,500 -5 % / -10 %, 3,600 -4 % / -6 %, 4,080 +1 % of the frame, 5,120 +15 % / +1 %, 8,040 +19 % / +4 %
#endif


// FORWARD LIST SPLITTING (round 6).  The launch of a grid that cannot fill the chip lasts as long as the longest tile's
// dependent chain; the forward recursion of a pixel can be CUT at any list position if the transmittance in front of it is
// known -- and that is a product of factors (1 - alpha) none of which depends on the others.  Three launches:
//   probe   (blend_forward_probe_kernel): segment s < split - 1 of every tile walks its share of the list and leaves, per
//           pixel, P_s = the product of (1 - alpha) over its hits (the 1/255 decisions taken exactly as everywhere else),
//           the share of the stop bracket those hits carry and their number -- no colours, no stop rule;
//   blend   (this kernel, split > 1): segment s starts from T_in = P_0 ... P_{s-1} and blends its share with the stop rule
//           applied against the true transmittance.  A pixel the reference stopped in an earlier segment is recognised by
//           T_in itself: the products (1 - alpha) only fall, so "some T' of an earlier segment was below 1e-4" is "T_in is"
//           -- decided by the pixel's bracket as every stop decision is, and by a replay of the pixel's history in the
//           reference's arithmetic inside it (gs_reference_stops_at).  T_in is the reference's transmittance re-associated
//           (s extra roundings: charged to the bracket), so decisions stay the reference's; colours, depth and state are
//           the un-split ones to rounding, not to the bit;
//   combine (blend_forward_combine_kernel): adds the segments' partial results in segment order, writes the outputs, and
//           turns the boundary states of segments > 0 (local colours) into prefix colours for the split backward pass.
// split == 1 is the un-split kernel, bit for bit.
constexpr int FSPLIT_PART_F4 = 3;   // float4 per (segment, pixel): (Cr, Cg, Cb, T_end) (D, Wd, count, last) (Er, Eg, Eb, alive at the start)
__device__ __forceinline__ void gs_segment_batches(int start, int end, int seg, int split, int &k_lo, int &k_hi) {
    const int nb = end > start ? (end - start + BATCH - 1) / BATCH : 0;   // batch k = list positions [start + 128 k, start + 128 (k + 1))
    k_lo = (int)((long long)seg * nb / split);
    k_hi = (int)((long long)(seg + 1) * nb / split);
}

template <bool AUX, bool STATE, bool DEBUG>
__global__ __launch_bounds__(SMALL_THREADS) void blend_forward_small_kernel(
    const int32_t *__restrict__ tile_start, const int32_t *__restrict__ tile_end,
    const int32_t *__restrict__ payload, const float4 *__restrict__ attrs, int width, int height, int row_begin,
    int row_step, float *__restrict__ image, float *__restrict__ depth, float *__restrict__ acc_alpha,
    int32_t *__restrict__ last_effective, int32_t *__restrict__ valid_count, uint32_t *__restrict__ debug_hits,
    const int32_t *__restrict__ tile_order, int32_t *__restrict__ tile_work, float4 *__restrict__ boundary,
    float4 *__restrict__ final_error, int split, const float4 *__restrict__ probe, float4 *__restrict__ parts) {
    __shared__ float4 s_p[BATCH], s_q[BATCH], s_c[BATCH];   // (gs_stage_forward)
    __shared__ float2 s_ro[BATCH];
    __shared__ int s_o[DEBUG ? BATCH : 1];
    __shared__ int s_red[SMALL_THREADS / GS_WAVE];
    const int tw = width / GS_TILE_WIDTH, th = height / GS_TILE_HEIGHT;
    const int seg = split > 1 ? (int)blockIdx.x % split : 0;
    const TileCoord tc = owned_tile_at(split > 1 ? (int)blockIdx.x / split : (int)blockIdx.x,
                                       split > 1 ? (int)gridDim.x / split : (int)gridDim.x, tw, row_begin, row_step, tile_order);
    const int tid = threadIdx.x;
    const int pu = tc.tile_u * GS_TILE_WIDTH + (tid & 15), pv = tc.tile_v * GS_TILE_HEIGHT + (tid >> 4);
    const int start = tile_start[tc.tile_id], list_end = tile_end[tc.tile_id];
    const float px = (float)pu + 0.5f, py = (float)pv + 0.5f;
    const size_t p = (size_t)pv * width + pu, n_pixels = (size_t)width * height;
    int k_lo = 0, k_hi = 0;
    if (split > 1) gs_segment_batches(start, list_end, seg, split, k_lo, k_hi);
    // this workgroup's share of the list: [first, end)
    const int first = split > 1 ? start + BATCH * k_lo : start;
    const int end = split > 1 ? min(start + BATCH * k_hi, list_end) : list_end;
    float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, D = 0.f, Wd = 0.f;
    // Rounding left behind by the colour sums (boundary states only): the image is the plain fp32 sum, as always; next to it
    // E collects what each fma rounded away, so that C + E is the prefix colour to ~1e-14 and the split backward pass can
    // form "colour still to come" = (C_final - C_b) + (E_final - E_b) without the cancellation error of the plain difference
    // (measured without it: slot sums 2e-4 of the column maximum away from the un-split ones)
    const bool track = STATE && boundary != nullptr;
    float Er = 0.f, Eg = 0.f, Eb = 0.f;
    unsigned long long alive_m = ~0ull;   // the lanes whose pixel has not saturated (wave-uniform, as in blend_forward_kernel)
    int last = start, cnt = 0;
    unsigned dh = 0u, dc = 0u;
    float thr = STOP_T;   // upper edge of the pixel's stop bracket (blend_forward_kernel)
    int cnt_in = 0;       // Gaussians blended in front of this segment (the bracket's rounding term counts them)
    bool alive_in = true;
    if (seg > 0) {
        // the state in front of the segment, from the probes of the segments before it (in list order)
        float bracket = 0.f;
#pragma clang loop unroll(disable)
        for (int q = 0; q < seg; ++q) {
            const float4 pr = probe[(size_t)q * n_pixels + p];
            T = T * pr.x;
            bracket += pr.y;
            cnt_in += __builtin_bit_cast(int, pr.z);
        }
        // the bracket as the un-split walk would hold it here (its batches in front of the cut are all full), plus one
        // rounding per segment product (T_in = (P_0 P_1) ... : seg multiplications the reference's chain does not have; the
        // chains' own roundings are the 4 u per blended Gaussian every stop bracket charges) and per partial sum of `bracket`
        const int walked_in = first - start;
        thr = STOP_T + bracket + (STOP_T * 1.1f * GS_STOP_ROUNDING_BAND * (float)walked_in +
                                  (float)(k_lo + 2 * seg) * GS_STOP_THR_SLACK + (float)seg * STOP_T * 1.1f * GS_BAND_SAFETY * GS_U24);
        {   // RAS:458-460 over the segments in front: has the reference stopped this pixel already?  (wave-convergent)
            float lo, hi;
            gs_stop_bracket(thr, walked_in, cnt_in, lo, hi);
            bool dead = cnt_in > 0 && T < lo;
            for (unsigned long long m = gs_ballot(cnt_in > 0 && !dead && T < hi); m != 0ull; m &= m - 1ull) {
                const int l = __builtin_ctzll(m);
                const bool stops = gs_reference_stops_at<false, false>(payload, attrs, start, first - 1, gs_readlane_f(px, l),
                                                                       gs_readlane_f(py, l), tc.tile_u, tc.tile_v, tw, th);
                if ((tid & (GS_WAVE - 1)) == l) dead = stops;
            }
            alive_in = !dead;
            alive_m = gs_ballot(alive_in);
        }
    }
    int pos = first;
    while (pos < end) {
        if (__syncthreads_and(alive_m == 0ull ? 1 : 0)) break;   // barrier (protects the staged batch) + whole-tile early exit
        const int batch_first = pos;
        {
            const int j = pos + tid;
            if (tid < BATCH && j < end) {
                const int o = payload[j];
                const float4 *g = attrs + 4 * (size_t)o;
                float4 P, Q, colour;
                float2 ro;
                gs_stage_forward(g[0], g[1], g[2], g[3], P, Q, colour, ro);
                s_p[tid] = P; s_q[tid] = Q; s_c[tid] = colour; s_ro[tid] = ro;
                if (DEBUG) s_o[tid] = o;
            }
        }
        const int nbuf = min(BATCH, end - pos);
        pos += BATCH;
        {
            const int padded = (nbuf + GROUP - 1) & ~(GROUP - 1);
            if (tid < padded - nbuf) {
                s_p[nbuf + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
                s_q[nbuf + tid] = make_float4(0.f, __builtin_inff(), 0.f, 0.f);   // (e_lo = +inf: never passes the hit test)
            }
        }
        __syncthreads();
        thr += STOP_T * 1.1f * GS_STOP_ROUNDING_BAND * (float)nbuf + GS_STOP_THR_SLACK;
        const int walked = min(pos, end) - start;   // list positions walked so far, this batch included (pos has already moved
                                                    // BATCH on: past `end` in the last, partial batch)
        const int cnt_base = AUX ? cnt_in + 1 : -1; // (+ cnt: the pixel's own number of blended Gaussians, this one included)
        // (the one-pixel forms of blend_forward_kernel's `blend`, `careful_entry` and group loop: see there)
        auto blend = [&](int e, const float4 c, float z, float al, float Tn, bool ok) {
            const float wgt = al * T;
            if (track) {   // (wave-uniform)  exact sum - rounded sum of this step, to first order
                const float nr = __builtin_fmaf(c.x, wgt, Cr), ng = __builtin_fmaf(c.y, wgt, Cg), nb_ = __builtin_fmaf(c.z, wgt, Cb);
                Er += __builtin_fmaf(c.x, wgt, Cr - nr);
                Eg += __builtin_fmaf(c.y, wgt, Cg - ng);
                Eb += __builtin_fmaf(c.z, wgt, Cb - nb_);
            }
            Cr = __builtin_fmaf(c.x, wgt, Cr);
            Cg = __builtin_fmaf(c.y, wgt, Cg);
            Cb = __builtin_fmaf(c.z, wgt, Cb);
            if (AUX) {
                D = __builtin_fmaf(z, wgt, D);
                Wd = Wd + wgt;
                cnt += ok ? 1 : 0;
            }
            T = Tn;
            if (STATE) last = ok ? batch_first + e + 1 : last;
            if (DEBUG) {
                const unsigned hv = (unsigned)(s_o[e] + 1) * GS_HASH_MUL;
                dc += ok ? 1u : 0u; dh += ok ? hv : 0u;
            }
        };
        auto careful_entry = [&](int e) {
            float ex;
            const float4 Q = s_q[e];
            const bool live = __builtin_amdgcn_inverse_ballot_w64(alive_m);
            const float a = live ? gs_pixel_alpha_forward(s_p[e], Q, px, py, ex) : 0.f;
            bool ok = a >= EPS_ALPHA;   // RAS:451
            {
                const bool in = a >= EPS_LO && a < EPS_HI;
                if (gs_ballot(in) != 0ull) {
                    const float2 ro = s_ro[e];
                    const float exact = gs_alpha_reference(ex, ro.x, ro.y);
                    if (in) ok = exact >= EPS_ALPHA;
                }
            }
            if (gs_ballot(ok) == 0ull) return;
            float al = ok ? __builtin_amdgcn_fmed3f(a, 0.f, CLAMP_ALPHA) : 0.f;
            const float4 c = s_c[e];
            thr = __builtin_fmaf(c.w, al, thr);
            float Tn = T * (1.f - al);
            float lo, hi;
            gs_stop_bracket(thr, walked, AUX ? cnt_base + cnt : -1, lo, hi);
            bool sat = ok && Tn < lo;   // RAS:458-460, below the bracket
            for (unsigned long long m = gs_ballot(ok && !sat && Tn < hi); m != 0ull; m &= m - 1ull) {
                const int l = __builtin_ctzll(m);
                const bool stops = gs_reference_stops_at<false, false>(payload, attrs, start, batch_first + e, gs_readlane_f(px, l),
                                                                       gs_readlane_f(py, l), tc.tile_u, tc.tile_v, tw, th);
                if ((tid & (GS_WAVE - 1)) == l) sat = stops;
            }
            if (sat) { al = 0.f; ok = false; }
            alive_m &= ~gs_ballot(sat);
            Tn = T * (1.f - al);
            blend(e, c, Q.w, al, Tn, ok);
        };
        for (int k = 0; k < nbuf; k += GROUP_FWD) {
            if (alive_m == 0ull) break;   // every pixel of this wave is saturated
            float ex[GROUP_FWD], elo[GROUP_FWD];
#pragma unroll
            for (int i = 0; i < GROUP_FWD; ++i) {
                const float4 P = s_p[k + i];
                const float2 q = *reinterpret_cast<const float2 *>(&s_q[k + i]);   // (B, e_lo)
                ex[i] = gs_exponent_forward(px - P.x, py - P.y, P.z, q.x, P.w);
                elo[i] = q.y;
            }
            int careful_from = GROUP_FWD;
#pragma unroll
            for (int i = 0; i < GROUP_FWD; ++i) {
                const unsigned long long mok = gs_ballot(ex[i] >= elo[i]) & alive_m;   // RAS:451 in the exponent's domain
                if (mok == 0ull) continue;
                const float4 c = s_c[k + i];
                const float2 az = reinterpret_cast<const float2 *>(&s_q[k + i])[1];   // (amp, depth)
                const float a = __builtin_amdgcn_exp2f(ex[i] * GS_LOG2E) * az.x;
                if ((mok ^ (gs_ballot(a >= EPS_HI) & mok)) != 0ull) { careful_from = i; break; }
                bool ok = __builtin_amdgcn_inverse_ballot_w64(mok);
                float al = ok ? __builtin_amdgcn_fmed3f(a, 0.f, CLAMP_ALPHA) : 0.f;
                thr = __builtin_fmaf(c.w, al, thr);
                float Tn = T * (1.f - al);
                if ((mok & gs_ballot(Tn < thr)) != 0ull) {              // RAS:458-460: saturates the pixel, NOT blended
                    float lo, hi;
                    gs_stop_bracket(thr, walked, AUX ? cnt_base + cnt : -1, lo, hi);
                    const bool sat = ok && Tn < lo;
                    if (gs_ballot(ok && !sat && Tn < hi) != 0ull) { careful_from = i; break; }   // (thr keeps this entry's share: the
                                                                                                 //  careful twin adds it again -- wider, safe)
                    if (sat) { al = 0.f; ok = false; }
                    alive_m &= ~gs_ballot(sat);
                    Tn = T * (1.f - al);
                }
                blend(k + i, c, az.y, al, Tn, ok);
            }
            if (careful_from < GROUP_FWD) {
#pragma clang loop unroll(disable)
                for (int e = k + careful_from; e < k + GROUP_FWD; ++e) careful_entry(e);
            }
        }
        // Boundary state after every 128 entries of the tile's list (split backward, blend_backward_small_kernel), at slot
        // (list position >> 7): unique, because the tiles' lists are disjoint ranges and two boundaries of one list are 128
        // positions apart
        if (track && pos < list_end) {
            float4 *st = boundary + ((size_t)(pos >> 7) * 256 + tid) * 2;
            st[0] = make_float4(T, Cr, Cg, Cb);
            st[1] = make_float4(Er, Eg, Eb, 0.f);
        }
    }
    if (split > 1) {   // this segment's partial results; blend_forward_combine_kernel adds them up
        float4 *out = parts + ((size_t)seg * n_pixels + p) * FSPLIT_PART_F4;
        out[0] = make_float4(Cr, Cg, Cb, T);
        out[1] = make_float4(D, Wd, __builtin_bit_cast(float, cnt), __builtin_bit_cast(float, last));
        out[2] = make_float4(Er, Eg, Eb, alive_in ? 1.f : 0.f);
        if (DEBUG) {   // (unsigned sums: any order; the caller zeroes the buffer)
            atomicAdd(&debug_hits[2 * p], dc);
            atomicAdd(&debug_hits[2 * p + 1], dh);
        }
        return;
    }
    image[3 * p] = Cr; image[3 * p + 1] = Cg; image[3 * p + 2] = Cb;
    if (track) final_error[p] = make_float4(Er, Eg, Eb, T);   // (.w: the final transmittance itself, see the split backward)
    if (AUX) {
        depth[p] = D / fmaxf(Wd, 1e-6f);  // RAS:479-480
        valid_count[p] = cnt;
    }
    if (STATE) {
        acc_alpha[p] = 1.f - T;
        last_effective[p] = last;
        if (tile_work != nullptr) {
            int mx = last;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, GS_WAVE));
            if ((tid & 63) == 0) s_red[tid >> 6] = mx;
            __syncthreads();
            if (tid == 0) tile_work[tc.index] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])) - start;
        }
    }
    if (DEBUG) { debug_hits[2 * p] = dc; debug_hits[2 * p + 1] = dh; }
}

// The probe of the forward list split (see blend_forward_small_kernel): workgroup (tile, s), s < split - 1, walks the batches
// of segment s and leaves per pixel (P_s, the hits' share of the stop bracket, their number).  Per pixel and entry the
// exponent, the hit test and alpha are the blend kernel's, operation by operation; an alpha short of EPS_HI is settled by the
// reference's expression on the spot (this kernel has registers to spare).  A pixel whose own product has fallen below
// 0.9e-4 is below every stop bracket whatever comes in front of it: it stops looking, and so does its tile when all have.
__global__ __launch_bounds__(SMALL_THREADS) void blend_forward_probe_kernel(
    const int32_t *__restrict__ tile_start, const int32_t *__restrict__ tile_end, const int32_t *__restrict__ payload,
    const float4 *__restrict__ attrs, int width, int height, int row_begin, int row_step,
    const int32_t *__restrict__ tile_order, int split, float4 *__restrict__ probe) {
    __shared__ float4 s_p[BATCH], s_q[BATCH];   // P and Q of gs_stage_forward
    __shared__ float4 s_w[BATCH];               // (stop weight, rescale, opacity, .)
    const int tw = width / GS_TILE_WIDTH;
    const int seg = (int)blockIdx.x % (split - 1);
    const TileCoord tc = owned_tile_at((int)blockIdx.x / (split - 1), (int)gridDim.x / (split - 1), tw, row_begin, row_step,
                                       tile_order);
    const int tid = threadIdx.x;
    const int pu = tc.tile_u * GS_TILE_WIDTH + (tid & 15), pv = tc.tile_v * GS_TILE_HEIGHT + (tid >> 4);
    const int start = tile_start[tc.tile_id], list_end = tile_end[tc.tile_id];
    const float px = (float)pu + 0.5f, py = (float)pv + 0.5f;
    const size_t p = (size_t)pv * width + pu, n_pixels = (size_t)width * height;
    int k_lo, k_hi;
    gs_segment_batches(start, list_end, seg, split, k_lo, k_hi);
    const int end = min(start + BATCH * k_hi, list_end);
    float P = 1.0f, bracket = 0.f;
    int cnt = 0;
    unsigned long long looking = ~0ull;   // the lanes whose pixel is still looking (wave-uniform)
    for (int pos = start + BATCH * k_lo; pos < end; pos += BATCH) {
        if (__syncthreads_and(looking == 0ull ? 1 : 0)) break;   // barrier (protects the staged batch) + whole-tile early exit
        {
            const int j = pos + tid;
            if (tid < BATCH && j < end) {
                const float4 *g = attrs + 4 * (size_t)payload[j];
                float4 Pq, Q, colour;
                float2 ro;
                gs_stage_forward(g[0], g[1], g[2], g[3], Pq, Q, colour, ro);
                s_p[tid] = Pq; s_q[tid] = Q; s_w[tid] = make_float4(colour.w, ro.x, ro.y, 0.f);
            }
        }
        const int nbuf = min(BATCH, end - pos);
        __syncthreads();
        for (int e = 0; e < nbuf; ++e) {
            if (looking == 0ull) break;
            const float4 Pq = s_p[e], Q = s_q[e];
            const float ex = gs_exponent_forward(px - Pq.x, py - Pq.y, Pq.z, Q.x, Pq.w);
            const unsigned long long mok = gs_ballot(ex >= Q.y) & looking;   // RAS:451 in the exponent's domain
            if (mok == 0ull) continue;
            const float4 w = s_w[e];
            const float a = __builtin_amdgcn_exp2f(ex * GS_LOG2E) * Q.z;
            bool ok = __builtin_amdgcn_inverse_ballot_w64(mok);
            const bool unsure = ok && !(a >= EPS_HI);
            if (gs_ballot(unsure) != 0ull) {   // rare: the reference's own expression decides
                const float exact = gs_alpha_reference(ex, w.y, w.z);
                if (unsure) ok = exact >= EPS_ALPHA;
            }
            const float al = ok ? __builtin_amdgcn_fmed3f(a, 0.f, CLAMP_ALPHA) : 0.f;
            P = P * (1.f - al);
            bracket = __builtin_fmaf(w.x, al, bracket);
            cnt += ok ? 1 : 0;
            looking &= ~gs_ballot(P < 0.9f * STOP_T);
        }
    }
    probe[(size_t)seg * n_pixels + p] = make_float4(P, bracket, __builtin_bit_cast(float, cnt), 0.f);
}

// error-free sum: s = fl(a + b), returns a + b - s (contraction is off in this file)
__device__ __forceinline__ float gs_two_sum(float a, float b, float &s) {
    s = a + b;
    const float bb = s - a;
    return (a - (s - bb)) + (b - bb);
}

// The last launch of the forward list split: the segments' partial results added in segment order -> the outputs of
// blend_forward_small_kernel; boundary states of segments > 0 (local colours) -> prefix colours (see there).
template <bool AUX, bool STATE>
__global__ __launch_bounds__(SMALL_THREADS) void blend_forward_combine_kernel(
    const int32_t *__restrict__ tile_start, const int32_t *__restrict__ tile_end, int width, int height, int row_begin,
    int row_step, float *__restrict__ image, float *__restrict__ depth, float *__restrict__ acc_alpha,
    int32_t *__restrict__ last_effective, int32_t *__restrict__ valid_count, const int32_t *__restrict__ tile_order,
    int32_t *__restrict__ tile_work, float4 *__restrict__ boundary, float4 *__restrict__ final_error, int split,
    const float4 *__restrict__ parts) {
    __shared__ int s_red[SMALL_THREADS / GS_WAVE];
    const int tw = width / GS_TILE_WIDTH;
    const TileCoord tc = owned_tile(tw, row_begin, row_step, tile_order);
    const int tid = threadIdx.x;
    const int pu = tc.tile_u * GS_TILE_WIDTH + (tid & 15), pv = tc.tile_v * GS_TILE_HEIGHT + (tid >> 4);
    const size_t p = (size_t)pv * width + pu, n_pixels = (size_t)width * height;
    const int start = tile_start[tc.tile_id], end = tile_end[tc.tile_id];
    const bool track = STATE && boundary != nullptr;
    float C[3] = {0.f, 0.f, 0.f}, E[3] = {0.f, 0.f, 0.f}, D = 0.f, Wd = 0.f, T = 1.0f;
    int cnt = 0, last = start;
#pragma clang loop unroll(disable)
    for (int q = 0; q < split; ++q) {
        const float4 *in = parts + ((size_t)q * n_pixels + p) * FSPLIT_PART_F4;
        const float4 r0 = in[0], r1 = in[1], r2 = in[2];
        if (track && q > 0) {   // segment q's boundary states: local colour + the colour in front of the segment
            int k_lo, k_hi;
            gs_segment_batches(start, end, q, split, k_lo, k_hi);
            for (int kb = k_lo; kb < k_hi; ++kb) {
                const int pos = start + BATCH * (kb + 1);
                if (pos >= end) break;
                float4 *st = boundary + ((size_t)(pos >> 7) * 256 + tid) * 2;
                float4 a = st[0], b = st[1];
                float sum;
                // a pixel the reference stopped in an earlier segment: the state behind its stop is its FINAL transmittance (what
                // the un-split walk keeps, and what a backward segment whose upper cut lies behind the pixel's last blended entry
                // starts it from) -- this segment only knew the product of the probes, stopping factor included
                if (r2.w == 0.f) a.x = T;
                b.x += E[0] + gs_two_sum(C[0], a.y, sum); a.y = sum;
                b.y += E[1] + gs_two_sum(C[1], a.z, sum); a.z = sum;
                b.z += E[2] + gs_two_sum(C[2], a.w, sum); a.w = sum;
                st[0] = a; st[1] = b;
            }
        }
        const float part[3] = {r0.x, r0.y, r0.z}, err[3] = {r2.x, r2.y, r2.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sum;
            E[c] += err[c] + gs_two_sum(C[c], part[c], sum);
            C[c] = sum;
        }
        D += r1.x; Wd += r1.y;
        cnt += __builtin_bit_cast(int, r1.z);
        last = max(last, __builtin_bit_cast(int, r1.w));
        if (r2.w != 0.f) T = r0.w;   // the transmittance the pixel ends with: that of the last segment it was alive in
    }
    image[3 * p] = C[0]; image[3 * p + 1] = C[1]; image[3 * p + 2] = C[2];
    if (track) final_error[p] = make_float4(E[0], E[1], E[2], T);
    if (AUX) {
        depth[p] = D / fmaxf(Wd, 1e-6f);  // RAS:479-480
        valid_count[p] = cnt;
    }
    if (STATE) {
        acc_alpha[p] = 1.f - T;
        last_effective[p] = last;
        if (tile_work != nullptr) {
            int mx = last;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, GS_WAVE));
            if ((tid & 63) == 0) s_red[tid >> 6] = mx;
            __syncthreads();
            if (tid == 0) tile_work[tc.index] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])) - start;
        }
    }
}

// LIST SPLITTING (round 4).  On a grid that cannot fill the chip the launch lasts as long as the longest tile's dependent
// chain (10k Gaussians at 256 x 256: 109 us for 256 workgroups).  The backward recursion of a pixel runs from the end of
// its list to the start, but it can be CUT at any list position b if the state at b is known: the transmittance T_b in
// front of entry b and the colour still to come, S_b = dL/dC . (C_final - C_b).  The forward pass leaves (T_b, C_b) of every
// pixel at every 128th list position (`boundary`, 16 B per pixel and boundary); with them `split` workgroups share a
// tile's list -- workgroup s walks the batches [s nb / split, (s + 1) nb / split) of 128 entries, the last one starts from
// the final state as before.  Every list entry is still walked by exactly one workgroup, so the (Gaussian, tile) slot
// records are written once, by plain stores, and stay bitwise reproducible; a slot sum differs from the un-split one in
// rounding only (T_b as the forward computed it instead of recovered by divisions, S_b by subtraction).  The per-pixel
// |grad uv| image is the sum of the segments' parts: each segment stores its part, the workgroup that arrives LAST at
// the tile's counter adds the parts in segment order (deterministic) and resets the counter.
template <bool DEBUG>
__global__ __launch_bounds__(SMALL_THREADS) void blend_backward_small_kernel(
    const int32_t *__restrict__ tile_start, const int32_t *__restrict__ payload, const float4 *__restrict__ attrs,
    const float *__restrict__ grad_image, const float *__restrict__ acc_alpha, const int32_t *__restrict__ last_effective,
    int width, int height, int row_begin, int row_step, const int32_t *__restrict__ slot_offsets,
    float4 *__restrict__ partials, uint8_t *__restrict__ slot_flags, float *__restrict__ magnitude_image,
    uint32_t *__restrict__ debug_hits, const int32_t *__restrict__ tile_order, int split,
    const float *__restrict__ image, const float4 *__restrict__ boundary, const float4 *__restrict__ final_error,
    int32_t *__restrict__ tile_counters, float2 *__restrict__ magnitude_parts) {
    __shared__ float4 s_p[BATCH], s_q[BATCH], s_c[BATCH];   // (gs_stage_backward)
    __shared__ int s_o[BATCH];
    // one slice of partial sums PER WAVE, written with plain stores and added in a fixed order by the flush: four waves
    // meeting in one row with ds_add_f32 would sum in arrival order (two waves are safe: a + b = b + a) and the gradients
    // would no longer be bitwise reproducible
    __shared__ float s_acc[SMALL_THREADS / GS_WAVE][BATCH][GS_ACC_STRIDE];
    __shared__ int s_max[SMALL_THREADS / GS_WAVE];
    const int tw = width / GS_TILE_WIDTH, th = height / GS_TILE_HEIGHT;
    const int seg = split > 1 ? (int)blockIdx.x % split : 0;
    const TileCoord tc = owned_tile_at(split > 1 ? (int)blockIdx.x / split : (int)blockIdx.x,
                                       split > 1 ? (int)gridDim.x / split : (int)gridDim.x, tw, row_begin, row_step,
                                       tile_order);
    const int tid = threadIdx.x, lane = tid & 63;
    const int pu = tc.tile_u * GS_TILE_WIDTH + (tid & 15), pv = tc.tile_v * GS_TILE_HEIGHT + (tid >> 4);
    const size_t p = (size_t)pv * width + pu;
    const int start = tile_start[tc.tile_id];
    const float px = (float)pu + 0.5f, py = (float)pv + 0.5f;
    const int last = last_effective[p];
    const float Gr = grad_image[3 * p], Gg = grad_image[3 * p + 1], Gb = grad_image[3 * p + 2];
    float mag_u = 0.f, mag_v = 0.f;
    unsigned dh = 0u, dc = 0u;
    int mx = last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, GS_WAVE));
    const int wave_end = mx;
    if (lane == 0) s_max[tid >> 6] = mx;
    __syncthreads();
    const int end = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    // this workgroup's batches of the tile's list (batch k = positions start + 128 k ...), highest first
    const int n_batches = end > start ? (end - start + BATCH - 1) / BATCH : 0;
    const int k_lo = (int)((long long)seg * n_batches / split), k_hi = (int)((long long)(seg + 1) * n_batches / split);
    float T, S;
    if (k_hi == n_batches) {   // the end of the list: the pixel's final state
        T = 1.0f - acc_alpha[p];
        S = 0.f;
    } else {                   // a cut: the state the forward pass left at list position start + 128 k_hi
        const float4 *rec = boundary + ((size_t)((start + BATCH * k_hi) >> 7) * 256 + tid) * 2;
        const float4 st = rec[0], eb = rec[1], ef = final_error[p];
        // The reference starts its recursion from T = 1 - acc_alpha (RAS:560), i.e. from the final transmittance ROUNDED
        // through acc_alpha -- for a nearly saturated pixel (T ~ 1e-3) that is a relative error of ~1e-4, carried by every
        // T and S of the walk as a common factor.  The cut reproduces it: rho = (1 - acc_alpha) / T_final scales the exact
        // state the forward pass left, so that a split walk computes what the un-split one does (to rounding) instead of
        // something more accurate but 1e-4 away from it.
        const float rho = (1.0f - acc_alpha[p]) / ef.w;
        T = st.x * rho;
        // colour still to come behind the cut: (C_final - C_b) + (E_final - E_b), see blend_forward_small_kernel
        const float dr = (image[3 * p] - st.y) + (ef.x - eb.x), dg = (image[3 * p + 1] - st.z) + (ef.y - eb.y),
                    db = (image[3 * p + 2] - st.w) + (ef.z - eb.z);
        S = __builtin_fmaf(db, Gb, __builtin_fmaf(dg, Gg, dr * Gr)) * rho;
    }
    const int row = lane >> 4;
    const int slot = ((row & 1) << 1) | (row >> 1);
    const bool row_tail = (lane & 15) == 15;
    for (int kb = k_hi - 1; kb >= k_lo; --kb) {
        __syncthreads();   // previous round fully flushed before its LDS is reused
        const int bottom = start + BATCH * kb;
        const int batch_first = min(bottom + BATCH, end) - 1;   // highest list position of the batch
        {
            const int j = batch_first - tid;
            if (tid < BATCH && j >= bottom) {
                const int o = payload[j];
                const float4 *g = attrs + 4 * (size_t)o;
                float4 P, Q, colour;
                gs_stage_backward(g[0], g[1], g[2], g[3], P, Q, colour);
                s_p[tid] = P; s_q[tid] = Q; s_c[tid] = colour;
                s_o[tid] = o;
            }
        }
        const int nbuf = batch_first - bottom + 1;
        {
            const int padded = (nbuf + GROUP - 1) & ~(GROUP - 1);
            if (tid < padded - nbuf) {
                s_p[nbuf + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
                s_q[nbuf + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {   // thread t clears rows t & 127 of slices 2 (t >> 7) and 2 (t >> 7) + 1
                const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 *z = reinterpret_cast<float4 *>(&s_acc[2 * (tid >> 7) + h][tid & (BATCH - 1)][0]);
                    z[0] = zero; z[1] = zero; z[2] = zero;
                }
            }
        }
        __syncthreads();
        for (int k = 0; k < nbuf; k += GROUP_BWD) {
            if (batch_first - (k + GROUP_BWD - 1) >= wave_end) continue;   // the whole group lies behind this wave's pixels
            float alpha[GROUP_BWD], m0[GROUP_BWD], m1[GROUP_BWD];
#pragma unroll
            for (int i = 0; i < GROUP_BWD; ++i) {
                float ex;
                alpha[i] = gs_pixel_alpha_backward(s_p[k + i], s_q[k + i], px, py, ex, m0[i], m1[i]);
            }
            // (A) the group's 1/255 decisions, (then) the bracketed ones settled by the reference's backward expression, (B) the
            // hit path: the one-pixel form of blend_backward_kernel's loop, see there
            unsigned long long ma[GROUP_BWD];   // (lane masks, as in blend_backward_kernel)
            unsigned bracketed = 0u;
#pragma unroll
            for (int i = 0; i < GROUP_BWD; ++i) {
                ma[i] = gs_ballot(alpha[i] >= EPS_LO);                   // RAS:631 (lower edge of the bracket)
                if (ma[i] != 0ull && (ma[i] ^ gs_ballot(alpha[i] >= EPS_HI)) != 0ull) bracketed |= 1u << i;
            }
            if (bracketed != 0u) {
#pragma clang loop unroll(disable)
                for (int ii = 0; ii < GROUP_BWD; ++ii) {
                    if (((bracketed >> ii) & 1u) == 0u) continue;
                    const int e = k + ii;
                    float ex, mm0, mm1;
                    const float4 Q = s_q[e];
                    const float al = gs_pixel_alpha_backward(s_p[e], Q, px, py, ex, mm0, mm1);
                    bool r0 = al >= EPS_ALPHA;
                    const bool in = al >= EPS_LO && al < EPS_HI;
                    if (gs_ballot(in) != 0ull) {
                        const float exact = gs_alpha_reference(ex, reinterpret_cast<const float *>(attrs + 4 * (size_t)s_o[e])[15], s_c[e].w);
                        if (in) r0 = exact >= EPS_ALPHA;
                    }
                    const unsigned long long mr = gs_ballot(r0);
#pragma unroll
                    for (int i = 0; i < GROUP_BWD; ++i)
                        if (i == ii) ma[i] = mr;
                }
            }
#pragma unroll
            for (int i = 0; i < GROUP_BWD; ++i) {
                if (ma[i] == 0ull) continue;
                const int jj = batch_first - (k + i);
                const unsigned long long mh = ma[i] & gs_ballot(jj < last);   // RAS:618 (effective range)
                if (mh == 0ull) continue;
                const bool hit = __builtin_amdgcn_inverse_ballot_w64(mh);
                const float h = hit ? 1.f : 0.f;
                const float al = hit ? __builtin_amdgcn_fmed3f(alpha[i], 0.f, CLAMP_ALPHA) : 0.f;
                const float inv1m = __builtin_amdgcn_rcpf(1.f - al);
                T = T * inv1m;                                           // RAS:643
                const float aT = al * T;
                const float4 c = s_c[k + i];
                const float gr = aT * Gr, gg = aT * Gg, gb = aT * Gb;
                const float cg = __builtin_fmaf(c.z, Gb, __builtin_fmaf(c.y, Gg, c.x * Gr));
                const float dLda = __builtin_fmaf(T, cg, -(S * inv1m)) * h;   // RAS:652-657
                S = __builtin_fmaf(cg, aT, S);
                const float w = dLda * alpha[i];
                const float v0 = w * m0[i], v1 = w * m1[i];              // UTL:331-348: m = conic @ d (from the evaluation of alpha)
                mag_u += fabsf(v0);
                mag_v += fabsf(v1);
                const float c00 = v0 * m0[i], c01 = v0 * m1[i], c11 = v1 * m1[i];
                const float nv = __builtin_amdgcn_sqrtf(__builtin_fmaf(v1, v1, v0 * v0));
                if (DEBUG) {
                    const unsigned hv = (unsigned)(s_o[k + i] + 1) * GS_HASH_MUL;
                    dc += hit ? 1u : 0u; dh += hit ? hv : 0u;
                }
                float t0, t1, t2;
                gs_wave_reduce12(v0, v1, c00, c01, c11, gr, gg, gb, w, nv, h, 0.f, t0, t1, t2);
                if (row_tail) {   // (every (wave, entry) row is written at most once per round)
                    float *A = &s_acc[tid >> 6][k + i][slot];
                    A[0] = t0; A[4] = t1; A[8] = t2;
                }
            }
        }
        __syncthreads();
        if (tid < nbuf) {   // flush: thread k owns staged entry k -> one 48-B store into the (Gaussian, tile) slot
            float4 r[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {   // (wave 0 + wave 1) + (wave 2 + wave 3)
                const float4 a0 = reinterpret_cast<const float4 *>(&s_acc[0][tid][0])[c];
                const float4 a1 = reinterpret_cast<const float4 *>(&s_acc[1][tid][0])[c];
                const float4 a2 = reinterpret_cast<const float4 *>(&s_acc[2][tid][0])[c];
                const float4 a3 = reinterpret_cast<const float4 *>(&s_acc[3][tid][0])[c];
                r[c] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                   (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
            }
            float4 r0 = r[0], r1 = r[1], r2 = r[2];
            if (r2.z > 0.f) {
                const float4 a = s_p[tid];
                int t0u, t1u, t0v, t1v;
                gs_tile_box(a.x, a.y, s_q[tid].z, tw, th, t0u, t1u, t0v, t1v);
                const int dst_slot = slot_offsets[s_o[tid]] + (t1v - t0v) * (tc.tile_u - t0u) + (tc.tile_v - t0v);
                float4 *dst = partials + 3 * (size_t)dst_slot;
                r0.z *= 0.5f; r0.w *= 0.5f; r1.x *= 0.5f;
                r2.x *= (1.f - s_c[tid].w);
                r2.z = __builtin_bit_cast(float, (int)r2.z);
                dst[0] = r0;
                dst[1] = r1;
                dst[2] = r2;
                slot_flags[dst_slot] = 1;
            }
        }
    }
    if (split <= 1) {
        magnitude_image[2 * p] = mag_u;
        magnitude_image[2 * p + 1] = mag_v;
        if (DEBUG) { debug_hits[2 * p] = dc; debug_hits[2 * p + 1] = dh; }
        return;
    }
    // several workgroups per tile: this segment's part of the |grad uv| image; the last workgroup to arrive adds the parts.
    // The parts are stored WRITE-THROUGH (agent-scope atomic stores = sc1) and read back with agent-scope loads, every
    // storing wave drains its stores before the workgroup's one arrival atomic: no release / acquire fence (a release
    // fence writes back the XCD's whole L2: measured, one per workgroup made the split kernel slower than the un-split one)
    const size_t n_pixels = (size_t)width * height;
    typedef unsigned long long u64;
    u64 *parts = reinterpret_cast<u64 *>(magnitude_parts);
    __hip_atomic_store(&parts[(size_t)seg * n_pixels + p],
                       (u64)__builtin_bit_cast(unsigned, mag_u) | ((u64)__builtin_bit_cast(unsigned, mag_v) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (DEBUG) {   // (unsigned sums: any order; the caller zeroes the buffer)
        atomicAdd(&debug_hits[2 * p], dc);
        atomicAdd(&debug_hits[2 * p + 1], dh);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains (Guideline 16, R1)
    __syncthreads();
    if (tid == 0) {
        const int arrived = __hip_atomic_fetch_add(&tile_counters[tc.index], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_max[0] = arrived == split - 1 ? 1 : 0;
        if (arrived == split - 1)   // clean for the next launch
            __hip_atomic_store(&tile_counters[tc.index], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_max[0]) {
        // (the last workgroup to arrive: an acquire fence at agent scope -- only here, once per tile -- orders the loads below
        // behind the arrival it has just observed; the other segments' stores were written through and drained before
        // their own arrival)
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        float su = 0.f, sv = 0.f;
        for (int q = 0; q < split; ++q) {   // segment order: the same sum on every run
            const u64 m = __hip_atomic_load(&parts[(size_t)q * n_pixels + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            su += __builtin_bit_cast(float, (unsigned)m);
            sv += __builtin_bit_cast(float, (unsigned)(m >> 32));
        }
        magnitude_image[2 * p] = su;
        magnitude_image[2 * p + 1] = sv;
    }
}

// Sums the flagged (Gaussian, tile) slots of every visible Gaussian into the accumulator record acc[i] (gs_slots.h: the
// code is shared with the fused per-point backward, which keeps the sums in registers; this kernel serves the callers
// that need acc in memory -- a multi-GPU run all-reduces it, tests compare it).
#ifndef GS_RP_MIN_BLOCKS
#define GS_RP_MIN_BLOCKS 6   // workgroups per CU = waves per SIMD: the kernel lives on memory-level parallelism (5: 63 -> 69 us
                             // at the headline size); 77 registers with one record in flight in the whole-wave path
#endif
// LANES = 1: one lane per Gaussian (+ whole-wave help for the rare heavy one); LANES = 16: sixteen lanes (one DPP
// row) per Gaussian, lane l taking the slot groups l, l + 16, ... -- chosen by the host when Gaussians own many slots
// on average (a wave full of heavy Gaussians would otherwise serialise them).
#ifndef GS_RP_CHUNK
#define GS_RP_CHUNK 2        // 48-B records in flight per lane (gs_slots.h)
#endif
template <int LANES>
__global__ __launch_bounds__(GS_BLOCK, GS_RP_MIN_BLOCKS) void reduce_partials_kernel(
    const int32_t *__restrict__ slot_offsets, const int32_t *__restrict__ ntiles_full,
    const uint8_t *__restrict__ slot_flags, const float4 *__restrict__ partials, int m, float4 *__restrict__ acc,
    const int32_t *__restrict__ nkeys, const float4 *__restrict__ attrs, int tw, int th) {
    const int i = (int)(((long long)blockIdx.x * GS_BLOCK + threadIdx.x) / LANES);
    const int sub = threadIdx.x & (LANES - 1);
    const bool live = i < m;
    SlotSum a;
    if (LANES > 1) {
        const int base = live ? slot_offsets[i] : 0, n = live && (nkeys == nullptr || nkeys[i] > 0) ? ntiles_full[i] : 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) a.v[k] = 0.f;
        a.npix = 0;
        for (int r0 = 4 * sub; r0 < n; r0 += 4 * LANES) rp_add_group<4, 4>(slot_flags, partials, base + r0, min(4, n - r0), a);
        // the LANES lanes of a Gaussian are one DPP row: totals land in lane 15 of the row
#pragma unroll
        for (int k = 0; k < 10; ++k) a.v[k] = gs_row_sum_to_lane15(a.v[k]);
        a.npix = (int)gs_row_sum_to_lane15((float)a.npix);   // < 2^24: exact as a float
    } else {
        gs_sum_slots_of_lane<GS_RP_CHUNK, 1>(live, i, slot_offsets, ntiles_full, slot_flags, partials, nkeys, attrs, tw, th, a);
    }
    if (live && sub == LANES - 1) {
        acc[3 * (size_t)i] = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
        acc[3 * (size_t)i + 1] = make_float4(a.v[4], a.v[5], a.v[6], a.v[7]);
        acc[3 * (size_t)i + 2] = make_float4(a.v[8], a.v[9], __builtin_bit_cast(float, a.npix), 0.f);
    }
}

template <bool STAGED, bool AUX, bool STATE>
static void launch_forward(bool debug, dim3 grid, hipStream_t s, const int32_t *bin_start, const int32_t *bin_end,
                           const int32_t *payload, const float4 *attrs, int width, int height, int rb, int rs,
                           int bin_shift, int filter, float *image, float *depth, float *acc_alpha,
                           int32_t *last_effective, int32_t *valid_count, uint32_t *debug_hits,
                           const int32_t *tile_order, int32_t *tile_work, int32_t *walked_list, int32_t *walked_start) {
    if (debug)
        hipLaunchKernelGGL((blend_forward_kernel<STAGED, AUX, STATE, true>), grid, dim3(BLEND_THREADS), 0, s, bin_start,
                           bin_end, payload, attrs, width, height, rb, rs, bin_shift, filter, image, depth, acc_alpha,
                           last_effective, valid_count, debug_hits, tile_order, tile_work, walked_list, walked_start);
    else
        hipLaunchKernelGGL((blend_forward_kernel<STAGED, AUX, STATE, false>), grid, dim3(BLEND_THREADS), 0, s, bin_start,
                           bin_end, payload, attrs, width, height, rb, rs, bin_shift, filter, image, depth, acc_alpha,
                           last_effective, valid_count, debug_hits, tile_order, tile_work, walked_list, walked_start);
}

template <bool STAGED>
static void launch_backward(bool debug, dim3 grid, hipStream_t s, const int32_t *bin_start, const int32_t *payload,
                            const float4 *attrs, const float *grad_image, const float *acc_alpha,
                            const int32_t *last_effective, int width, int height, int rb, int rs, int bin_shift,
                            int filter, const int32_t *slot_offsets, float4 *partials, uint8_t *slot_flags,
                            float *magnitude_image, uint32_t *debug_hits, const int32_t *tile_order, bool skewed) {
    constexpr int ARM = STAGED ? GS_BWD_REDUCE_STAGED : GS_BWD_REDUCE_DIRECT;
#define GS_BWD_LAUNCH(DBG, REDUCE)                                                                                   \
    hipLaunchKernelGGL((blend_backward_kernel<STAGED, DBG, REDUCE>), grid, dim3(BLEND_THREADS), 0, s, bin_start,       \
                       payload, attrs, grad_image, acc_alpha, last_effective, width, height, rb, rs, bin_shift, filter, \
                       slot_offsets, partials, slot_flags, magnitude_image, debug_hits, tile_order)
    if (skewed && GS_BWD_REDUCE_SKEWED != ARM) {   // (GS_BLEND_SKEWED_WALKS: see GS_BWD_REDUCE_SKEWED)
        if (debug) GS_BWD_LAUNCH(true, GS_BWD_REDUCE_SKEWED); else GS_BWD_LAUNCH(false, GS_BWD_REDUCE_SKEWED);
    } else {
        if (debug) GS_BWD_LAUNCH(true, ARM); else GS_BWD_LAUNCH(false, ARM);
    }
#undef GS_BWD_LAUNCH
}

}  // namespace

extern "C" {

static int owned_row_count(int th, int begin, int step, int end) {
    const int hi = end < th ? end : th;
    return begin < hi ? (hi - 1 - begin) / step + 1 : 0;
}

// workgroups per tile of the split backward pass for a grid of `tiles` tiles (1 = no split)
static int backward_split_for(int tiles) {
    static const int forced = getenv("GS_BWD_SPLIT") ? atoi(getenv("GS_BWD_SPLIT")) : 0;   // tuning knob
    if (forced > 0) return forced > GS_MAX_BACKWARD_SPLIT ? GS_MAX_BACKWARD_SPLIT : forced;
    // (a grid of ~1000 tiles x four waves fills the chip: beyond that the extra workgroups only cost -- measured: 2,500
    //  tiles split in two 0.33 -> 0.45 ms per cfg-2 step, a 1,080-tile band 0.135 -> 0.29 ms)
    return tiles <= 512 ? 4 : (tiles <= 1024 ? 2 : 1);
}

// [ (list_length >> 7) + 2 slots x 256 pixels x 2 float4 ][ width x height float4: what the image's sums rounded away ]
static size_t boundary_slots_bytes(int64_t list_length) {
    return ((size_t)((list_length > 0 ? list_length : 0) >> 7) + 2) * 256 * 2 * sizeof(float4);
}
size_t gs_blend_boundary_bytes(int64_t list_length, int width, int height) {
    return boundary_slots_bytes(list_length) + (size_t)width * height * sizeof(float4);
}

int gs_blend_read_stats(uint64_t *counters, int clear, void *stream) {
    GS_REQUIRE(counters != nullptr, "counters");
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_HIP(hipStreamSynchronize(s));
#if GS_STATS
    GS_CHECK_HIP(hipMemcpyFromSymbol(counters, HIP_SYMBOL(gs_blend_stats_dev), sizeof(uint64_t) * GS_BLEND_STATS));
    if (clear) {
        const uint64_t zero[GS_BLEND_STATS] = {};
        GS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(gs_blend_stats_dev), zero, sizeof(zero)));
    }
#else   // the product build has no counters
    (void)clear;
    for (int i = 0; i < GS_BLEND_STATS; ++i) counters[i] = 0;
#endif
    return GS_STATS;
}

size_t gs_blend_split_workspace_bytes(int width, int height) {
    const size_t tiles = (size_t)(width / GS_TILE_WIDTH) * (height / GS_TILE_HEIGHT);
    return ((tiles * sizeof(int32_t) + 255) & ~(size_t)255) + (size_t)GS_MAX_BACKWARD_SPLIT * width * height * sizeof(float2);
}

int gs_blend_forward(const int32_t *bin_start, const int32_t *bin_end, const int32_t *payload, const float *attrs,
                     int width, int height, int tile_row_begin, int tile_row_step, int tile_row_end, int bin_shift,
                     int filter, float *image, float *depth, float *acc_alpha, int32_t *last_effective,
                     int32_t *valid_count, int flags, uint32_t *debug_pixel_hits, int32_t *tile_order,
                     int32_t *tile_work, int32_t *walked_list, int32_t *walked_start, void *stream) {
    return gs_blend_forward_split(bin_start, bin_end, payload, attrs, width, height, tile_row_begin, tile_row_step,
                                  tile_row_end, bin_shift, filter, image, depth, acc_alpha, last_effective, valid_count, flags,
                                  debug_pixel_hits, tile_order, tile_work, walked_list, walked_start, nullptr, 0, nullptr,
                                  stream);
}

int gs_blend_forward_with_boundaries(const int32_t *bin_start, const int32_t *bin_end, const int32_t *payload,
                                     const float *attrs, int width, int height, int tile_row_begin, int tile_row_step,
                                     int tile_row_end, int bin_shift, int filter, float *image, float *depth,
                                     float *acc_alpha, int32_t *last_effective, int32_t *valid_count, int flags,
                                     uint32_t *debug_pixel_hits, int32_t *tile_order, int32_t *tile_work,
                                     int32_t *walked_list, int32_t *walked_start, float *boundary_states,
                                     int64_t list_length, void *stream) {
    return gs_blend_forward_split(bin_start, bin_end, payload, attrs, width, height, tile_row_begin, tile_row_step,
                                  tile_row_end, bin_shift, filter, image, depth, acc_alpha, last_effective, valid_count, flags,
                                  debug_pixel_hits, tile_order, tile_work, walked_list, walked_start, boundary_states,
                                  list_length, nullptr, stream);
}

size_t gs_blend_forward_split_workspace_bytes(int width, int height) {
    const size_t pixels = (size_t)width * height;
    return pixels * sizeof(float4) * ((GS_MAX_FORWARD_SPLIT - 1) + (size_t)GS_MAX_FORWARD_SPLIT * FSPLIT_PART_F4);
}

// workgroups per tile of the split forward pass for a grid of `tiles` tiles (1 = no split)
static int forward_split_for(int tiles) {
    static const int forced = getenv("GS_FWD_SPLIT") ? atoi(getenv("GS_FWD_SPLIT")) : 0;   // tuning knob
    if (forced > 0) return forced > GS_MAX_FORWARD_SPLIT ? GS_MAX_FORWARD_SPLIT : forced;
    // A tile's chain becomes (probe + blend) / split long, plus two launches: two workgroups per tile buy nothing, four halve
    // it -- and pay only while the un-split launch is far from filling the chip.  Measured (tools/small_frame_bench.py,
    // operator forward + backward back to back, un-split -> four workgroups): 256 tiles 0.29-0.32 -> 0.21 ms, 400 tiles
    // 0.245-0.27 -> 0.27, 576 tiles 0.25-0.28 -> 0.29, 1,024 tiles 0.29 -> 0.37-0.42 (two workgroups: slower at every size)
    return tiles <= GS_FORWARD_SPLIT_TILES ? GS_MAX_FORWARD_SPLIT : 1;
}

int gs_blend_forward_split(const int32_t *bin_start, const int32_t *bin_end, const int32_t *payload,
                           const float *attrs, int width, int height, int tile_row_begin, int tile_row_step,
                           int tile_row_end, int bin_shift, int filter, float *image, float *depth,
                           float *acc_alpha, int32_t *last_effective, int32_t *valid_count, int flags,
                           uint32_t *debug_pixel_hits, int32_t *tile_order, int32_t *tile_work,
                           int32_t *walked_list, int32_t *walked_start, float *boundary_states,
                           int64_t list_length, void *forward_split_workspace, void *stream) {
    // (boundary states are only produced by the four-waves-per-tile kernel on per-tile lists: what small frames use)
    float4 *boundary = reinterpret_cast<float4 *>(boundary_states);
    float4 *final_error = boundary_states == nullptr ? nullptr
                                                     : reinterpret_cast<float4 *>((char *)boundary_states + boundary_slots_bytes(list_length));
    GS_REQUIRE(width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0, "image size must be a multiple of 16");
    GS_REQUIRE(tile_row_step >= 1 && tile_row_begin >= 0 && tile_row_end >= 0, "tile row ownership");
    GS_REQUIRE(bin_shift >= 0 && bin_shift <= 4, "bin_shift");
    GS_REQUIRE(bin_shift == 0 || (filter & GS_FILTER_BOX), "lists that cover several tiles need the box filter");
    const bool staged = bin_shift > 0 || filter != 0;   // per-tile lists taken as they are go the direct way
    const bool aux = !(flags & GS_BLEND_RGB_ONLY), state = !(flags & GS_BLEND_NO_STATE);
    GS_REQUIRE(image != nullptr, "image");
    GS_REQUIRE(!aux || (depth != nullptr && valid_count != nullptr), "depth / valid_count (or pass GS_BLEND_RGB_ONLY)");
    GS_REQUIRE(!state || (acc_alpha != nullptr && last_effective != nullptr),
               "acc_alpha / last_effective (or pass GS_BLEND_NO_STATE)");
    const int tw = width / GS_TILE_WIDTH;
    const int rows = owned_row_count(height / GS_TILE_HEIGHT, tile_row_begin, tile_row_step, tile_row_end);
    if (rows == 0 || tw == 0) return 0;
    const dim3 grid(tw * rows);
    const float4 *a4 = reinterpret_cast<const float4 *>(attrs);
    hipStream_t s = (hipStream_t)stream;
    const bool dbg = debug_pixel_hits != nullptr;
    GS_REQUIRE(tile_work == nullptr || state, "tile_work is the backward's walk length: it needs the state outputs");
    GS_REQUIRE((walked_list == nullptr) == (walked_start == nullptr), "walked_list and walked_start go together");
    GS_REQUIRE(walked_list == nullptr || (staged && state), "walked lists are emitted by the filtering (binned) forward with state");
    const bool four_waves = !staged && !(flags & GS_BLEND_TWO_WAVES) &&
                            ((flags & (GS_BLEND_FOUR_WAVES | GS_BLEND_SPLIT_FORWARD)) || tw * rows <= GS_SMALL_GRID_TILES);
    // several workgroups per tile on a grid that cannot fill the chip (see blend_forward_small_kernel): probe, blend, combine
    const int split = !(four_waves && forward_split_workspace != nullptr) ? 1
                      : (flags & GS_BLEND_SPLIT_FORWARD) ? GS_MAX_FORWARD_SPLIT : forward_split_for(tw * rows);
    // A launch whose workgroups are all resident at once (four-wave kernels: at least four per CU) has no dispatch order to
    // speak of: the ordering launch (5 us, a quarter of what is left of a 256-tile frame's forward pass) is left out
    if (four_waves && tw * rows * split <= 1024) tile_order = nullptr;
    if (tile_order != nullptr) {   // longest lists first
        hipLaunchKernelGGL(tile_order_kernel, dim3(ORDER_WGS), dim3(ORDER_THREADS), 0, s, (const int32_t *)nullptr, bin_start,
                           bin_end, tw * rows, tw, tile_row_begin, tile_row_step, bin_shift, tile_order, (uint4 *)nullptr, 0LL);
        GS_CHECK_LAUNCH();
    }
    const size_t n_pixels = (size_t)width * height;
    float4 *probe = reinterpret_cast<float4 *>(forward_split_workspace);
    float4 *parts = probe == nullptr ? nullptr : probe + (GS_MAX_FORWARD_SPLIT - 1) * n_pixels;
    if (split > 1) {
        hipLaunchKernelGGL(blend_forward_probe_kernel, dim3(tw * rows * (split - 1)), dim3(SMALL_THREADS), 0, s, bin_start,
                           bin_end, payload, a4, width, height, tile_row_begin, tile_row_step, tile_order, split, probe);
        GS_CHECK_LAUNCH();
    }
    const dim3 sgrid(tw * rows * split);
#define GS_FWD_SMALL(AUX, STATE, DBG)                                                                                \
    hipLaunchKernelGGL((blend_forward_small_kernel<AUX, STATE, DBG>), sgrid, dim3(SMALL_THREADS), 0, s, bin_start,    \
                       bin_end, payload, a4, width, height, tile_row_begin, tile_row_step, image, depth, acc_alpha,   \
                       last_effective, valid_count, debug_pixel_hits, tile_order, tile_work, boundary, final_error,   \
                       split, probe, parts)
#define GS_FWD_COMBINE(AUX, STATE)                                                                                   \
    hipLaunchKernelGGL((blend_forward_combine_kernel<AUX, STATE>), grid, dim3(SMALL_THREADS), 0, s, bin_start, bin_end, \
                       width, height, tile_row_begin, tile_row_step, image, depth, acc_alpha, last_effective,         \
                       valid_count, tile_order, tile_work, boundary, final_error, split, parts)
#define GS_FWD_SMALL2(AUX, STATE) do { if (dbg) GS_FWD_SMALL(AUX, STATE, true); else GS_FWD_SMALL(AUX, STATE, false); } while (0)
#define GS_FWD(STAGED, AUX, STATE)                                                                                  \
    launch_forward<STAGED, AUX, STATE>(dbg, grid, s, bin_start, bin_end, payload, a4, width, height, tile_row_begin, \
                                       tile_row_step, bin_shift, filter, image, depth, acc_alpha, last_effective,   \
                                       valid_count, debug_pixel_hits, tile_order, tile_work, walked_list, walked_start)
#define GS_FWD2(STAGED)                                                                                             \
    do {                                                                                                            \
        if (aux && state) GS_FWD(STAGED, true, true);                                                               \
        else if (aux) GS_FWD(STAGED, true, false);                                                                  \
        else if (state) GS_FWD(STAGED, false, true);                                                                \
        else GS_FWD(STAGED, false, false);                                                                          \
    } while (0)
    if (four_waves) {
        if (aux && state) GS_FWD_SMALL2(true, true);
        else if (aux) GS_FWD_SMALL2(true, false);
        else if (state) GS_FWD_SMALL2(false, true);
        else GS_FWD_SMALL2(false, false);
        if (split > 1) {
            GS_CHECK_LAUNCH();
            if (aux && state) GS_FWD_COMBINE(true, true);
            else if (aux) GS_FWD_COMBINE(true, false);
            else if (state) GS_FWD_COMBINE(false, true);
            else GS_FWD_COMBINE(false, false);
        }
    } else if (staged) GS_FWD2(true); else GS_FWD2(false);
#undef GS_FWD_COMBINE
#undef GS_FWD_SMALL2
#undef GS_FWD_SMALL
#undef GS_FWD2
#undef GS_FWD
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_blend_backward(const int32_t *bin_start, const int32_t *payload, const float *attrs,
                      const float *grad_image, const float *acc_alpha, const int32_t *last_effective,
                      const int32_t *slot_offsets, int64_t n_slots, int width, int height, int tile_row_begin,
                      int tile_row_step, int tile_row_end, int bin_shift, int filter, float *partials,
                      uint8_t *slot_flags, float *magnitude_image, uint32_t *debug_pixel_hits, int flags,
                      const int32_t *tile_work, int32_t *tile_order, void *stream) {
    return gs_blend_backward_split(bin_start, payload, attrs, grad_image, acc_alpha, last_effective, slot_offsets, n_slots,
                                   width, height, tile_row_begin, tile_row_step, tile_row_end, bin_shift, filter, partials,
                                   slot_flags, magnitude_image, debug_pixel_hits, flags, tile_work, tile_order, nullptr,
                                   nullptr, 0, nullptr, stream);
}

int gs_blend_backward_split(const int32_t *bin_start, const int32_t *payload, const float *attrs,
                            const float *grad_image, const float *acc_alpha, const int32_t *last_effective,
                            const int32_t *slot_offsets, int64_t n_slots, int width, int height, int tile_row_begin,
                            int tile_row_step, int tile_row_end, int bin_shift, int filter, float *partials,
                            uint8_t *slot_flags, float *magnitude_image, uint32_t *debug_pixel_hits, int flags,
                            const int32_t *tile_work, int32_t *tile_order, const float *image,
                            const float *boundary_states, int64_t list_length, void *split_workspace, void *stream) {
    GS_REQUIRE(width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0, "image size must be a multiple of 16");
    GS_REQUIRE(tile_row_step >= 1 && tile_row_begin >= 0 && tile_row_end >= 0, "tile row ownership");
    GS_REQUIRE(bin_shift >= 0 && bin_shift <= 4, "bin_shift");
    GS_REQUIRE(bin_shift == 0 || (filter & GS_FILTER_BOX), "lists that cover several tiles need the box filter");
    GS_REQUIRE(n_slots >= 0, "n_slots");
    hipStream_t s = (hipStream_t)stream;
    // the flag buffer is padded to a multiple of 16 bytes (header): one aligned fill instead of an aligned fill plus a
    // second launch for the odd tail -- and no fill launch at all when the dispatch order is computed below: the order
    // kernel's launch carries the workgroups that clear the flags (tile_order_kernel)
    const size_t flag_bytes = ((size_t)n_slots + 15) & ~(size_t)15;
    const int tw = width / GS_TILE_WIDTH;
    const int rows = owned_row_count(height / GS_TILE_HEIGHT, tile_row_begin, tile_row_step, tile_row_end);
    const bool order_clears_flags = tile_work != nullptr && rows > 0 && tw > 0 &&
                                    (reinterpret_cast<uintptr_t>(slot_flags) & 15u) == 0;
    if (n_slots > 0 && !order_clears_flags) GS_CHECK_HIP(hipMemsetAsync(slot_flags, 0, flag_bytes, s));
    if (rows == 0 || tw == 0) return 0;
    const dim3 grid(tw * rows);
    const float4 *a4 = reinterpret_cast<const float4 *>(attrs);
    float4 *p4 = reinterpret_cast<float4 *>(partials);
    const bool staged = bin_shift > 0 || filter != 0;
    GS_REQUIRE(tile_work == nullptr || tile_order != nullptr, "tile_work needs the tile_order buffer");
    if (tile_work != nullptr) {   // longest walks first, from the lengths the forward pass recorded
        const int zero_wgs = n_slots > 0 && order_clears_flags ? gs_div_up((long long)flag_bytes, ORDER_ZERO_BYTES_PER_WG) : 0;
        hipLaunchKernelGGL(tile_order_kernel, dim3(ORDER_WGS + zero_wgs), dim3(ORDER_THREADS), 0, s, tile_work,
                           (const int32_t *)nullptr, (const int32_t *)nullptr, tw * rows, tw, tile_row_begin, tile_row_step,
                           bin_shift, tile_order, reinterpret_cast<uint4 *>(slot_flags), zero_wgs ? (long long)flag_bytes : 0LL);
        GS_CHECK_LAUNCH();
    }
    const bool four_waves = !staged && !(flags & (GS_BLEND_TWO_WAVES | GS_BLEND_ONE_WAVE)) &&
                            ((flags & GS_BLEND_FOUR_WAVES) || tw * rows <= GS_SMALL_GRID_TILES);
    // per-tile lists on a grid that fills the chip: one wave per tile, four pixels per lane (blend_backward_wide_kernel)
    // (GS_BWD_ONE_WAVE=1: the library's own choice on large grids, A/B knob.  Measured at the headline size, same box: two
    // waves per tile with the LDS reduction 0.400 ms, one wave per tile 0.408, two waves with the register reduce-scatter
    // 0.423: profiles/r06_backward_arms.md)
    static const bool wide_default = getenv("GS_BWD_ONE_WAVE") && atoi(getenv("GS_BWD_ONE_WAVE")) == 1;
    const bool one_wave = !staged && !four_waves && ((flags & GS_BLEND_ONE_WAVE) || (!(flags & GS_BLEND_TWO_WAVES) && wide_default));
    if (four_waves) {
        // several workgroups per tile when the forward pass left its boundary states (see blend_backward_small_kernel)
        const bool can_split = image != nullptr && boundary_states != nullptr && split_workspace != nullptr;
        const int split = can_split ? backward_split_for(tw * rows) : 1;
        int32_t *counters = (int32_t *)split_workspace;
        float2 *parts = reinterpret_cast<float2 *>((char *)split_workspace +
                                                   (((size_t)(tw * (height / GS_TILE_HEIGHT)) * sizeof(int32_t) + 255) & ~(size_t)255));
        const dim3 sgrid(tw * rows * split);
        const float4 *b4 = reinterpret_cast<const float4 *>(boundary_states);
        const float4 *e4 = boundary_states == nullptr ? nullptr
                                                      : reinterpret_cast<const float4 *>((const char *)boundary_states + boundary_slots_bytes(list_length));
        if (debug_pixel_hits != nullptr)
            hipLaunchKernelGGL(blend_backward_small_kernel<true>, sgrid, dim3(SMALL_THREADS), 0, s, bin_start, payload, a4,
                               grad_image, acc_alpha, last_effective, width, height, tile_row_begin, tile_row_step,
                               slot_offsets, p4, slot_flags, magnitude_image, debug_pixel_hits, tile_order, split, image, b4,
                               e4, counters, parts);
        else
            hipLaunchKernelGGL(blend_backward_small_kernel<false>, sgrid, dim3(SMALL_THREADS), 0, s, bin_start, payload, a4,
                               grad_image, acc_alpha, last_effective, width, height, tile_row_begin, tile_row_step,
                               slot_offsets, p4, slot_flags, magnitude_image, debug_pixel_hits, tile_order, split, image, b4,
                               e4, counters, parts);
    } else if (one_wave) {
        if (debug_pixel_hits != nullptr)
            hipLaunchKernelGGL(blend_backward_wide_kernel<true>, grid, dim3(GS_WAVE), 0, s, bin_start, payload, a4, grad_image,
                               acc_alpha, last_effective, width, height, tile_row_begin, tile_row_step, slot_offsets, p4,
                               slot_flags, magnitude_image, debug_pixel_hits, tile_order);
        else
            hipLaunchKernelGGL(blend_backward_wide_kernel<false>, grid, dim3(GS_WAVE), 0, s, bin_start, payload, a4, grad_image,
                               acc_alpha, last_effective, width, height, tile_row_begin, tile_row_step, slot_offsets, p4,
                               slot_flags, magnitude_image, debug_pixel_hits, tile_order);
    } else if (staged)
        launch_backward<true>(debug_pixel_hits != nullptr, grid, s, bin_start, payload, a4, grad_image, acc_alpha,
                              last_effective, width, height, tile_row_begin, tile_row_step, bin_shift, filter,
                              slot_offsets, p4, slot_flags, magnitude_image, debug_pixel_hits, tile_order,
                              (flags & GS_BLEND_SKEWED_WALKS) != 0);
    else
        launch_backward<false>(debug_pixel_hits != nullptr, grid, s, bin_start, payload, a4, grad_image, acc_alpha,
                               last_effective, width, height, tile_row_begin, tile_row_step, bin_shift, filter,
                               slot_offsets, p4, slot_flags, magnitude_image, debug_pixel_hits, tile_order,
                               (flags & GS_BLEND_SKEWED_WALKS) != 0);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_reduce_partials(const int32_t *slot_offsets, const int32_t *num_overlap_tiles, const uint8_t *slot_flags,
                       const float *partials, int n_visible, float *acc, const int32_t *num_keys,
                       int64_t n_slots_hint, const float *attrs, int width, int height, void *stream) {
    GS_REQUIRE(n_visible >= 0, "n_visible");
    GS_REQUIRE(attrs == nullptr || (width > 0 && height > 0 && width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0),
               "image size (needed with attrs)");
    GS_REQUIRE((reinterpret_cast<uintptr_t>(slot_flags) & 3u) == 0, "slot_flags must be 4-byte aligned (it is read as dwords)");
    if (n_visible == 0) return 0;
    const float4 *p4 = reinterpret_cast<const float4 *>(partials);
    float4 *a4 = reinterpret_cast<float4 *>(acc);
    const float4 *r4 = reinterpret_cast<const float4 *>(attrs);
    const int tw = width / GS_TILE_WIDTH, th = height / GS_TILE_HEIGHT;
    // hundreds of slots per Gaussian on average (every Gaussian is large: the reference's stress distribution): sixteen
    // lanes each.  Otherwise one lane each, the few large ones handed to their whole wave -- which also skips the empty
    // part of a needle's box, so a frame whose average is inflated by needles (72 slots per Gaussian with 1 % of screen-
    // long needles) belongs here too
    if (n_slots_hint > 512 * (int64_t)n_visible)
        hipLaunchKernelGGL(reduce_partials_kernel<16>, dim3(gs_div_up(16LL * n_visible, GS_BLOCK)), dim3(GS_BLOCK), 0,
                           (hipStream_t)stream, slot_offsets, num_overlap_tiles, slot_flags, p4, n_visible, a4, num_keys,
                           r4, tw, th);
    else
        hipLaunchKernelGGL(reduce_partials_kernel<1>, dim3(gs_div_up(n_visible, GS_BLOCK)), dim3(GS_BLOCK), 0,
                           (hipStream_t)stream, slot_offsets, num_overlap_tiles, slot_flags, p4, n_visible, a4, num_keys,
                           r4, tw, th);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
