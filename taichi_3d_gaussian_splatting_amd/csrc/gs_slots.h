// gs_slots.h -- per-Gaussian sum of the (Gaussian, tile) slot records the backward blend stores (gs_blend_backward).
// Shared by reduce_partials_kernel (gs_blend.hip: acc[M,12] in memory, what a multi-GPU run all-reduces) and by the fused
// per-point backward (gs_point_backward.hip: the sums never leave the registers).  Both run the SAME code, so a
// Gaussian's sums are the same bits either way: one lane per Gaussian walks its slots in ascending order (typically
// ~10); a Gaussian with more than RP_HEAVY slots (the reference's stress scene: 8,160 each, 2.2 ms when a single lane
// scanned them) is handed to the whole wave instead: lane l takes the slot groups l, l + 64, ... (four consecutive
// slots each) in ascending order and the 64 partial records are added in a fixed DPP order.  The summation order
// depends only on the slot layout, so gradients are bitwise reproducible.  Value 10 (pixel count) is summed as an integer.
#pragma once
#include "gs_common.h"

#ifdef __HIPCC__
constexpr int RP_HEAVY = 128;
struct SlotSum { float v[10]; int npix; };
// The flags of up to MAXCNT consecutive slots as a bit mask (bit r: slot first + r is raised).  The bytes are fetched as the
// ALIGNED dwords that hold them, all requested before the first one is looked at -- one memory latency for the whole group
// (a loop of byte loads made the compiler wait for each one in turn: about ten dependent round trips per lane, most of
// reduce_partials_kernel's 59 us at the headline size).  Flags are 0 / 1 bytes (gs_blend_backward); the buffer is
// dword-aligned and allocated in multiples of 16 bytes (include/gsplat_hip.h), so the aligned dwords around a group are
// inside it.  The first four dwords serve groups of up to 13 slots whatever their alignment; the rest is requested only
// when a lane of the wave needs it.
template <int MAXCNT>
__device__ __forceinline__ unsigned rp_flag_mask(const uint8_t *__restrict__ flags, int first, int cnt) {
    constexpr int ND = (MAXCNT + 3 + 3) / 4, ND0 = ND < 4 ? ND : 4;
    const int sh = first & 3, nd = cnt > 0 ? (sh + cnt + 3) >> 2 : 1;   // dwords that hold the group (>= 1: the loads are unconditional)
    const unsigned *w = reinterpret_cast<const unsigned *>(flags) + (cnt > 0 ? first >> 2 : 0);
    unsigned d[ND];
#pragma unroll
    for (int i = 0; i < ND0; ++i) d[i] = w[min(i, nd - 1)];   // (a repeated dword lands beyond bit cnt: masked off below)
#pragma unroll
    for (int i = ND0; i < ND; ++i) d[i] = 0u;
    if (ND > ND0 && __builtin_amdgcn_ballot_w64(nd > ND0) != 0ull) {
#pragma unroll
        for (int i = ND0; i < ND; ++i) d[i] = w[min(i, nd - 1)];
    }
    unsigned long long bits = 0ull;
#pragma unroll
    for (int i = 0; i < ND; ++i)   // bytes b0..b3 in {0,1} -> b0 | b1 << 1 | b2 << 2 | b3 << 3 (products land on distinct bits: no carries)
        bits |= (unsigned long long)((((d[i] & 0x01010101u) * 0x01020408u) >> 24) & 0xFu) << (4 * i);
    return (unsigned)(bits >> sh) & (cnt >= 32 ? ~0u : (1u << cnt) - 1u);
}

// raised slots fetched together: CHUNK 48-B records are requested before the first one is added (one record per round
// left the kernel waiting on a full memory latency per slot); added in ascending order, so CHUNK does not change a bit
template <int CHUNK, int MAXCNT = 32>
__device__ __forceinline__ void rp_add_group(const uint8_t *__restrict__ flags, const float4 *__restrict__ partials,
                                             int first, int cnt, SlotSum &a) {
    unsigned mask = rp_flag_mask<MAXCNT>(flags, first, cnt);
    while (mask) {
        int r[CHUNK];
        float4 p[CHUNK][3];
#pragma unroll
        for (int q = 0; q < CHUNK; ++q) {
            r[q] = mask ? __builtin_ctz(mask) : -1;
            mask &= mask - 1;   // (0 stays 0)
        }
#pragma unroll
        for (int q = 0; q < CHUNK; ++q)
            if (r[q] >= 0) {
                const float4 *src = partials + 3 * (size_t)(first + r[q]);
                p[q][0] = src[0]; p[q][1] = src[1]; p[q][2] = src[2];
            }
#pragma unroll
        for (int q = 0; q < CHUNK; ++q)
            if (r[q] >= 0) {
                a.v[0] += p[q][0].x; a.v[1] += p[q][0].y; a.v[2] += p[q][0].z; a.v[3] += p[q][0].w;
                a.v[4] += p[q][1].x; a.v[5] += p[q][1].y; a.v[6] += p[q][1].z; a.v[7] += p[q][1].w;
                a.v[8] += p[q][2].x; a.v[9] += p[q][2].y;
                a.npix += __builtin_bit_cast(int, p[q][2].z);
            }
    }
}

// One lane per Gaussian (+ whole-wave help for the rare heavy one).  Must be called by all 64 lanes of a wave
// (`live` false for lanes without a Gaussian); i = index into the visible list.  CHUNK / HEAVY_CHUNK: records in flight per
// lane in the ordinary / the whole-wave path (registers vs memory-level parallelism; sums do not depend on them).
template <int CHUNK, int HEAVY_CHUNK>
__device__ __forceinline__ void gs_sum_slots_of_lane(bool live, int i, const int32_t *__restrict__ slot_offsets,
                                                     const int32_t *__restrict__ ntiles_full,
                                                     const uint8_t *__restrict__ slot_flags,
                                                     const float4 *__restrict__ partials,
                                                     const int32_t *__restrict__ nkeys, const float4 *__restrict__ attrs,
                                                     int tw, int th, SlotSum &a) {
    const int lane = gs_lane();
    // a Gaussian that emitted no sort key on this GPU (tile-row sharding) was blended nowhere: no slot to look at
    const int base = live ? slot_offsets[i] : 0, n = live && (nkeys == nullptr || nkeys[i] > 0) ? ntiles_full[i] : 0;
#pragma unroll
    for (int k = 0; k < 10; ++k) a.v[k] = 0.f;
    a.npix = 0;
    if (n <= RP_HEAVY)
        for (int r0 = 0; r0 < n; r0 += 32) rp_add_group<CHUNK>(slot_flags, partials, base + r0, min(32, n - r0), a);
    // A heavy Gaussian's slots are the tiles of its reference box (column-major, gs_make_keys), but only tiles the
    // level set q <= qmax reaches can have been blended -- for a needle a thin diagonal of a huge square (10,000
    // screen-long needles: 6,000 slots each, 0.45 ms of flag scanning).  With the packed records at hand the wave
    // visits, one tile column of the cull box per lane, only the rows the level set crosses (gs_common.h, conservative).
    unsigned long long heavy = __builtin_amdgcn_ballot_w64(n > RP_HEAVY);
    int t0u = 0, t1u = 0, t0v = 0, t1v = 0, c0u = 0, c1u = 0, c0v = 0, c1v = 0;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    bool narrow = false;
    if (attrs != nullptr && n > RP_HEAVY) {
        r0 = attrs[4 * (size_t)i]; r1 = attrs[4 * (size_t)i + 1];
        gs_tile_box(r0.x, r0.y, r1.w, tw, th, t0u, t1u, t0v, t1v);
        c0u = t0u; c1u = t1u; c0v = t0v; c1v = t1v;
        const float det = r1.x * r1.z - r1.y * r1.y;
        narrow = r0.w < 1e30f && r0.w >= 0.f && det > 0.f && (t1u - t0u) * (t1v - t0v) == n;
        if (narrow) gs_cull_box(r0.x, r0.y, r1.x, r1.y, r1.z, r0.w, c0u, c1u, c0v, c1v);
    }
    while (heavy) {   // wave-uniform loop over the heavy Gaussians of this wave
        const int L = __builtin_ctzll(heavy);
        heavy &= heavy - 1;
        const int bL = __builtin_amdgcn_readlane(base, L), nL = __builtin_amdgcn_readlane(n, L);
        SlotSum h;
#pragma unroll
        for (int k = 0; k < 10; ++k) h.v[k] = 0.f;
        h.npix = 0;
        if (__builtin_amdgcn_readlane((int)narrow, L)) {
            const float u = gs_readlane_f(r0.x, L), v = gs_readlane_f(r0.y, L), qm = gs_readlane_f(r0.w, L);
            const float A = gs_readlane_f(r1.x, L), B = gs_readlane_f(r1.y, L), C = gs_readlane_f(r1.z, L);
            const int b0u = __builtin_amdgcn_readlane(t0u, L), b0v = __builtin_amdgcn_readlane(t0v, L);
            const int nv = __builtin_amdgcn_readlane(t1v, L) - b0v;
            const int k0u = __builtin_amdgcn_readlane(c0u, L), k1u = __builtin_amdgcn_readlane(c1u, L);
            const int k0v = __builtin_amdgcn_readlane(c0v, L), k1v = __builtin_amdgcn_readlane(c1v, L);
            for (int cu = k0u + lane; cu < k1u; cu += GS_WAVE) {   // one tile column per lane and round
                int ra, rb;
                gs_cull_rows_in_column(u, v, A, B, C, qm, cu, ra, rb);
                ra = max(ra, k0v); rb = min(rb, k1v);
                const int first = bL + nv * (cu - b0u) - b0v;   // slot of (cu, row) = first + row
                for (int row = ra; row < rb; row += 4)
                    rp_add_group<HEAVY_CHUNK, 4>(slot_flags, partials, first + row, min(4, rb - row), h);
            }
        } else {
            for (int r0_ = 4 * lane; r0_ < nL; r0_ += 4 * GS_WAVE)
                rp_add_group<HEAVY_CHUNK, 4>(slot_flags, partials, bL + r0_, min(4, nL - r0_), h);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float t = gs_readlane63(gs_wave_sum_to_lane63(h.v[k]));
            if (lane == L) a.v[k] = t;
        }
        const float np = gs_readlane63(gs_wave_sum_to_lane63((float)h.npix));   // < 2^24: exact as a float
        if (lane == L) a.npix = (int)np;
    }
}
#endif  // __HIPCC__
