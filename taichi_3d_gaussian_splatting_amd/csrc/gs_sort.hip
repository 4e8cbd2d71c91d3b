// gs_sort.hip -- stable LSD radix sort of (64-bit key, 32-bit payload) pairs for gfx950.
// Replaces torch.sort + gather of the reference (RAS:947-950) with the STABLE tie rule
// (ties keep key-generation order = ascending offset into the visible list).
//
// Only the bits that can differ are sorted: the quantised-depth field [0, depth_bits) and the
// tile field [32, 32+tile_bits); 8-bit digits.  Per pass:
//   1. digit histogram per workgroup (GS_BLOCK x rounds keys each) -> counts[digit][block]
//   2. exclusive scan of every digit row + digit totals            (256 workgroups)
//   3. stable scatter: wave-level digit matching with ballots (64-lane match-any), per-wave running digit
//      counters in LDS, block-local permutation in LDS, coalesced write-out of each digit's run.
// All hand-written; no rocPRIM/hipCUB.
#include "gs_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int RADIX_BITS = 8;
#ifndef GS_SORT_ROUNDS
#define GS_SORT_ROUNDS 16   // (tuning variants: tools/build_variants.sh)
#endif
#ifndef GS_SORT_SMALL_ROUNDS
#define GS_SORT_SMALL_ROUNDS 4   // rounds per wave when the large workgroups would not fill the chip (see sort_rounds_for)
#endif
#ifndef GS_SORT_SMALL_BLOCKS
#define GS_SORT_SMALL_BLOCKS 512   // fewer large workgroups than this: sort with the small ones
#endif
constexpr int WAVES = GS_BLOCK / GS_WAVE;

// Keys per workgroup = GS_BLOCK x rounds.  16 rounds amortise the per-workgroup prefix work best, but a workgroup ranks its
// rounds one after the other (a dependent chain of ballots and LDS counter updates): with few keys -- small frames, a
// GPU's band of a sharded frame -- 4,096-key workgroups leave most of the 256 CUs idle and the launch lasts as long as
// one workgroup's chain (17 us for 360k keys where 2.9M keys take 23).  Below GS_SORT_SMALL_BLOCKS large workgroups the
// sort runs with a quarter of the chain per workgroup and four times the workgroups.
static int sort_rounds_for(int64_t n_keys) {
    return gs_div_up(n_keys > 0 ? n_keys : 1, GS_BLOCK * GS_SORT_ROUNDS) < GS_SORT_SMALL_BLOCKS ? GS_SORT_SMALL_ROUNDS
                                                                                               : GS_SORT_ROUNDS;
}

template <typename KeyT, int RB = RADIX_BITS>
__device__ __forceinline__ unsigned digit_of(KeyT key, int shift, KeyT flip) {
    return (unsigned)(((key ^ flip) >> shift) & ((1 << RB) - 1));
}

// (RB: bits of the digit -- 8 for the LSD passes; 8, 9 or 10 for the one partitioning pass of the MSD-first sort)
template <typename KeyT, int SORT_ROUNDS, int RB>
__global__ __launch_bounds__(GS_BLOCK) void sort_hist_kernel(const KeyT *__restrict__ keys, long long n,
                                                            const int32_t *__restrict__ n_device, int shift,
                                                            KeyT flip, int nblk, int32_t *__restrict__ counts,
                                                            int4 *__restrict__ also_zero, long long also_zero_int4) {
    constexpr int SORT_ITEMS = GS_BLOCK * SORT_ROUNDS;
    constexpr int DIG = 1 << RB;
    __shared__ int hist[WAVES][DIG];
    // a caller's zero-fill rides along with the first launch of the sort (the frame's list ranges: no fill launch of its own)
    for (long long z = (long long)blockIdx.x * GS_BLOCK + threadIdx.x; z < also_zero_int4; z += (long long)gridDim.x * GS_BLOCK)
        also_zero[z] = make_int4(0, 0, 0, 0);
    if (n_device) n = min((long long)*n_device, n);   // grid and workspace are sized by the capacity n   // one histogram per wave: a quarter of the same-address LDS atomics
#pragma unroll
    for (int k = 0; k < WAVES; ++k)
#pragma unroll
        for (int d = threadIdx.x; d < DIG; d += GS_BLOCK) hist[k][d] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const long long base = (long long)blockIdx.x * SORT_ITEMS;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        long long i = base + r * GS_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&hist[w][digit_of<KeyT, RB>(keys[i], shift, flip)], 1);
    }
    __syncthreads();
#pragma unroll
    for (int d = threadIdx.x; d < DIG; d += GS_BLOCK) {
        int c = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) c += hist[k][d];
        counts[(size_t)d * nblk + blockIdx.x] = c;
    }
}

// workgroup d: exclusive scan of row d (nblk entries) in place, row total -> totals[d]
__global__ __launch_bounds__(GS_BLOCK) void sort_scan_rows_kernel(int32_t *__restrict__ counts, int nblk,
                                                                 int32_t *__restrict__ totals) {
    __shared__ int lds[4];
    int32_t *row = counts + (size_t)blockIdx.x * nblk;
    int carry = 0;
    for (int base = 0; base < nblk; base += GS_BLOCK) {
        int i = base + threadIdx.x;
        int v = i < nblk ? row[i] : 0;
        int total;
        int ex = gs_block_excl_scan(v, &total, lds);
        if (i < nblk) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// Stable scatter of one 8-bit digit.  A workgroup owns SORT_ITEMS consecutive keys, wave w the w-th quarter of them
// (rounds of 64 consecutive keys, so global reads are coalesced and the block order is (wave, round, lane)).
//   1. ranking: per round a 64-lane match-any on the digit gives the rank inside the wave; a per-wave running
//      digit counter in LDS gives the keys of earlier rounds -- no workgroup barrier inside the loop;
//   2. the per-wave counters are turned into exclusive prefixes over waves and digits (block-local digit starts);
//   3. keys and payloads are permuted into digit order IN LDS, then written out with consecutive threads writing
//      consecutive addresses of a digit's run.  (Scattering straight from registers wrote 4-byte pieces all over
//      the output: rocprofv3 showed 2.8x the algorithmic HBM write bytes.)
template <typename KeyT, int SORT_ROUNDS, int RB>
__global__ __launch_bounds__(GS_BLOCK) void sort_scatter_kernel(
    const KeyT *__restrict__ keys_in, const int32_t *__restrict__ payload_in, long long n,
    const int32_t *__restrict__ n_device, int shift, KeyT flip, int nblk, const int32_t *__restrict__ row_offsets,
    const int32_t *__restrict__ totals, KeyT *__restrict__ keys_out, int32_t *__restrict__ payload_out) {
    constexpr int SORT_ITEMS = GS_BLOCK * SORT_ROUNDS;
    constexpr int DIG = 1 << RB, PER = DIG / GS_BLOCK;   // thread t looks after the digits [t PER, (t + 1) PER)
    if (n_device) n = min((long long)*n_device, n);
    __shared__ int s_cnt[WAVES][DIG];   // running per-wave digit counts, later exclusive prefix over waves
    __shared__ unsigned long long s_reg[WAVES][DIG];   // per wave and digit: the lanes that hold it this round (match-any)
    __shared__ int s_local[DIG];        // block-local start of each digit's run
    __shared__ int s_gbase[DIG];        // global position of the block's first key of each digit
    __shared__ KeyT s_keys[SORT_ITEMS];
    __shared__ int32_t s_pay[SORT_ITEMS];
    __shared__ int lds[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    {
        int tot[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { tot[j] = totals[threadIdx.x * PER + j]; sum += tot[j]; }
        int total;
        int ex = gs_block_excl_scan(sum, &total, lds);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            s_gbase[threadIdx.x * PER + j] = ex + row_offsets[(size_t)(threadIdx.x * PER + j) * nblk + blockIdx.x];
            ex += tot[j];
        }
    }
#pragma unroll
    for (int k = 0; k < WAVES; ++k)
#pragma unroll
        for (int d = threadIdx.x; d < DIG; d += GS_BLOCK) { s_cnt[k][d] = 0; s_reg[k][d] = 0ull; }
    __syncthreads();
    const long long block_base = (long long)blockIdx.x * SORT_ITEMS;
    const long long wave_base = block_base + (long long)w * (SORT_ITEMS / WAVES);
    // (LDS-qualified volatile accesses: the LDS operations of a wave to its counters stay in program order; a generic volatile
    // pointer makes this compiler emit a flat-address null check it cannot encode, see sort_local_kernel)
#define SC_CNT(d) (*(volatile __attribute__((address_space(3))) int *)(&s_cnt[w][d]))
#define SC_REG(d) (*(volatile __attribute__((address_space(3))) unsigned long long *)(&s_reg[w][d]))
    KeyT key[SORT_ROUNDS];
    int32_t pay[SORT_ROUNDS];
    int rnk[SORT_ROUNDS];   // rank among the same-digit keys of this wave; -1 = past the end of the array
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const long long i = wave_base + r * GS_WAVE + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : (KeyT)0;
        pay[r] = valid ? payload_in[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const long long i = wave_base + r * GS_WAVE + lane;
        const bool valid = i < n;
        const unsigned d = digit_of<KeyT, RB>(key[r], shift, flip);
        // 64-lane match-any on the digit THROUGH LDS (round 6: as sort_local_kernel has done since round 4): every lane ORs
        // its bit into its digit's word and reads the word back -- the lanes of the wave that hold the same digit.  OR
        // commutes, a wave's LDS operations execute in program order.  (RB ballots with their per-lane 64-bit selects were
        // ~45 VALU instructions per round: sixteen dependent rounds of them per wave.)
        if (valid) __hip_atomic_fetch_or(&s_reg[w][d], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        const unsigned long long peers = valid ? SC_REG(d) : 0ull;
        const int rank = gs_mbcnt(peers);
        const int before = valid ? SC_CNT(d) : 0;                       // every lane of a group reads ...
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) {
            SC_CNT(d) = before + __popcll(peers);   // ... before its leader bumps the counter
            SC_REG(d) = 0ull;                       // ... and clears the word for the next round
        }
        rnk[r] = valid ? before + rank : -1;
    }
#undef SC_CNT
#undef SC_REG
    __syncthreads();
    {
        int run[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            run[j] = 0;
#pragma unroll
            for (int k = 0; k < WAVES; ++k) {
                const int c = s_cnt[k][threadIdx.x * PER + j];
                s_cnt[k][threadIdx.x * PER + j] = run[j];
                run[j] += c;
            }
            sum += run[j];
        }
        int total;
        int ex = gs_block_excl_scan(sum, &total, lds);
#pragma unroll
        for (int j = 0; j < PER; ++j) { s_local[threadIdx.x * PER + j] = ex; ex += run[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        if (rnk[r] >= 0) {
            const unsigned d = digit_of<KeyT, RB>(key[r], shift, flip);
            const int pos = s_local[d] + s_cnt[w][d] + rnk[r];
            s_keys[pos] = key[r];
            s_pay[pos] = pay[r];
        }
    }
    __syncthreads();
    const long long left = n - block_base;
    const int nb = left < SORT_ITEMS ? (int)left : SORT_ITEMS;
    for (int p = threadIdx.x; p < nb; p += GS_BLOCK) {
        const KeyT k = s_keys[p];
        const unsigned d = digit_of<KeyT, RB>(k, shift, flip);
        const int dst = s_gbase[d] + (p - s_local[d]);
        keys_out[dst] = k;
        payload_out[dst] = s_pay[p];
    }
}

// ---------------------------------------------------------------------------------------------- bucket-local sort
// Second half of the MSD-first sort of compressed 32-bit keys (sort_msd_first below): after ONE stable scatter pass on
// the TOP eight to ten bits in use the array is 256 ... 1,024 buckets in key-generation order, bucket b = the keys whose top
// digit is b, totals[b] long.  Workgroup b sorts its bucket by the remaining low_bits with stable LSD passes of <= 8 bits that never
// leave the CU: a bucket of up to LS_CAP pairs sits in registers (16 pairs per thread) and is permuted through the 160-KB
// LDS -- no histogram / row-scan launches, no HBM traffic between the passes, one read and one write of the pairs.
// Ranking is the scatter kernel's: 64-lane match-any on the digit + per-wave running digit counters, wave w owns a
// contiguous segment of the chunk so (wave, round, lane) is list order.  A bucket LARGER than LS_CAP (a skewed key
// distribution) is handled by the same code chunk by chunk, with the bucket's range of the other buffer as the
// intermediate (digit histogram of the bucket first, then every chunk's runs appended behind the earlier chunks'): slower,
// same result.
constexpr int LS_THREADS = 1024;
constexpr int LS_WAVES = LS_THREADS / GS_WAVE;
constexpr int LS_ROUNDS = 16;                        // pairs per thread
constexpr int LS_CAP = LS_THREADS * LS_ROUNDS;       // pairs of a chunk
constexpr int LS_BITS = 6;                           // bits per bucket-local pass (64 digits = the lanes of a wave)
constexpr int LS_MAX_RADIX = 1 << LS_BITS;

// exclusive scan of `v` over the 1,024 threads of the workgroup into `ex` (two barriers).  A macro over the LDS array
// itself: handed to a function as a pointer the array loses its address space, and this compiler then emits a flat-address
// null check it cannot encode ("Illegal instruction detected: V_CMP_NE_U32 0, src_shared_base").
#define LS_BLOCK_EXCL_SCAN(ex, v, wave_tot, first)                          \
    do {                                                                    \
        const int _incl = gs_wave_incl_scan(v);                             \
        if (lane == GS_WAVE - 1) wave_tot[(first) + w] = _incl;            \
        __syncthreads();                                                    \
        int _base = 0;                                                      \
        for (int _i = 0; _i < w; ++_i) _base += wave_tot[(first) + _i];    \
        __syncthreads();                                                    \
        ex = _base + _incl - (v);                                           \
    } while (0)

__global__ __launch_bounds__(LS_THREADS) void sort_local_kernel(uint32_t *__restrict__ keys, int32_t *__restrict__ payload,
                                                                uint32_t *__restrict__ keys_tmp,
                                                                int32_t *__restrict__ payload_tmp,
                                                                const int32_t *__restrict__ totals, int low_bits,
                                                                int part_shift, int part_bits, int key_depth_bits,
                                                                int n_tiles, int32_t *__restrict__ tile_start,
                                                                int32_t *__restrict__ tile_end) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "sort_local_kernel keeps a whole bucket in the 160 KB of LDS a gfx950 workgroup can have: this library is written for gfx950 (CDNA4) only"
#endif
    __shared__ uint32_t s_keys[LS_CAP];                 // (gfx950: 160 KB of LDS per workgroup)
    __shared__ int32_t s_pay[LS_CAP];
    __shared__ int s_cnt[LS_WAVES][LS_MAX_RADIX];       // running per-wave digit counts, then exclusive prefixes over waves
    __shared__ int s_local[LS_MAX_RADIX];               // chunk-local start of each digit's run
    __shared__ int s_dbase[LS_MAX_RADIX];               // bucket-wide start of each digit (chunked buckets)
    __shared__ int s_drun[LS_MAX_RADIX];                // keys of the digit written by earlier chunks
    __shared__ int s_ccnt[LS_MAX_RADIX];                // the chunk's keys of each digit
    __shared__ unsigned long long s_reg[LS_WAVES][LS_MAX_RADIX];   // per wave and digit: the lanes that hold it this round
    __shared__ int s_wave[2 * LS_WAVES];                // scratch of the block scans
    __shared__ int s_misc[2];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w == 0) {   // the bucket's range: start = sum of the totals before it
        int part = 0;
        for (int j = lane; j < (int)blockIdx.x; j += GS_WAVE) part += totals[j];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, GS_WAVE);
        if (lane == 0) { s_misc[0] = part; s_misc[1] = totals[blockIdx.x]; }
    }
    __syncthreads();
    const int start = s_misc[0], nb = s_misc[1];
    if (nb == 0) return;   // (workgroup-uniform)
    // tile_start != NULL: the bucket is made of whole bins (the partitioning digit lies inside the bin field), so the
    // [start, end) range of every bin (RAS:175-193; arrays zeroed by the sort's first launch) is known right here
#define LS_EMIT_RANGE(k, prev_differs, next_differs, pos)                                              \
    do {                                                                                               \
        const unsigned _bin = (k) >> key_depth_bits;                                                   \
        if (_bin < (unsigned)n_tiles) {   /* (keys the caller did not generate are never written) */   \
            if (prev_differs) tile_start[_bin] = start + (pos);                                        \
            if (next_differs) tile_end[_bin] = start + (pos) + 1;                                      \
        }                                                                                              \
    } while (0)
    if (nb == 1) {
        if (tile_start != nullptr && threadIdx.x == 0) LS_EMIT_RANGE(keys[start], true, true, 0);
        return;
    }
    // the bits a bucket is still to be sorted by, closed up: those above the partitioning digit moved down onto it
    const uint32_t below = (1u << part_shift) - 1u;
    const int above_shift = part_shift + part_bits;
#define LS_REST(k) ((above_shift >= 32 ? 0u : ((k) >> above_shift) << part_shift) | ((k) & below))
    const int npass = (low_bits + LS_BITS - 1) / LS_BITS;
    const int lbits = (low_bits + npass - 1) / npass;   // <= 6 bits per pass, passes of equal width
    const int radix = 1 << lbits;
    const bool in_lds = nb <= LS_CAP;
    uint32_t *src_k = keys + start, *dst_k = keys_tmp + start;
    int32_t *src_p = payload + start, *dst_p = payload_tmp + start;
    uint32_t key[LS_ROUNDS];
    int32_t pay[LS_ROUNDS];
    int rnk[LS_ROUNDS];
    // (volatile: the LDS accesses of a wave to its counters stay in program order)
    // (an LDS-qualified pointer: through a generic volatile pointer this compiler emits a flat-address null check it cannot
    // encode -- "Illegal instruction detected: V_CMP_NE_U32 0, src_shared_base")
#define LS_CNT(d) (*(volatile __attribute__((address_space(3))) int *)(&s_cnt[w][d]))
#define LS_REG(d) (*(volatile __attribute__((address_space(3))) unsigned long long *)(&s_reg[w][d]))
    LS_REG(lane) = 0ull;   // (64 digits = 64 lanes; every round's leader leaves its word zero again)
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = pass * lbits;
        const unsigned dmask = (unsigned)radix - 1u;   // (LS_REST has low_bits bits)
        if (!in_lds) {   // bucket-wide digit starts of this pass
            for (int d = lane; d < radix; d += GS_WAVE) LS_CNT(d) = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < nb; i += LS_THREADS) atomicAdd(&s_cnt[w][(LS_REST(src_k[i]) >> shift) & dmask], 1);
            __syncthreads();
            int t = 0;
            if ((int)threadIdx.x < radix)
                for (int k = 0; k < LS_WAVES; ++k) t += s_cnt[k][threadIdx.x];
            int ex;
            LS_BLOCK_EXCL_SCAN(ex, t, s_wave, 0);
            if ((int)threadIdx.x < radix) { s_dbase[threadIdx.x] = ex; s_drun[threadIdx.x] = 0; }
            __syncthreads();
        }
        const int nchunks = in_lds ? 1 : (nb + LS_CAP - 1) / LS_CAP;
        for (int c = 0; c < nchunks; ++c) {
            const int cbase = c * LS_CAP;
            const int cn = min(nb - cbase, LS_CAP);
            const int nrounds = (cn + LS_THREADS - 1) / LS_THREADS;   // rounds of 64 consecutive pairs per wave
            const int seg = nrounds * GS_WAVE;                        // wave w owns positions [w seg, (w + 1) seg)
            if (pass == 0 || !in_lds) {
#pragma unroll
                for (int r = 0; r < LS_ROUNDS; ++r) {
                    const int p = w * seg + r * GS_WAVE + lane;
                    const bool valid = r < nrounds && p < cn;
                    key[r] = valid ? src_k[cbase + p] : 0u;
                    pay[r] = valid ? src_p[cbase + p] : 0;
                }
            }
            for (int d = lane; d < radix; d += GS_WAVE) LS_CNT(d) = 0;
#pragma unroll
            for (int r = 0; r < LS_ROUNDS; ++r) {
                rnk[r] = -1;
                if (r < nrounds) {   // (workgroup-uniform)
                    const int p = w * seg + r * GS_WAVE + lane;
                    const bool valid = p < cn;
                    const unsigned d = (LS_REST(key[r]) >> shift) & dmask;
                    // 64-lane match-any on the digit through LDS: every lane ORs its bit into its digit's word, then
                    // reads the word back -- the lanes of the wave that hold the same digit.  OR commutes, so the word does
                    // not depend on the order in which the LDS serves the lanes; a wave's LDS operations execute in
                    // program order, so all bits are in before the first read.  (The scatter kernel's eight ballots with
                    // their per-lane 64-bit selects cost ~45 VALU instructions per round here and made this kernel
                    // VALU-bound: 2,575 -> see profiles/r04_sort.md.)
                    if (valid) __hip_atomic_fetch_or(&s_reg[w][d], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __builtin_amdgcn_wave_barrier();   // (no instruction: keeps the compiler from moving the read above the ORs)
                    const unsigned long long peers = valid ? LS_REG(d) : 0ull;
                    const int rank = gs_mbcnt(peers);
                    const int before = valid ? LS_CNT(d) : 0;                       // every lane of a group reads ...
                    __builtin_amdgcn_wave_barrier();   // (... and the leader's clear below the reads)
                    if (valid && rank == 0) {
                        LS_CNT(d) = before + __popcll(peers);   // ... before its leader bumps the counter
                        LS_REG(d) = 0ull;                       // ... and clears the word for the next round
                    }
                    rnk[r] = valid ? before + rank : -1;
                }
            }
            __syncthreads();
            {
                int run = 0;
                if ((int)threadIdx.x < radix) {
                    for (int k = 0; k < LS_WAVES; ++k) {
                        const int v = s_cnt[k][threadIdx.x];
                        s_cnt[k][threadIdx.x] = run;
                        run += v;
                    }
                    s_ccnt[threadIdx.x] = run;
                }
                int ex;
                LS_BLOCK_EXCL_SCAN(ex, run, s_wave, LS_WAVES);
                if ((int)threadIdx.x < radix) s_local[threadIdx.x] = ex;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < LS_ROUNDS; ++r) {
                if (rnk[r] >= 0) {
                    const unsigned d = (LS_REST(key[r]) >> shift) & dmask;
                    const int pos = s_local[d] + s_cnt[w][d] + rnk[r];
                    s_keys[pos] = key[r];
                    s_pay[pos] = pay[r];
                }
            }
            __syncthreads();
            if (in_lds) {
                if (pass + 1 < npass) {   // back into the registers, in position order
#pragma unroll
                    for (int r = 0; r < LS_ROUNDS; ++r) {
                        const int p = w * seg + r * GS_WAVE + lane;
                        if (r < nrounds && p < cn) { key[r] = s_keys[p]; pay[r] = s_pay[p]; }
                    }
                } else {                  // sorted: out, in place
                    for (int p = threadIdx.x; p < cn; p += LS_THREADS) {
                        const uint32_t k = s_keys[p];
                        src_k[p] = k;
                        src_p[p] = s_pay[p];
                        if (tile_start != nullptr)
                            LS_EMIT_RANGE(k, p == 0 || (s_keys[p - 1] >> key_depth_bits) != (k >> key_depth_bits),
                                          p == cn - 1 || (s_keys[p + 1] >> key_depth_bits) != (k >> key_depth_bits), p);
                    }
                }
            } else {   // the chunk's digit runs behind the earlier chunks' in the other buffer
                for (int p = threadIdx.x; p < cn; p += LS_THREADS) {
                    const uint32_t k = s_keys[p];
                    const unsigned d = (LS_REST(k) >> shift) & dmask;
                    const int dst = s_dbase[d] + s_drun[d] + (p - s_local[d]);
                    dst_k[dst] = k;
                    dst_p[dst] = s_pay[p];
                }
                __syncthreads();
                if ((int)threadIdx.x < radix) s_drun[threadIdx.x] += s_ccnt[threadIdx.x];
                __syncthreads();
            }
        }
        if (!in_lds) {
            uint32_t *tk = src_k; src_k = dst_k; dst_k = tk;
            int32_t *tp = src_p; src_p = dst_p; dst_p = tp;
        }
    }
    if (!in_lds && (npass & 1)) {   // an odd number of passes left a chunked bucket in the other buffer
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += LS_THREADS) { dst_k[i] = src_k[i]; dst_p[i] = src_p[i]; }
    }
    if (!in_lds && tile_start != nullptr) {   // a chunked bucket lies sorted in global memory
        __syncthreads();
        const uint32_t *sorted = keys + start;
        for (int i = threadIdx.x; i < nb; i += LS_THREADS) {
            const uint32_t k = sorted[i];
            LS_EMIT_RANGE(k, i == 0 || (sorted[i - 1] >> key_depth_bits) != (k >> key_depth_bits),
                          i == nb - 1 || (sorted[i + 1] >> key_depth_bits) != (k >> key_depth_bits), i);
        }
    }
#undef LS_EMIT_RANGE
#undef LS_REST
#undef LS_CNT
#undef LS_REG
#undef LS_BLOCK_EXCL_SCAN
}

}  // namespace

template <typename KeyT, int SORT_ROUNDS, int RB = RADIX_BITS>
static int sort_passes(KeyT *keys, int32_t *payload, KeyT *keys_alt, int32_t *payload_alt, int64_t n_keys,
                       const int32_t *n_dev, const int *shifts, int n_pass, KeyT flip, int allow_result_in_alt,
                       void *workspace, hipStream_t s, void *also_zero, size_t also_zero_bytes) {
    if (n_pass == 0 && also_zero_bytes) GS_CHECK_HIP(hipMemsetAsync(also_zero, 0, also_zero_bytes, s));
    const int nblk = gs_div_up(n_keys, GS_BLOCK * SORT_ROUNDS);
    constexpr int DIG = 1 << RB;
    int32_t *counts = (int32_t *)workspace;
    int32_t *totals = counts + (size_t)DIG * nblk;
    KeyT *kin = keys, *kout = keys_alt;
    int32_t *pin = payload, *pout = payload_alt;
    for (int p = 0; p < n_pass; ++p) {
        hipLaunchKernelGGL((sort_hist_kernel<KeyT, SORT_ROUNDS, RB>), dim3(nblk), dim3(GS_BLOCK), 0, s, kin, (long long)n_keys,
                           n_dev, shifts[p], flip, nblk, counts, (int4 *)also_zero,
                           (long long)(p == 0 ? also_zero_bytes / 16 : 0));
        GS_CHECK_LAUNCH();
        hipLaunchKernelGGL(sort_scan_rows_kernel, dim3(DIG), dim3(GS_BLOCK), 0, s, counts, nblk, totals);
        GS_CHECK_LAUNCH();
        hipLaunchKernelGGL((sort_scatter_kernel<KeyT, SORT_ROUNDS, RB>), dim3(nblk), dim3(GS_BLOCK), 0, s, kin, pin,
                           (long long)n_keys, n_dev, shifts[p], flip, nblk, counts, totals, kout, pout);
        GS_CHECK_LAUNCH();
        KeyT *tk = kin; kin = kout; kout = tk;
        int32_t *tp = pin; pin = pout; pout = tp;
    }
    if (kin != keys) {  // odd number of passes: result sits in the alt buffers
        if (allow_result_in_alt) return 1;
        GS_CHECK_HIP(hipMemcpyAsync(keys, kin, sizeof(KeyT) * n_keys, hipMemcpyDeviceToDevice, s));
        GS_CHECK_HIP(hipMemcpyAsync(payload, pin, sizeof(int32_t) * n_keys, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

template <typename KeyT, int RB = RADIX_BITS>
static int sort_pairs_impl(KeyT *keys, int32_t *payload, KeyT *keys_alt, int32_t *payload_alt, int64_t n_keys,
                           const int32_t *n_dev, const int *shifts, int n_pass, KeyT flip, int allow_result_in_alt,
                           void *workspace, hipStream_t s, void *also_zero, size_t also_zero_bytes) {
    // (n_keys is the capacity when the count lives on the device: the choice follows the capacity, as the grids do)
    if (sort_rounds_for(n_keys) == GS_SORT_SMALL_ROUNDS)
        return sort_passes<KeyT, GS_SORT_SMALL_ROUNDS, RB>(keys, payload, keys_alt, payload_alt, n_keys, n_dev, shifts, n_pass,
                                                       flip, allow_result_in_alt, workspace, s, also_zero, also_zero_bytes);
    return sort_passes<KeyT, GS_SORT_ROUNDS, RB>(keys, payload, keys_alt, payload_alt, n_keys, n_dev, shifts, n_pass, flip,
                                             allow_result_in_alt, workspace, s, also_zero, also_zero_bytes);
}

// MSD-first sort of compressed 32-bit keys with `bits` (> 8) bits in use: ONE histogram / row-scan / scatter pass on the top
// eight of them (the three launches of an LSD pass), then sort_local_kernel on the 256 buckets -- four launches where the
// LSD sort takes three per eight bits, and one round trip of the pairs through HBM less per pass saved.  Pays while a
// bucket fits a workgroup's LDS (n_keys / 256 pairs on average): the caller falls back to the LSD passes above that.
constexpr int MSD_MAX_BITS = 9;   // (1,024 buckets measured slower than the LSD passes at every size: profiles/r04_sort.md)
#ifndef GS_SORT_MSD_BUCKET_KEYS
#define GS_SORT_MSD_BUCKET_KEYS 8200          // pairs per bucket that a CAPACITY (1.3 x the previous frame's keys) may hold on
#endif                                        // average when the buckets are runs of ADJACENT bins (ascending order asked for):
                                              // 6,300 actual pairs -- such buckets differ by 2.3x (headline: mean 5.6k, max 13.2k)
#ifndef GS_SORT_MSD_GROUPED_BUCKET_KEYS
#define GS_SORT_MSD_GROUPED_BUCKET_KEYS 17000 // the same when a bucket is every 256th / 512th bin of the frame: 13,000 actual
#endif                                        // pairs, even buckets

// bits of the partitioning digit for `n_keys` (a capacity) pairs: the fewest of 8, 9 that keep the average bucket at
// `bucket_keys`, 0 = too many keys (the LSD passes sort them)
static int msd_digit_bits(int64_t n_keys, int bits, int64_t bucket_keys) {
    for (int m = RADIX_BITS; m <= MSD_MAX_BITS; ++m)
        if (n_keys <= bucket_keys << m) return m < bits ? m : bits - 1;   // (bits > 8: at least one bit left to sort by)
    return 0;
}

// part_shift: position of the partitioning digit -- bits - RB (the top bits: ascending output) or the lowest bits of the
// bin field (bins in any order, see gs_sort_pairs_and_zero)
template <int RB>
static int sort_msd_first_rb(uint32_t *keys, int32_t *payload, uint32_t *keys_alt, int32_t *payload_alt, int64_t n_keys,
                             const int32_t *n_dev, int bits, int part_shift, int allow_result_in_alt, void *workspace,
                             hipStream_t s, void *also_zero, size_t also_zero_bytes, int key_depth_bits, int n_tiles,
                             int32_t *tile_start, int32_t *tile_end) {
    const int rc = sort_pairs_impl<uint32_t, RB>(keys, payload, keys_alt, payload_alt, n_keys, n_dev, &part_shift, 1, 0u, 1,
                                                 workspace, s, also_zero, also_zero_bytes);
    if (rc < 0) return rc;   // (rc == 1: the partitioned pairs are in the alt buffers)
    const int nblk = gs_div_up(n_keys, GS_BLOCK * sort_rounds_for(n_keys));
    const int32_t *totals = (const int32_t *)workspace + ((size_t)nblk << RB);
    hipLaunchKernelGGL(sort_local_kernel, dim3(1 << RB), dim3(LS_THREADS), 0, s, keys_alt, payload_alt, keys, payload,
                       totals, bits - RB, part_shift, RB, key_depth_bits, n_tiles, tile_start, tile_end);
    GS_CHECK_LAUNCH();
    const int ranges_written = tile_start != nullptr ? 2 : 0;
    if (allow_result_in_alt) return 1 | ranges_written;
    GS_CHECK_HIP(hipMemcpyAsync(keys, keys_alt, sizeof(uint32_t) * n_keys, hipMemcpyDeviceToDevice, s));
    GS_CHECK_HIP(hipMemcpyAsync(payload, payload_alt, sizeof(int32_t) * n_keys, hipMemcpyDeviceToDevice, s));
    return ranges_written;
}

extern "C" {

size_t gs_sort_workspace_bytes(int64_t n_keys) {
    const size_t nblk = (size_t)gs_div_up(n_keys > 0 ? n_keys : 1, GS_BLOCK * sort_rounds_for(n_keys));
    return sizeof(int32_t) * (((size_t)nblk << MSD_MAX_BITS) + (1 << MSD_MAX_BITS) + 64);   // (2^MSD_MAX_BITS = 512 digit rows at most)
}

int gs_sort_pairs(void *keys, int32_t *payload, void *keys_alt, int32_t *payload_alt, int64_t n_keys,
                  const int32_t *n_keys_device, int key_depth_bits, int depth_bits, int tile_bits,
                  int allow_result_in_alt, void *workspace, void *stream) {
    return gs_sort_pairs_and_zero(keys, payload, keys_alt, payload_alt, n_keys, n_keys_device, key_depth_bits, depth_bits,
                                  tile_bits, allow_result_in_alt, 0, workspace, nullptr, 0, nullptr, nullptr, 0, stream);
}

int gs_sort_pairs_and_zero(void *keys, int32_t *payload, void *keys_alt, int32_t *payload_alt, int64_t n_keys,
                           const int32_t *n_keys_device, int key_depth_bits, int depth_bits, int tile_bits,
                           int allow_result_in_alt, int bins_in_any_order, void *workspace, void *also_zero,
                           size_t also_zero_bytes, int32_t *tile_start, int32_t *tile_end, int n_tiles, void *stream) {
    GS_REQUIRE((tile_start == nullptr) == (tile_end == nullptr) && n_tiles >= 0, "ranges");
    GS_REQUIRE(also_zero_bytes % 16 == 0 && ((uintptr_t)also_zero & 15) == 0, "also_zero must be 16-byte aligned");
    GS_REQUIRE(n_keys >= 0 && n_keys < 0x7fffffffLL, "n_keys must fit int32");
    GS_REQUIRE(depth_bits >= 0 && depth_bits <= 64 && tile_bits >= 0 && tile_bits <= 31, "bit ranges");
    GS_REQUIRE(key_depth_bits >= 0 && key_depth_bits < 32, "key_depth_bits");
    hipStream_t s = (hipStream_t)stream;
    if (n_keys <= 1) {
        if (also_zero_bytes) GS_CHECK_HIP(hipMemsetAsync(also_zero, 0, also_zero_bytes, s));
        return 0;
    }
    int shifts[16], n_pass = 0;
    if (key_depth_bits > 0) {  // compressed 32-bit keys: one contiguous field
        GS_REQUIRE(key_depth_bits + tile_bits <= 32, "compressed key does not fit 32 bits");
        // GS_SORT_IMPL=lsd: the LSD passes at every size (A/B measurements)
        static const bool msd_allowed = !(getenv("GS_SORT_IMPL") && strcmp(getenv("GS_SORT_IMPL"), "lsd") == 0);
        // GS_SORT_MSD_BITS=8|9: the partitioning digit's width whatever the size (measurements)
        static const int forced_bits = getenv("GS_SORT_MSD_BITS") ? atoi(getenv("GS_SORT_MSD_BITS")) : 0;
        const int bits = key_depth_bits + tile_bits;
        // bins in any order: partition by the LOWEST bits of the bin field -- a bucket is then every 256th (512th) bin of
        // the frame, and the buckets are as even as the frame is large; by the top bits a bucket is a run of adjacent bins
        // and follows the density of the scene (headline: 79 of 255 buckets above the LDS capacity)
        const bool grouped = bins_in_any_order && tile_bits >= RADIX_BITS;
        int m = (msd_allowed && bits > RADIX_BITS)
                    ? msd_digit_bits(n_keys, bits, grouped ? GS_SORT_MSD_GROUPED_BUCKET_KEYS : GS_SORT_MSD_BUCKET_KEYS) : 0;
        if (m && forced_bits >= RADIX_BITS && forced_bits <= MSD_MAX_BITS) m = forced_bits < bits ? forced_bits : bits - 1;
        if (grouped && m > tile_bits) m = tile_bits;
        if (m >= RADIX_BITS) {
            const int part_shift = grouped ? key_depth_bits : bits - m;
            uint32_t *k = (uint32_t *)keys, *ka = (uint32_t *)keys_alt;
            // buckets of whole bins (the digit inside the bin field): the bucket-local sort knows every bin's range
            int32_t *ts = part_shift >= key_depth_bits ? tile_start : nullptr, *te = ts ? tile_end : nullptr;
            return m == 8 ? sort_msd_first_rb<8>(k, payload, ka, payload_alt, n_keys, n_keys_device, bits, part_shift,
                                                 allow_result_in_alt, workspace, s, also_zero, also_zero_bytes,
                                                 key_depth_bits, n_tiles, ts, te)
                          : sort_msd_first_rb<9>(k, payload, ka, payload_alt, n_keys, n_keys_device, bits, part_shift,
                                                 allow_result_in_alt, workspace, s, also_zero, also_zero_bytes,
                                                 key_depth_bits, n_tiles, ts, te);
        }
        for (int sh = 0; sh < key_depth_bits + tile_bits; sh += RADIX_BITS) shifts[n_pass++] = sh;
        return sort_pairs_impl<uint32_t>((uint32_t *)keys, payload, (uint32_t *)keys_alt, payload_alt, n_keys,
                                         n_keys_device, shifts, n_pass, 0u, allow_result_in_alt, workspace, s, also_zero,
                                         also_zero_bytes);
    }
    uint64_t flip = 0;
    if (depth_bits >= 64) {  // full signed 64-bit order
        for (int sh = 0; sh < 64; sh += RADIX_BITS) shifts[n_pass++] = sh;
        flip = 0x8000000000000000ull;
    } else {
        for (int sh = 0; sh < depth_bits && sh < 32; sh += RADIX_BITS) shifts[n_pass++] = sh;
        for (int sh = 32; sh < 32 + tile_bits; sh += RADIX_BITS) shifts[n_pass++] = sh;
    }
    return sort_pairs_impl<uint64_t>((uint64_t *)keys, payload, (uint64_t *)keys_alt, payload_alt, n_keys,
                                     n_keys_device, shifts, n_pass, flip, allow_result_in_alt, workspace, s, also_zero,
                                     also_zero_bytes);
}

}  // extern "C"
