// MEASUREMENT RECORD (round 4), not part of the build: gs_sort.hip with the single-sweep passes, chained look-back variant.
// Results: profiles/r04_sort.md.  Drop it over csrc/gs_sort.hip to rebuild the arm (GS_SORT_IMPL=lsd3 selects the three-launch sort).
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
constexpr int RADIX = 1 << RADIX_BITS;
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

template <typename KeyT>
__device__ __forceinline__ unsigned digit_of(KeyT key, int shift, KeyT flip) {
    return (unsigned)(((key ^ flip) >> shift) & (RADIX - 1));
}

template <typename KeyT, int SORT_ROUNDS>
__global__ __launch_bounds__(GS_BLOCK) void sort_hist_kernel(const KeyT *__restrict__ keys, long long n,
                                                            const int32_t *__restrict__ n_device, int shift,
                                                            KeyT flip, int nblk, int32_t *__restrict__ counts) {
    constexpr int SORT_ITEMS = GS_BLOCK * SORT_ROUNDS;
    __shared__ int hist[WAVES][RADIX];
    if (n_device) n = min((long long)*n_device, n);   // grid and workspace are sized by the capacity n   // one histogram per wave: a quarter of the same-address LDS atomics
#pragma unroll
    for (int k = 0; k < WAVES; ++k) hist[k][threadIdx.x] = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    const long long base = (long long)blockIdx.x * SORT_ITEMS;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        long long i = base + r * GS_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&hist[w][digit_of<KeyT>(keys[i], shift, flip)], 1);
    }
    __syncthreads();
    int c = 0;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) c += hist[k][threadIdx.x];
    counts[(size_t)threadIdx.x * nblk + blockIdx.x] = c;
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
template <typename KeyT, int SORT_ROUNDS>
__global__ __launch_bounds__(GS_BLOCK) void sort_scatter_kernel(
    const KeyT *__restrict__ keys_in, const int32_t *__restrict__ payload_in, long long n,
    const int32_t *__restrict__ n_device, int shift, KeyT flip, int nblk, const int32_t *__restrict__ row_offsets,
    const int32_t *__restrict__ totals, KeyT *__restrict__ keys_out, int32_t *__restrict__ payload_out) {
    constexpr int SORT_ITEMS = GS_BLOCK * SORT_ROUNDS;
    if (n_device) n = min((long long)*n_device, n);
    __shared__ int s_cnt[WAVES][RADIX];   // running per-wave digit counts, later exclusive prefix over waves
    __shared__ int s_local[RADIX];        // block-local start of each digit's run
    __shared__ int s_gbase[RADIX];        // global position of the block's first key of each digit
    __shared__ KeyT s_keys[SORT_ITEMS];
    __shared__ int32_t s_pay[SORT_ITEMS];
    __shared__ int lds[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    {
        int total;
        const int ex = gs_block_excl_scan(totals[threadIdx.x], &total, lds);
        s_gbase[threadIdx.x] = ex + row_offsets[(size_t)threadIdx.x * nblk + blockIdx.x];
    }
#pragma unroll
    for (int k = 0; k < WAVES; ++k) s_cnt[k][threadIdx.x] = 0;
    __syncthreads();
    const long long block_base = (long long)blockIdx.x * SORT_ITEMS;
    const long long wave_base = block_base + (long long)w * (SORT_ITEMS / WAVES);
    volatile int *cnt = &s_cnt[w][0];   // volatile: LDS accesses of a wave stay in program order
    KeyT key[SORT_ROUNDS];
    int32_t pay[SORT_ROUNDS];
    int rnk[SORT_ROUNDS];   // rank among the same-digit keys of this wave; -1 = past the end of the array
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const long long i = wave_base + r * GS_WAVE + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : (KeyT)0;
        pay[r] = valid ? payload_in[i] : 0;
        const unsigned d = digit_of<KeyT>(key[r], shift, flip);
        unsigned long long peers = __ballot(valid);   // 64-lane match-any on the digit
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const int rank = gs_mbcnt(peers);
        const int before = valid ? cnt[d] : 0;                       // every lane of a group reads ...
        if (valid && rank == 0) cnt[d] = before + __popcll(peers);   // ... before its leader bumps the counter
        rnk[r] = valid ? before + rank : -1;
    }
    __syncthreads();
    {
        int run = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) {
            const int c = s_cnt[k][threadIdx.x];
            s_cnt[k][threadIdx.x] = run;
            run += c;
        }
        int total;
        s_local[threadIdx.x] = gs_block_excl_scan(run, &total, lds);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        if (rnk[r] >= 0) {
            const unsigned d = digit_of<KeyT>(key[r], shift, flip);
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
        const unsigned d = digit_of<KeyT>(k, shift, flip);
        const int dst = s_gbase[d] + (p - s_local[d]);
        keys_out[dst] = k;
        payload_out[dst] = s_pay[p];
    }
}


// =================================================================== single-sweep passes (round 4)
// One launch per digit instead of three (histogram / row scan / scatter), keys read once per pass instead of twice:
//   sort_digit_totals_kernel  ONE read of the keys gives the digit histograms of ALL passes (a digit's histogram does
//                             not depend on the order of the keys), as per-workgroup partial rows -- plain stores, no
//                             global atomics (a same-address device atomic costs ~0.25 us here, profiles/
//                             r03_ubench_atomics_lds_sort.txt); it also zeroes the look-back status words.
//   sort_sweep_kernel         one 1024-thread workgroup per tile of 1024 x R consecutive keys.  Tiles are handed out by a
//                             ticket (one returning atomic per workgroup), so a tile only ever waits for tiles whose
//                             workgroups are already running: no assumption about dispatch order or residency.  The
//                             workgroup ranks its keys by the pass's digit
//                             in LDS (per-wave running counters + 64-lane match-any, as before), publishes its 256
//                             digit counts, obtains the counts of all EARLIER tiles by decoupled look-back (one thread
//                             per digit walks back over the predecessors' published {state, value} words, eight loads
//                             in flight, until it meets a tile that has published its inclusive prefix), publishes its
//                             own inclusive prefix, and writes each digit's run to its final position.
// A status word is one 8-byte granule {epoch * 4 + state, value} written by a single agent-scope (sc1) store and polled
// with relaxed agent-scope loads: the data is the flag, no fences (cdna_hip_programming.md, Guideline 16 R2); epoch =
// pass + 1, so the words of an earlier pass read as "not yet published" and one array serves all passes.
constexpr int SW_THREADS = 1024;
constexpr int SW_WAVES = SW_THREADS / GS_WAVE;
constexpr int SW_MAX_PASSES = 8;
constexpr int SW_TOTALS_BLOCKS = 128;   // workgroups of sort_digit_totals_kernel = partial rows every sweep workgroup sums
constexpr int SW_LOOKBACK_WINDOW = 8;   // predecessor status words in flight per look-back step
constexpr unsigned SW_STATE_AGGREGATE = 1u, SW_STATE_PREFIX = 2u;
struct SweepShifts { int shift[SW_MAX_PASSES]; int n_pass; };
typedef unsigned long long sw_u64;

__device__ __forceinline__ void sw_publish(sw_u64 *word, unsigned epoch, unsigned state, unsigned value) {
    __hip_atomic_store(word, ((sw_u64)(epoch * 4u + state) << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ sw_u64 sw_peek(const sw_u64 *word) {
    return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename KeyT>
__global__ __launch_bounds__(SW_THREADS) void sort_digit_totals_kernel(
    const KeyT *__restrict__ keys, long long n, const int32_t *__restrict__ n_device, SweepShifts sh, KeyT flip,
    int32_t *__restrict__ partial /* [gridDim.x][n_pass][RADIX] */, sw_u64 *__restrict__ status, long long status_words,
    int4 *__restrict__ also_zero, long long also_zero_int4) {
    __shared__ int s_hist[4][SW_MAX_PASSES][RADIX];   // four copies: a quarter of the same-address LDS atomics
    if (n_device) n = min((long long)*n_device, n);
    for (int k = threadIdx.x; k < 4 * SW_MAX_PASSES * RADIX; k += SW_THREADS) (&s_hist[0][0][0])[k] = 0;
    __syncthreads();
    const int copy = (threadIdx.x >> 6) & 3;
    const long long stride = (long long)gridDim.x * SW_THREADS;
    for (long long i = (long long)blockIdx.x * SW_THREADS + threadIdx.x; i < n; i += 4 * stride) {
        KeyT k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = i + j * stride < n ? keys[i + j * stride] : (KeyT)0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i + j * stride < n)
                for (int p = 0; p < sh.n_pass; ++p) atomicAdd(&s_hist[copy][p][digit_of<KeyT>(k[j], sh.shift[p], flip)], 1);
    }
    // the frame's zero-fills ride along (look-back status words; optionally a caller's buffer, e.g. the list ranges)
    for (long long w = (long long)blockIdx.x * SW_THREADS + threadIdx.x; w < status_words; w += stride) status[w] = 0ull;
    for (long long w = (long long)blockIdx.x * SW_THREADS + threadIdx.x; w < also_zero_int4; w += stride)
        also_zero[w] = make_int4(0, 0, 0, 0);
    __syncthreads();
    for (int k = threadIdx.x; k < sh.n_pass * RADIX; k += SW_THREADS) {
        const int p = k / RADIX, d = k % RADIX;
        partial[((size_t)blockIdx.x * sh.n_pass + p) * RADIX + d] =
            (s_hist[0][p][d] + s_hist[1][p][d]) + (s_hist[2][p][d] + s_hist[3][p][d]);
    }
}

// exclusive scan of one value per digit (threads 0 .. RADIX-1 hold it, the others pass 0) over the whole workgroup;
// lds: SW_WAVES ints.  Two barriers.
__device__ __forceinline__ int sw_scan_digits(int v, int *lds) {
    const int w = threadIdx.x >> 6;
    const int incl = gs_wave_incl_scan(v);
    if (gs_lane() == GS_WAVE - 1) lds[w] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < RADIX / GS_WAVE; ++k) base += k < w ? lds[k] : 0;
    __syncthreads();
    return base + incl - v;
}

template <typename KeyT, int R>
__global__ __launch_bounds__(SW_THREADS) void sort_sweep_kernel(
    const KeyT *__restrict__ keys_in, const int32_t *__restrict__ payload_in, long long n,
    const int32_t *__restrict__ n_device, int shift, KeyT flip, int pass, int n_pass,
    const int32_t *__restrict__ partial, int partial_rows, sw_u64 *__restrict__ status_base, int use_ticket,
    KeyT *__restrict__ keys_out, int32_t *__restrict__ payload_out) {
    constexpr int TILE = SW_THREADS * R;
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_lds[];
    KeyT *s_keys = reinterpret_cast<KeyT *>(sw_lds);
    int32_t *s_pay = reinterpret_cast<int32_t *>(s_keys + TILE);
    int (*s_cnt)[RADIX] = reinterpret_cast<int (*)[RADIX]>(s_pay + TILE);   // [SW_WAVES][RADIX]
    int *s_local = &s_cnt[SW_WAVES][0];     // tile-local start of each digit's run
    int *s_gbase = s_local + RADIX;         // global position of the tile's first key of each digit
    int *s_dbase = s_gbase + RADIX;         // global position of the first key of each digit (all tiles)
    int *s_part = s_dbase + RADIX;          // [4][RADIX] partial sums of the digit totals
    int *s_misc = s_part + 4 * RADIX;       // [SW_WAVES]
    if (n_device) n = min((long long)*n_device, n);
    const int n_tiles = (int)((n + TILE - 1) / TILE);
    // status_base: SW_MAX_PASSES ticket words (one per pass), then RADIX status words per tile
    sw_u64 *status = status_base + SW_MAX_PASSES;
    int tile = blockIdx.x;
    if (use_ticket) {
        if (threadIdx.x == 0)
            s_misc[0] = (int)__hip_atomic_fetch_add(&status_base[pass], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        tile = s_misc[0];
        __syncthreads();
    }
    if (tile >= n_tiles) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned epoch = (unsigned)pass + 1u;
    {   // digit totals of this pass: sum of the partial rows of sort_digit_totals_kernel (L2-resident), then their scan
        const int d = threadIdx.x & (RADIX - 1), j = threadIdx.x >> RADIX_BITS;
        int sum = 0;
        for (int r = j; r < partial_rows; r += SW_THREADS / RADIX)
            sum += partial[((size_t)r * n_pass + pass) * RADIX + d];
        s_part[j * RADIX + d] = sum;
        __syncthreads();
        const int total = threadIdx.x < RADIX ? (s_part[d] + s_part[RADIX + d]) + (s_part[2 * RADIX + d] + s_part[3 * RADIX + d]) : 0;
        const int ex = sw_scan_digits(total, s_misc);
        if (threadIdx.x < RADIX) s_dbase[threadIdx.x] = ex;
    }
    {
#pragma unroll
        for (int k = 0; k < SW_WAVES * RADIX / SW_THREADS; ++k) (&s_cnt[0][0])[k * SW_THREADS + threadIdx.x] = 0;
        __syncthreads();
        const long long tile_base = (long long)tile * TILE;
        const long long wave_base = tile_base + (long long)w * (R * GS_WAVE);
        volatile int *cnt = &s_cnt[w][0];   // volatile: LDS accesses of a wave stay in program order
        KeyT key[R];
        int32_t pay[R];
        int rnk[R];   // rank among the same-digit keys of this wave; -1 = past the end of the array
#pragma unroll
        for (int r = 0; r < R; ++r) {   // all loads of the tile in flight before the first ballot
            const long long i = wave_base + r * GS_WAVE + lane;
            key[r] = i < n ? keys_in[i] : (KeyT)0;
            pay[r] = i < n ? payload_in[i] : 0;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool valid = wave_base + r * GS_WAVE + lane < n;
            const unsigned d = digit_of<KeyT>(key[r], shift, flip);
            unsigned long long peers = __ballot(valid);   // 64-lane match-any on the digit
#pragma unroll
            for (int b = 0; b < RADIX_BITS; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const int rank = gs_mbcnt(peers);
            const int before = valid ? cnt[d] : 0;                       // every lane of a group reads ...
            if (valid && rank == 0) cnt[d] = before + __popcll(peers);   // ... before its leader bumps the counter
            rnk[r] = valid ? before + rank : -1;
        }
        __syncthreads();
        int count = 0;
        if (threadIdx.x < RADIX) {   // per-wave counts -> exclusive prefix over the waves; the tile's digit counts
#pragma unroll
            for (int k = 0; k < SW_WAVES; ++k) {
                const int c = s_cnt[k][threadIdx.x];
                s_cnt[k][threadIdx.x] = count;
                count += c;
            }
            // published at once: the successors' look-back can add it while this tile is still permuting
            sw_publish(&status[(size_t)tile * RADIX + threadIdx.x], epoch, tile == 0 ? SW_STATE_PREFIX : SW_STATE_AGGREGATE,
                       (unsigned)count);
        }
        const int local = sw_scan_digits(count, s_misc);
        if (threadIdx.x < RADIX) s_local[threadIdx.x] = local;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (rnk[r] >= 0) {
                const unsigned d = digit_of<KeyT>(key[r], shift, flip);
                const int pos = s_local[d] + s_cnt[w][d] + rnk[r];
                s_keys[pos] = key[r];
                s_pay[pos] = pay[r];
            }
        }
        if (threadIdx.x < RADIX) {   // decoupled look-back: keys of digit d in all earlier tiles
            unsigned before = 0;
            int p = tile - 1;
            unsigned spins = 0;
            while (p >= 0) {
                sw_u64 st[SW_LOOKBACK_WINDOW];
#pragma unroll
                for (int j = 0; j < SW_LOOKBACK_WINDOW; ++j)
                    st[j] = p - j >= 0 ? sw_peek(&status[(size_t)(p - j) * RADIX + threadIdx.x]) : 0ull;
                int used = 0;
                bool done = false;
#pragma unroll
                for (int j = 0; j < SW_LOOKBACK_WINDOW; ++j) {
                    const unsigned tag = (unsigned)(st[j] >> 32);
                    const bool ready = p - j >= 0 && (tag >> 2) == epoch && !done && used == j;
                    if (ready) {
                        before += (unsigned)st[j];
                        ++used;
                        done = (tag & 3u) == SW_STATE_PREFIX;
                    }
                }
                p = done ? -1 : p - used;
                if (!done && used == 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 24)) break;   // a lost predecessor must not hang the GPU (the output is then wrong)
                }
            }
            if (tile > 0) sw_publish(&status[(size_t)tile * RADIX + threadIdx.x], epoch, SW_STATE_PREFIX, before + (unsigned)count);
            s_gbase[threadIdx.x] = s_dbase[threadIdx.x] + (int)before;
        }
        __syncthreads();
        const long long left = n - tile_base;
        const int nb = left < TILE ? (int)left : TILE;
        for (int p = threadIdx.x; p < nb; p += SW_THREADS) {
            const KeyT k = s_keys[p];
            const unsigned d = digit_of<KeyT>(k, shift, flip);
            const int dst = s_gbase[d] + (p - s_local[d]);
            keys_out[dst] = k;
            payload_out[dst] = s_pay[p];
        }
    }
}

template <typename KeyT, int R>
constexpr size_t sweep_lds_bytes() {
    return (size_t)SW_THREADS * R * (sizeof(KeyT) + sizeof(int32_t)) + sizeof(int) * ((SW_WAVES + 7) * RADIX + SW_WAVES);
}

}  // namespace

template <typename KeyT, int SORT_ROUNDS>
static int sort_passes(KeyT *keys, int32_t *payload, KeyT *keys_alt, int32_t *payload_alt, int64_t n_keys,
                       const int32_t *n_dev, const int *shifts, int n_pass, KeyT flip, int allow_result_in_alt,
                       void *workspace, hipStream_t s) {
    const int nblk = gs_div_up(n_keys, GS_BLOCK * SORT_ROUNDS);
    int32_t *counts = (int32_t *)workspace;
    int32_t *totals = counts + (size_t)RADIX * nblk;
    KeyT *kin = keys, *kout = keys_alt;
    int32_t *pin = payload, *pout = payload_alt;
    for (int p = 0; p < n_pass; ++p) {
        hipLaunchKernelGGL((sort_hist_kernel<KeyT, SORT_ROUNDS>), dim3(nblk), dim3(GS_BLOCK), 0, s, kin, (long long)n_keys,
                           n_dev, shifts[p], flip, nblk, counts);
        GS_CHECK_LAUNCH();
        hipLaunchKernelGGL(sort_scan_rows_kernel, dim3(RADIX), dim3(GS_BLOCK), 0, s, counts, nblk, totals);
        GS_CHECK_LAUNCH();
        hipLaunchKernelGGL((sort_scatter_kernel<KeyT, SORT_ROUNDS>), dim3(nblk), dim3(GS_BLOCK), 0, s, kin, pin,
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


// ------------------------------------------------------------------ single-sweep passes: host side
static int sweep_rounds_for(int64_t n_keys, int max_rounds) {
    static const int forced = getenv("GS_SWEEP_ROUNDS") ? atoi(getenv("GS_SWEEP_ROUNDS")) : 0;   // tuning knob
    if (forced > 0) return forced > max_rounds ? max_rounds : forced;
    // the smallest tile that still gives at most one tile per CU (256): small inputs spread over many CUs, the look-back
    // chain stays short; beyond 256 x the largest tile the launch runs in several waves of workgroups
    const int choices[] = {1, 2, 4, 8, 11};
    for (int r : choices)
        if (r <= max_rounds && gs_div_up(n_keys > 0 ? n_keys : 1, (long long)SW_THREADS * r) <= 256) return r;
    return max_rounds;
}
template <typename KeyT> constexpr int sweep_max_rounds() { return sizeof(KeyT) == 4 ? 11 : 8; }

template <typename KeyT, int R>
static int sweep_launch(const KeyT *kin, const int32_t *pin, int64_t n_keys, const int32_t *n_dev, int shift, KeyT flip,
                        int pass, int n_pass, const int32_t *partial, int partial_rows, sw_u64 *status, int use_ticket,
                        KeyT *kout, int32_t *pout, hipStream_t s) {
    constexpr size_t lds = sweep_lds_bytes<KeyT, R>();
    static bool configured[64] = {};   // per device: the kernel may use more than the default 64 KB of dynamic LDS
    int dev = 0;
    GS_CHECK_HIP(hipGetDevice(&dev));
    if (lds > 48 * 1024 && !configured[dev & 63]) {
        GS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&sort_sweep_kernel<KeyT, R>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured[dev & 63] = true;
    }
    const int n_tiles = gs_div_up(n_keys, (long long)SW_THREADS * R);
    hipLaunchKernelGGL((sort_sweep_kernel<KeyT, R>), dim3(n_tiles), dim3(SW_THREADS), lds, s, kin, pin, (long long)n_keys,
                       n_dev, shift, flip, pass, n_pass, partial, partial_rows, status, use_ticket, kout, pout);
    GS_CHECK_LAUNCH();
    return 0;
}

static size_t sweep_workspace_bytes(int64_t n_keys) {
    const size_t tiles = (size_t)gs_div_up(n_keys > 0 ? n_keys : 1, (long long)SW_THREADS * 8) + 256;   // (>= any choice)
    return sizeof(int32_t) * (size_t)SW_TOTALS_BLOCKS * SW_MAX_PASSES * RADIX + sizeof(sw_u64) * (SW_MAX_PASSES + tiles * RADIX) + 64;
}

template <typename KeyT>
static int sort_sweep_impl(KeyT *keys, int32_t *payload, KeyT *keys_alt, int32_t *payload_alt, int64_t n_keys,
                           const int32_t *n_dev, const int *shifts, int n_pass, KeyT flip, int allow_result_in_alt,
                           void *workspace, hipStream_t s, void *also_zero, size_t also_zero_bytes) {
    GS_REQUIRE(n_pass <= SW_MAX_PASSES, "too many radix passes");
    GS_REQUIRE(also_zero_bytes % 16 == 0 && ((uintptr_t)also_zero & 15) == 0, "also_zero must be 16-byte aligned");
    static const int use_ticket = getenv("GS_SWEEP_TICKET") ? atoi(getenv("GS_SWEEP_TICKET")) : 1;
    const int rounds = sweep_rounds_for(n_keys, sweep_max_rounds<KeyT>());
    const int n_tiles = gs_div_up(n_keys, (long long)SW_THREADS * rounds);
    int32_t *partial = (int32_t *)workspace;
    sw_u64 *status = (sw_u64 *)(((uintptr_t)(partial + (size_t)SW_TOTALS_BLOCKS * SW_MAX_PASSES * RADIX) + 15) & ~(uintptr_t)15);
    SweepShifts sh;
    sh.n_pass = n_pass;
    for (int p = 0; p < n_pass; ++p) sh.shift[p] = shifts[p];
    const int rows = (int)(gs_div_up(n_keys, SW_THREADS) < SW_TOTALS_BLOCKS ? gs_div_up(n_keys, SW_THREADS) : SW_TOTALS_BLOCKS);
    hipLaunchKernelGGL(sort_digit_totals_kernel<KeyT>, dim3(rows), dim3(SW_THREADS), 0, s, keys, (long long)n_keys, n_dev,
                       sh, flip, partial, status, (long long)(SW_MAX_PASSES + (size_t)n_tiles * RADIX), (int4 *)also_zero,
                       (long long)(also_zero_bytes / 16));
    GS_CHECK_LAUNCH();
    KeyT *kin = keys, *kout = keys_alt;
    int32_t *pin = payload, *pout = payload_alt;
    for (int p = 0; p < n_pass; ++p) {
        int rc;
#define GS_SWEEP_CASE(RR)                                                                                            \
    case RR:                                                                                                          \
        rc = sweep_launch<KeyT, RR>(kin, pin, n_keys, n_dev, shifts[p], flip, p, n_pass, partial, rows, status,       \
                                    use_ticket, kout, pout, s);                                                       \
        break;
        switch (rounds) {
            GS_SWEEP_CASE(1)
            GS_SWEEP_CASE(2)
            GS_SWEEP_CASE(4)
            GS_SWEEP_CASE(8)
            default:
                if constexpr (sizeof(KeyT) == 4)
                    rc = sweep_launch<KeyT, 11>(kin, pin, n_keys, n_dev, shifts[p], flip, p, n_pass, partial, rows, status,
                                                use_ticket, kout, pout, s);
                else
                    rc = sweep_launch<KeyT, 8>(kin, pin, n_keys, n_dev, shifts[p], flip, p, n_pass, partial, rows, status,
                                               use_ticket, kout, pout, s);
        }
#undef GS_SWEEP_CASE
        if (rc != 0) return rc;
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

template <typename KeyT>
static int sort_pairs_impl(KeyT *keys, int32_t *payload, KeyT *keys_alt, int32_t *payload_alt, int64_t n_keys,
                           const int32_t *n_dev, const int *shifts, int n_pass, KeyT flip, int allow_result_in_alt,
                           void *workspace, hipStream_t s, void *also_zero, size_t also_zero_bytes) {
    // GS_SORT_IMPL=lsd3: the three-launches-per-pass sort of rounds 1-3 (measurement arm)
    static const bool lsd3 = getenv("GS_SORT_IMPL") && !strcmp(getenv("GS_SORT_IMPL"), "lsd3");
    if (!lsd3)
        return sort_sweep_impl<KeyT>(keys, payload, keys_alt, payload_alt, n_keys, n_dev, shifts, n_pass, flip,
                                     allow_result_in_alt, workspace, s, also_zero, also_zero_bytes);
    if (also_zero_bytes) GS_CHECK_HIP(hipMemsetAsync(also_zero, 0, also_zero_bytes, s));
    // (n_keys is the capacity when the count lives on the device: the choice follows the capacity, as the grids do)
    if (sort_rounds_for(n_keys) == GS_SORT_SMALL_ROUNDS)
        return sort_passes<KeyT, GS_SORT_SMALL_ROUNDS>(keys, payload, keys_alt, payload_alt, n_keys, n_dev, shifts, n_pass,
                                                       flip, allow_result_in_alt, workspace, s);
    return sort_passes<KeyT, GS_SORT_ROUNDS>(keys, payload, keys_alt, payload_alt, n_keys, n_dev, shifts, n_pass, flip,
                                             allow_result_in_alt, workspace, s);
}

extern "C" {

size_t gs_sort_workspace_bytes(int64_t n_keys) {
    const size_t nblk = (size_t)gs_div_up(n_keys > 0 ? n_keys : 1, GS_BLOCK * sort_rounds_for(n_keys));
    const size_t lsd3 = sizeof(int32_t) * (RADIX * nblk + RADIX + 64), sweep = sweep_workspace_bytes(n_keys);
    return lsd3 > sweep ? lsd3 : sweep;
}

int gs_sort_pairs(void *keys, int32_t *payload, void *keys_alt, int32_t *payload_alt, int64_t n_keys,
                  const int32_t *n_keys_device, int key_depth_bits, int depth_bits, int tile_bits,
                  int allow_result_in_alt, void *workspace, void *stream) {
    return gs_sort_pairs_and_zero(keys, payload, keys_alt, payload_alt, n_keys, n_keys_device, key_depth_bits, depth_bits,
                                  tile_bits, allow_result_in_alt, workspace, nullptr, 0, stream);
}

int gs_sort_pairs_and_zero(void *keys, int32_t *payload, void *keys_alt, int32_t *payload_alt, int64_t n_keys,
                           const int32_t *n_keys_device, int key_depth_bits, int depth_bits, int tile_bits,
                           int allow_result_in_alt, void *workspace, void *also_zero, size_t also_zero_bytes,
                           void *stream) {
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
