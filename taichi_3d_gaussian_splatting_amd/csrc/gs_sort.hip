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
                                                            KeyT flip, int nblk, int32_t *__restrict__ counts,
                                                            int4 *__restrict__ also_zero, long long also_zero_int4) {
    constexpr int SORT_ITEMS = GS_BLOCK * SORT_ROUNDS;
    __shared__ int hist[WAVES][RADIX];
    // a caller's zero-fill rides along with the first launch of the sort (the frame's list ranges: no fill launch of its own)
    for (long long z = (long long)blockIdx.x * GS_BLOCK + threadIdx.x; z < also_zero_int4; z += (long long)gridDim.x * GS_BLOCK)
        also_zero[z] = make_int4(0, 0, 0, 0);
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

}  // namespace

template <typename KeyT, int SORT_ROUNDS>
static int sort_passes(KeyT *keys, int32_t *payload, KeyT *keys_alt, int32_t *payload_alt, int64_t n_keys,
                       const int32_t *n_dev, const int *shifts, int n_pass, KeyT flip, int allow_result_in_alt,
                       void *workspace, hipStream_t s, void *also_zero, size_t also_zero_bytes) {
    if (n_pass == 0 && also_zero_bytes) GS_CHECK_HIP(hipMemsetAsync(also_zero, 0, also_zero_bytes, s));
    const int nblk = gs_div_up(n_keys, GS_BLOCK * SORT_ROUNDS);
    int32_t *counts = (int32_t *)workspace;
    int32_t *totals = counts + (size_t)RADIX * nblk;
    KeyT *kin = keys, *kout = keys_alt;
    int32_t *pin = payload, *pout = payload_alt;
    for (int p = 0; p < n_pass; ++p) {
        hipLaunchKernelGGL((sort_hist_kernel<KeyT, SORT_ROUNDS>), dim3(nblk), dim3(GS_BLOCK), 0, s, kin, (long long)n_keys,
                           n_dev, shifts[p], flip, nblk, counts, (int4 *)also_zero,
                           (long long)(p == 0 ? also_zero_bytes / 16 : 0));
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

template <typename KeyT>
static int sort_pairs_impl(KeyT *keys, int32_t *payload, KeyT *keys_alt, int32_t *payload_alt, int64_t n_keys,
                           const int32_t *n_dev, const int *shifts, int n_pass, KeyT flip, int allow_result_in_alt,
                           void *workspace, hipStream_t s, void *also_zero, size_t also_zero_bytes) {
    // (n_keys is the capacity when the count lives on the device: the choice follows the capacity, as the grids do)
    if (sort_rounds_for(n_keys) == GS_SORT_SMALL_ROUNDS)
        return sort_passes<KeyT, GS_SORT_SMALL_ROUNDS>(keys, payload, keys_alt, payload_alt, n_keys, n_dev, shifts, n_pass,
                                                       flip, allow_result_in_alt, workspace, s, also_zero, also_zero_bytes);
    return sort_passes<KeyT, GS_SORT_ROUNDS>(keys, payload, keys_alt, payload_alt, n_keys, n_dev, shifts, n_pass, flip,
                                             allow_result_in_alt, workspace, s, also_zero, also_zero_bytes);
}

extern "C" {

size_t gs_sort_workspace_bytes(int64_t n_keys) {
    const size_t nblk = (size_t)gs_div_up(n_keys > 0 ? n_keys : 1, GS_BLOCK * sort_rounds_for(n_keys));
    return sizeof(int32_t) * (RADIX * nblk + RADIX + 64);
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
