// gs_shard.hip -- multi-GPU (tile-row sharded) backward: sparse exchange of the per-Gaussian accumulators.
//
// The reference is single-GPU.  Under tile-row sharding (distributed.py) every rank back-propagates its own tiles and
// holds PARTIAL accumulator records acc[M,12]; the per-point pass needs their sum over ranks.  Round 2 summed the dense
// [M,12] array with one all-reduce (47 MB at the headline size -- as expensive as what sharding saves).  With bands a
// Gaussian is blended by one rank (two when it straddles a boundary), so all but ~1/G of a rank's rows are zero: here
// each rank sends only the rows it produced (num_keys > 0), as a list of (row id, 48-B record) in ascending id order,
// the lists are all-gathered, and every rank adds them up in RANK ORDER -- the same additions in the same order on every
// rank, so the replicated gradients stay bit-identical across ranks and from run to run.
//   gs_compact_rows : acc + num_keys -> ascending list of produced rows (ordered compaction: ballot + mbcnt, block
//                     offsets by summing the L2-resident counts of the blocks before, as gs_filter_compact)
//   gs_merge_rows   : G gathered lists -> dense acc[M,12]; a workgroup owns 256 consecutive row ids, finds its span in
//                     every (sorted) list by binary search and adds the lists in rank order in LDS
#include "gs_common.h"

namespace {

constexpr int ROWS_PER_BLOCK = GS_BLOCK * 4;   // rows examined per workgroup of the compaction (4 rounds of 256)

__global__ __launch_bounds__(GS_BLOCK) void rows_count_kernel(const int32_t *__restrict__ num_keys, int m,
                                                            int32_t *__restrict__ block_counts) {
    __shared__ int s_count;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    int local = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = blockIdx.x * ROWS_PER_BLOCK + r * GS_BLOCK + threadIdx.x;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(i < m && num_keys[i] > 0);
        if (gs_lane() == 0) local += __popcll(b);
    }
    if (gs_lane() == 0) atomicAdd(&s_count, local);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_count;
}

__global__ __launch_bounds__(GS_BLOCK) void rows_compact_kernel(const int32_t *__restrict__ num_keys,
                                                              const float4 *__restrict__ acc, int m,
                                                              const int32_t *__restrict__ block_counts, int capacity,
                                                              int32_t *__restrict__ ids, float4 *__restrict__ rows,
                                                              int32_t *__restrict__ total_out) {
    __shared__ int s_wave[GS_BLOCK / GS_WAVE], s_before[GS_BLOCK / GS_WAVE];
    const int w = threadIdx.x >> 6;
    {
        int part = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += GS_BLOCK) part += block_counts[b];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, GS_WAVE);
        if (gs_lane() == 0) s_before[w] = part;
        __syncthreads();
    }
    int running = 0;
#pragma unroll
    for (int k = 0; k < GS_BLOCK / GS_WAVE; ++k) running += s_before[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = blockIdx.x * ROWS_PER_BLOCK + r * GS_BLOCK + threadIdx.x;
        const bool keep = i < m && num_keys[i] > 0;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(keep);
        const int rank = gs_mbcnt(b);
        if (gs_lane() == 0) s_wave[w] = __popcll(b);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < GS_BLOCK / GS_WAVE; ++k) {
            const int c = s_wave[k];
            if (k < w) before += c;
            total += c;
        }
        const int dst = running + before + rank;
        if (keep && dst < capacity) {
            ids[dst] = i;
            rows[3 * (size_t)dst] = acc[3 * (size_t)i];
            rows[3 * (size_t)dst + 1] = acc[3 * (size_t)i + 1];
            rows[3 * (size_t)dst + 2] = acc[3 * (size_t)i + 2];
        }
        running += total;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = running;
}

// first index in the ascending array a[0, n) whose value is >= x
__device__ __forceinline__ int lower_bound(const int32_t *__restrict__ a, int n, int x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int MERGE_IDS = GS_BLOCK;   // row ids owned by one workgroup
constexpr int MERGE_MAX_WORLD = GS_BLOCK / 2;   // one thread per (list, span end) in the search phase
constexpr int MERGE_LISTS_IN_FLIGHT = 4;        // lists whose rows a thread holds in registers at a time
// lists: world blocks of `stride` 32-bit words each: [capacity] ids, then [capacity][12] rows; counts[g] = valid entries.
// A workgroup first finds its span in ALL lists at once (2 x world threads, one binary search each: one chain of ~log2(n)
// dependent loads per workgroup instead of one per list -- with the searches inside the list loop the launch took 99 us at
// G = 8, M = 1e6, four times what its 120 MB of traffic cost), then adds the lists in rank order, the rows of four lists in
// flight at a time.
__global__ __launch_bounds__(GS_BLOCK) void rows_merge_kernel(const int32_t *__restrict__ lists, long long stride,
                                                            int capacity, const int32_t *__restrict__ counts, int world,
                                                            int m, float4 *__restrict__ acc) {
    __shared__ float s_acc[MERGE_IDS][GS_ACC_STRIDE + 1];   // (+1: rows on different banks)
    __shared__ int s_npix[MERGE_IDS];
    __shared__ int s_span[2 * MERGE_MAX_WORLD];
    const int lo_id = blockIdx.x * MERGE_IDS, hi_id = min(lo_id + MERGE_IDS, m);
#pragma unroll
    for (int k = 0; k < GS_ACC_STRIDE; ++k) s_acc[threadIdx.x][k] = 0.f;
    s_npix[threadIdx.x] = 0;
    if ((int)threadIdx.x < 2 * world) {
        const int g = threadIdx.x >> 1;
        s_span[threadIdx.x] = lower_bound(lists + (size_t)g * stride, min(counts[g], capacity),
                                          (threadIdx.x & 1) ? hi_id : lo_id);
    }
    __syncthreads();
    for (int g0 = 0; g0 < world; g0 += MERGE_LISTS_IN_FLIGHT) {
        int row[MERGE_LISTS_IN_FLIGHT];
        float4 a[MERGE_LISTS_IN_FLIGHT], b[MERGE_LISTS_IN_FLIGHT], c[MERGE_LISTS_IN_FLIGHT];
#pragma unroll
        for (int k = 0; k < MERGE_LISTS_IN_FLIGHT; ++k) {
            const int g = g0 + k;
            row[k] = -1;
            if (g < world) {
                const int32_t *ids = lists + (size_t)g * stride;
                const float4 *rows = reinterpret_cast<const float4 *>(ids + capacity);
                const int first = s_span[2 * g], cnt = s_span[2 * g + 1] - first;   // <= MERGE_IDS: ids are distinct within a list
                if ((int)threadIdx.x < cnt) {
                    const int j = first + threadIdx.x;
                    row[k] = ids[j] - lo_id;
                    a[k] = rows[3 * (size_t)j]; b[k] = rows[3 * (size_t)j + 1]; c[k] = rows[3 * (size_t)j + 2];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MERGE_LISTS_IN_FLIGHT; ++k) {   // rank order: the same additions in the same order on every rank
            if (row[k] >= 0) {
                float *d = s_acc[row[k]];   // one thread per row of this list: no two threads touch the same row
                d[0] += a[k].x; d[1] += a[k].y; d[2] += a[k].z; d[3] += a[k].w;
                d[4] += b[k].x; d[5] += b[k].y; d[6] += b[k].z; d[7] += b[k].w;
                d[8] += c[k].x; d[9] += c[k].y;
                s_npix[row[k]] += __builtin_bit_cast(int, c[k].z);   // pixel count: int32 bits, summed as an integer
            }
            __syncthreads();                // this list fully added before another thread adds the next one into the same rows
        }
    }
    const int i = lo_id + threadIdx.x;
    if (i < hi_id) {
        const float *d = s_acc[threadIdx.x];
        acc[3 * (size_t)i] = make_float4(d[0], d[1], d[2], d[3]);
        acc[3 * (size_t)i + 1] = make_float4(d[4], d[5], d[6], d[7]);
        acc[3 * (size_t)i + 2] = make_float4(d[8], d[9], __builtin_bit_cast(float, s_npix[threadIdx.x]), 0.f);
    }
}

}  // namespace

extern "C" {

size_t gs_compact_rows_workspace_bytes(int n_visible) {
    return sizeof(int32_t) * ((size_t)gs_div_up(n_visible > 0 ? n_visible : 1, ROWS_PER_BLOCK) + 64);
}

int gs_compact_rows(const float *acc, const int32_t *num_keys, int n_visible, int capacity, int32_t *ids, float *rows,
                    int32_t *count, void *workspace, void *stream) {
    GS_REQUIRE(n_visible >= 0 && capacity >= 0, "sizes");
    hipStream_t s = (hipStream_t)stream;
    if (n_visible == 0) {
        GS_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
        return 0;
    }
    int32_t *block_counts = (int32_t *)workspace;
    const int nblk = gs_div_up(n_visible, ROWS_PER_BLOCK);
    hipLaunchKernelGGL(rows_count_kernel, dim3(nblk), dim3(GS_BLOCK), 0, s, num_keys, n_visible, block_counts);
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(rows_compact_kernel, dim3(nblk), dim3(GS_BLOCK), 0, s, num_keys,
                       reinterpret_cast<const float4 *>(acc), n_visible, block_counts, capacity, ids,
                       reinterpret_cast<float4 *>(rows), count);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_merge_rows(const int32_t *lists, int64_t list_stride_words, int capacity, const int32_t *counts, int world,
                  int n_visible, float *acc, void *stream) {
    GS_REQUIRE(world >= 1 && world <= MERGE_MAX_WORLD && n_visible >= 0 && capacity >= 0, "sizes (world <= 128)");
    GS_REQUIRE(list_stride_words >= 13LL * capacity && capacity % 4 == 0,
               "a list holds `capacity` ids followed by `capacity` 48-B rows; capacity must be a multiple of 4 (16-B rows)");
    if (n_visible == 0) return 0;
    hipLaunchKernelGGL(rows_merge_kernel, dim3(gs_div_up(n_visible, MERGE_IDS)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       lists, (long long)list_stride_words, capacity, counts, world, n_visible,
                       reinterpret_cast<float4 *>(acc));
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
