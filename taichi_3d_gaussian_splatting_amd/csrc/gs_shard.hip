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


// =================================================================== owner-sharded Gaussians: routed exchange (round 4)
// Tile-row bands shard the pixels; with a replicated point cloud every rank still projects all N points and back-propagates
// all M visible ones.  Here rank g OWNS rows [N g / G, N (g+1) / G) of the point cloud (contiguous: the concatenation of the
// ranks' visible lists in rank order is the un-sharded visible list, so the stable tie order is unchanged), projects only
// those, and ROUTES each 64-B record to the band(s) its tile box touches; a band blends what it receives, and returns one
// 48-B accumulator row per received record to its owner, which runs the per-point backward on its own rows: per-Gaussian
// work and parameters never replicate, gradients and optimiser state never cross a link.
//   gs_route_count    per destination band: number of records this rank sends (ordered-compaction counts per workgroup +
//                     their scan); the header of each send chunk
//   gs_route_scatter  copies each record into the chunk of every band it touches, in visible-list order, and remembers
//                     the position (pos[band][i]) for the way back
//   gs_gather_returned_rows   owner side of the backward exchange: acc[i] = sum over bands, in band order, of the row
//                     returned for record pos[band][i] -- a fixed order: bitwise reproducible
// Send / receive buffers: float[world][capacity + 1][16] (forward; slot 0 of a chunk is the header {count as int32 bits})
// and float[world][capacity + 1][12] (backward, same slots).  Fixed-size chunks: one all_to_all with equal splits.
constexpr int ROUTE_MAX_WORLD = 64;

// bands [b0, b1] whose tile rows the Gaussian's tile box (RAS:81-103, shrunk to the alpha >= 1/255 level set's bounding box
// when the exact cull is on: attrs[3] < inf) reaches.  The receiving band walks the same box, so nothing it would emit a
// key for is missing.
// band g = tile rows [row[g], row[g + 1]); equal blocks, or boundaries that balance the bands' work (they may leave a band
// empty).  Passed to the kernels by value.
struct BandBounds { short row[ROUTE_MAX_WORLD + 1]; };

__device__ __forceinline__ bool route_bands(const float4 a0, const float4 a1, int tw, int th, const BandBounds &bands,
                                            int world, int &b0, int &b1) {
    int t0u, t1u, t0v, t1v;
    gs_tile_box(a0.x, a0.y, a1.w, tw, th, t0u, t1u, t0v, t1v);
    gs_cull_box(a0.x, a0.y, a1.x, a1.y, a1.z, a0.w, t0u, t1u, t0v, t1v);
    if (t1u <= t0u || t1v <= t0v) return false;
    b0 = 0;
    while (b0 < world - 1 && bands.row[b0 + 1] <= t0v) ++b0;       // the band that holds tile row t0v
    b1 = b0;
    while (b1 < world - 1 && bands.row[b1 + 1] <= t1v - 1) ++b1;   // ... and the one that holds the last row
    return true;
}

template <bool SCATTER>
__global__ __launch_bounds__(GS_BLOCK) void route_kernel(
    const float4 *__restrict__ attrs, const int32_t *__restrict__ num_keys, int m_capacity,
    const int32_t *__restrict__ counters, int width, int height, BandBounds bands, int world, int nblk,
    int32_t *__restrict__ block_counts /* [world][nblk]: counts (SCATTER = false) / exclusive offsets (true) */,
    int capacity, float4 *__restrict__ send, int32_t *__restrict__ pos, const int32_t *__restrict__ counts) {
    __shared__ int s_cnt[GS_BLOCK / GS_WAVE][ROUTE_MAX_WORLD];
    if (SCATTER && blockIdx.x == 0 && (int)threadIdx.x < world)   // the chunks' headers: number of valid records
        send[(size_t)threadIdx.x * (capacity + 1) * 4] =
            make_float4(__builtin_bit_cast(float, min(counts[threadIdx.x], capacity)), 0.f, 0.f, 0.f);
    const int m = counters ? min(counters[GS_COUNTER_NUM_VISIBLE], m_capacity) : m_capacity;
    const int i = blockIdx.x * GS_BLOCK + threadIdx.x, w = threadIdx.x >> 6;
    int b0 = 0, b1 = -1;
    float4 rec[4];
    if (i < m && num_keys[i] > 0) {   // (num_keys == 0: the record is incomplete and can contribute nowhere)
        rec[0] = attrs[4 * (size_t)i];
        rec[1] = attrs[4 * (size_t)i + 1];
        if (!route_bands(rec[0], rec[1], width / GS_TILE_WIDTH, height / GS_TILE_HEIGHT, bands, world, b0, b1)) { b0 = 0; b1 = -1; }
        if (SCATTER && b1 >= b0) { rec[2] = attrs[4 * (size_t)i + 2]; rec[3] = attrs[4 * (size_t)i + 3]; }
    }
    for (int b = 0; b < world; ++b) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(b0 <= b && b <= b1);
        if (gs_lane() == 0) s_cnt[w][b] = __popcll(bal);
    }
    __syncthreads();
    if (!SCATTER) {
        if ((int)threadIdx.x < world) {
            int c = 0;
#pragma unroll
            for (int k = 0; k < GS_BLOCK / GS_WAVE; ++k) c += s_cnt[k][threadIdx.x];
            block_counts[(size_t)threadIdx.x * nblk + blockIdx.x] = c;
        }
        return;
    }
    for (int b = 0; b < world; ++b) {
        const bool goes = b0 <= b && b <= b1;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(goes);
        int p = -1;
        if (goes) {
            p = block_counts[(size_t)b * nblk + blockIdx.x] + gs_mbcnt(bal);
#pragma unroll
            for (int k = 0; k < GS_BLOCK / GS_WAVE; ++k) p += k < w ? s_cnt[k][b] : 0;
            if (p < capacity) {
                float4 *dst = send + ((size_t)b * (capacity + 1) + 1 + p) * 4;
                dst[0] = rec[0]; dst[1] = rec[1]; dst[2] = rec[2]; dst[3] = rec[3];
            } else {
                p = -1;   // (the host sees the overflow in the counts and repeats the exchange with a larger capacity)
            }
        }
        if (i < m_capacity) pos[(size_t)b * m_capacity + i] = p;
    }
}

// workgroup b: exclusive scan of band b's per-workgroup counts (in place), total -> counts[b] and the chunk's header
__global__ __launch_bounds__(GS_BLOCK) void route_scan_kernel(int32_t *__restrict__ block_counts, int nblk,
                                                             int32_t *__restrict__ counts) {
    __shared__ int lds[4];
    int32_t *row = block_counts + (size_t)blockIdx.x * nblk;
    int carry = 0;
    for (int base = 0; base < nblk; base += GS_BLOCK) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? row[i] : 0;
        int total;
        const int ex = gs_block_excl_scan(v, &total, lds);
        if (i < nblk) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = carry;
}

__global__ void route_headers_kernel(const int32_t *__restrict__ counts, int world, int capacity, float4 *__restrict__ send) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < world)
        send[(size_t)b * (capacity + 1) * 4] = make_float4(__builtin_bit_cast(float, min(counts[b], capacity)), 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(GS_BLOCK) void gather_returned_kernel(const float4 *__restrict__ returned,
                                                                 const int32_t *__restrict__ pos, int m, int m_capacity,
                                                                 int world, int capacity, float4 *__restrict__ acc) {
    const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i >= m) return;
    float d[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int npix = 0;
    for (int b = 0; b < world; ++b) {   // band order = rank order: the same additions in the same order on every run
        const int p = pos[(size_t)b * m_capacity + i];
        if (p < 0) continue;
        const float4 *row = returned + ((size_t)b * (capacity + 1) + 1 + p) * 3;
        const float4 x = row[0], y = row[1], z = row[2];
        d[0] += x.x; d[1] += x.y; d[2] += x.z; d[3] += x.w;
        d[4] += y.x; d[5] += y.y; d[6] += y.z; d[7] += y.w;
        d[8] += z.x; d[9] += z.y;
        npix += __builtin_bit_cast(int, z.z);   // pixel count: int32 bits, summed as an integer
    }
    acc[3 * (size_t)i] = make_float4(d[0], d[1], d[2], d[3]);
    acc[3 * (size_t)i + 1] = make_float4(d[4], d[5], d[6], d[7]);
    acc[3 * (size_t)i + 2] = make_float4(d[8], d[9], __builtin_bit_cast(float, npix), 0.f);
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

size_t gs_route_workspace_bytes(int n_visible_capacity, int world) {
    return sizeof(int32_t) * ((size_t)gs_div_up(n_visible_capacity > 0 ? n_visible_capacity : 1, GS_BLOCK) * (size_t)world + 64);
}

// the bands' boundaries as the kernels take them: equal blocks of rows_per_band tile rows, or the caller's world + 1 rows
static int band_bounds(int height, int rows_per_band, int world, const int32_t *band_row_bounds, BandBounds &out) {
    const int th = height / GS_TILE_HEIGHT;
    GS_REQUIRE(th < 32768, "more than 32,767 tile rows");
    for (int g = 0; g <= world; ++g) {
        long long r = band_row_bounds ? band_row_bounds[g] : (long long)g * rows_per_band;
        if (band_row_bounds)
            GS_REQUIRE(r >= 0 && r <= th && (g == 0 ? r == 0 : r >= band_row_bounds[g - 1]) && (g < world || r == th),
                       "band_row_bounds: world + 1 non-decreasing tile rows from 0 to the number of tile rows");
        out.row[g] = (short)(r < th ? r : th);
    }
    return 0;
}

int gs_route_count(const float *attrs, const int32_t *num_keys, int n_visible_capacity, const int32_t *counters, int width,
                   int height, int rows_per_band, int world, const int32_t *band_row_bounds, int32_t *counts,
                   void *workspace, void *stream) {
    GS_REQUIRE(world >= 1 && world <= ROUTE_MAX_WORLD && rows_per_band >= 1 && n_visible_capacity >= 0, "sizes (world <= 64)");
    BandBounds bands;
    if (band_bounds(height, rows_per_band, world, band_row_bounds, bands) < 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (n_visible_capacity == 0) {
        GS_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * world, s));
        return 0;
    }
    const int nblk = gs_div_up(n_visible_capacity, GS_BLOCK);
    int32_t *block_counts = (int32_t *)workspace;
    hipLaunchKernelGGL(route_kernel<false>, dim3(nblk), dim3(GS_BLOCK), 0, s, reinterpret_cast<const float4 *>(attrs),
                       num_keys, n_visible_capacity, counters, width, height, bands, world, nblk, block_counts, 0,
                       (float4 *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr);
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(route_scan_kernel, dim3(world), dim3(GS_BLOCK), 0, s, block_counts, nblk, counts);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_route_scatter(const float *attrs, const int32_t *num_keys, int n_visible_capacity, const int32_t *counters,
                     int width, int height, int rows_per_band, int world, const int32_t *band_row_bounds, int capacity,
                     const int32_t *counts, float *send, int32_t *pos, void *workspace, void *stream) {
    GS_REQUIRE(world >= 1 && world <= ROUTE_MAX_WORLD && rows_per_band >= 1 && n_visible_capacity >= 0 && capacity >= 0,
               "sizes (world <= 64)");
    BandBounds bands;
    if (band_bounds(height, rows_per_band, world, band_row_bounds, bands) < 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (n_visible_capacity == 0) {   // nothing to send: the headers alone
        hipLaunchKernelGGL(route_headers_kernel, dim3(1), dim3(ROUTE_MAX_WORLD), 0, s, counts, world, capacity,
                           reinterpret_cast<float4 *>(send));
        GS_CHECK_LAUNCH();
        return 0;
    }
    const int nblk = gs_div_up(n_visible_capacity, GS_BLOCK);
    hipLaunchKernelGGL(route_kernel<true>, dim3(nblk), dim3(GS_BLOCK), 0, s, reinterpret_cast<const float4 *>(attrs),
                       num_keys, n_visible_capacity, counters, width, height, bands, world, nblk,
                       (int32_t *)workspace, capacity, reinterpret_cast<float4 *>(send), pos, counts);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_gather_returned_rows(const float *returned, const int32_t *pos, int n_visible, int n_visible_capacity, int world,
                            int capacity, float *acc, void *stream) {
    GS_REQUIRE(world >= 1 && world <= ROUTE_MAX_WORLD && n_visible >= 0 && n_visible <= n_visible_capacity && capacity >= 0,
               "sizes (world <= 64)");
    if (n_visible == 0) return 0;
    hipLaunchKernelGGL(gather_returned_kernel, dim3(gs_div_up(n_visible, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(returned), pos, n_visible, n_visible_capacity, world, capacity,
                       reinterpret_cast<float4 *>(acc));
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
