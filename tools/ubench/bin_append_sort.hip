// Micro-benchmark for the bucket + per-bin sort design (round 3): what do K returned / fire-and-forget device-scope
// atomics spread over NB counters cost, and what does an LDS bitonic sort of NB segments cost?
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/bin_append_sort.hip -o /tmp/bin_append_sort ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// (a) histogram: fire-and-forget atomics
__global__ __launch_bounds__(256) void k_hist(const int32_t *bins, int n, int32_t *count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&count[bins[i]], 1);
}
// (b) append: returned atomics + 8-byte scattered store
__global__ __launch_bounds__(256) void k_append(const int32_t *bins, int n, const int32_t *start, int32_t *cursor,
                                                uint64_t *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int b = bins[i];
        const int pos = start[b] + atomicAdd(&cursor[b], 1);
        out[pos] = ((uint64_t)hash32(i) << 32) | (uint32_t)i;
    }
}
// (c) the same with wave-aggregated atomics for equal neighbours is pointless for random bins; instead: L2-scope atomics
// (workgroup scope on global memory: executed in the XCD's L2, NOT coherent across XCDs -- timing reference only)
__global__ __launch_bounds__(256) void k_append_l2(const int32_t *bins, int n, const int32_t *start, int32_t *cursor,
                                                   uint64_t *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int b = bins[i];
        const int pos = start[b] + __hip_atomic_fetch_add(&cursor[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (pos < n) out[pos] = ((uint64_t)hash32(i) << 32) | (uint32_t)i;
    }
}
// (d) store only (positions precomputed): what the scattered 8-B stores alone cost
__global__ __launch_bounds__(256) void k_scatter(const int32_t *pos, int n, uint64_t *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[pos[i]] = ((uint64_t)hash32(i) << 32) | (uint32_t)i;
}

// ---------------------------------------------------------------- LDS bitonic sort of one segment per workgroup
template <int CAP>
__global__ __launch_bounds__(256) void k_bitonic(const int32_t *start, const int32_t *count, uint64_t *data, int32_t *tile_out) {
    __shared__ uint64_t s[CAP];
    const int seg = blockIdx.x, base = start[seg], len = min(count[seg], CAP);
    int n = 64;
    while (n < len) n <<= 1;
    for (int i = threadIdx.x; i < n; i += 256) s[i] = i < len ? data[base + i] : ~0ull;
    __syncthreads();
    const int tid = threadIdx.x, pairs = n >> 1;
    const int per_wave = pairs >> 2;   // pairs >= 32 -> per_wave >= 8; wave w owns pairs [w, w + 1) * per_wave
    const int w = tid >> 6, lane = tid & 63;
    bool prev_cross = true;            // (the load above ended with a workgroup barrier)
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            // a step with 2 j <= n / 4 stays inside the wave's own quarter of the elements: no workgroup barrier
            const bool cross = 2 * j > (n >> 2);
            if (cross && !prev_cross) __syncthreads();
            for (int t0 = 0; t0 < per_wave; t0 += 64) {
                const int t = w * per_wave + t0 + lane;
                if (t0 + lane < per_wave) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                    const uint64_t a = s[i], b = s[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[p] = a; }
                }
            }
            if (cross) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
            prev_cross = cross;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += 256) data[base + i] = s[i];
    // expansion stand-in: each wave compacts "its" tile's entries (bit w of the low key bits) in order
    {
        int run = 0;
        for (int i0 = 0; i0 < len; i0 += 64) {
            const int i = i0 + lane;
            const uint64_t e = i < len ? s[i] : 0;
            const bool keep = i < len && ((e >> w) & 1);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            const int r = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (keep) tile_out[4 * base + w * len + run + r] = (int32_t)(e >> 16);
            run += __popcll(m);
        }
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2880000;
    const int nbs[] = {503, 2010, 8040};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int dist = 0; dist < 2; ++dist)
    for (int NB : nbs) {
        std::vector<int32_t> bins(K), cnt(NB, 0), st(NB + 1, 0), pos(K);
        uint32_t rng = 12345u;
        for (int i = 0; i < K; ++i) {
            rng = rng * 1664525u + 1013904223u;
            uint32_t r = rng >> 8;
            int b;
            if (dist == 0) b = r % NB;
            else { // centre-heavy: product of two uniform-ish draws
                rng = rng * 1664525u + 1013904223u;
                uint32_t r2 = rng >> 8;
                b = (int)(((uint64_t)(r % NB) + (r2 % NB)) / 2);
            }
            bins[i] = b; cnt[b]++;
        }
        for (int b = 0; b < NB; ++b) st[b + 1] = st[b] + cnt[b];
        { std::vector<int32_t> cur(st.begin(), st.end() - 1); for (int i = 0; i < K; ++i) pos[i] = cur[bins[i]]++; }
        int32_t *d_bins, *d_cnt, *d_st, *d_cur, *d_pos, *d_tile;
        uint64_t *d_out;
        CHECK(hipMalloc(&d_bins, 4 * K)); CHECK(hipMalloc(&d_cnt, 4 * NB)); CHECK(hipMalloc(&d_st, 4 * (NB + 1)));
        CHECK(hipMalloc(&d_cur, 4 * NB)); CHECK(hipMalloc(&d_pos, 4 * K)); CHECK(hipMalloc(&d_out, 8 * (size_t)K));
        CHECK(hipMalloc(&d_tile, 16 * (size_t)K));
        CHECK(hipMemcpy(d_bins, bins.data(), 4 * K, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_st, st.data(), 4 * (NB + 1), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_pos, pos.data(), 4 * K, hipMemcpyHostToDevice));
        const int grid = (K + 255) / 256, reps = 20;
        float t_hist = 0, t_app = 0, t_l2 = 0, t_sc = 0, t_sort = 0;
        for (int r = 0; r < reps + 2; ++r) {
            CHECK(hipMemsetAsync(d_cnt, 0, 4 * NB, 0));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_hist, dim3(grid), dim3(256), 0, 0, d_bins, K, d_cnt);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            if (r >= 2) t_hist += time_ms(e0, e1);
            CHECK(hipMemsetAsync(d_cur, 0, 4 * NB, 0));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_append_l2, dim3(grid), dim3(256), 0, 0, d_bins, K, d_st, d_cur, d_out);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            if (r >= 2) t_l2 += time_ms(e0, e1);
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(256), 0, 0, d_pos, K, d_out);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            if (r >= 2) t_sc += time_ms(e0, e1);
            CHECK(hipMemsetAsync(d_cur, 0, 4 * NB, 0));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_append, dim3(grid), dim3(256), 0, 0, d_bins, K, d_st, d_cur, d_out);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            if (r >= 2) t_app += time_ms(e0, e1);
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_bitonic<8192>, dim3(NB), dim3(256), 0, 0, d_st, d_cnt, d_out, d_tile);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            if (r >= 2) t_sort += time_ms(e0, e1);
        }
        // verify the last sort
        std::vector<uint64_t> out(K);
        CHECK(hipMemcpy(out.data(), d_out, 8 * (size_t)K, hipMemcpyDeviceToHost));
        long bad = 0; int maxlen = 0;
        for (int b = 0; b < NB; ++b) {
            maxlen = std::max(maxlen, cnt[b]);
            if (cnt[b] > 8192) continue;
            for (int i = st[b] + 1; i < st[b + 1]; ++i) bad += out[i - 1] > out[i];
        }
        printf("dist=%d K=%d NB=%d (max segment %d): hist(no-return atomics) %.1f us | append(returned atomics + store) %.1f us | "
               "append with L2-scope atomics %.1f us | scatter store only %.1f us | LDS bitonic sort + expansion %.1f us | unsorted pairs %ld\n",
               dist, K, NB, maxlen, 1e3 * t_hist / reps, 1e3 * t_app / reps, 1e3 * t_l2 / reps, 1e3 * t_sc / reps, 1e3 * t_sort / reps, bad);
        hipFree(d_bins); hipFree(d_cnt); hipFree(d_st); hipFree(d_cur); hipFree(d_pos); hipFree(d_out); hipFree(d_tile);
    }
    return 0;
}
