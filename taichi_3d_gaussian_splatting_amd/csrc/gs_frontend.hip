// gs_frontend.hip -- per-point front end of the gfx950 rasteriser:
//   pose inverse, frustum filter + ordered compaction, EWA projection (+ tile counting),
//   block-sum scan, sort-key generation, tile ranges.
// Hand-written HIP for CDNA4 (wave64).  The arithmetic follows the reference formulas
// (cited per function; RAS/GP3/SPH/UTL as in include/gsplat_hip.h) and is evaluated in the
// same left-to-right order as the CPU oracle with FMA contraction disabled, so that the
// discrete decisions derived from it (frustum test, depth quantisation, tile boxes) agree
// bit-for-bit with the oracle.
#include "gs_common.h"
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------ small math (device)
struct Mat3 { float m[9]; };

// GP3:31-48 rotation_matrix_from_quaternion (q = x,y,z,w; not normalised here)
__device__ __forceinline__ Mat3 rotmat_from_q(float x, float y, float z, float w) {
    Mat3 R;
    gs_rotmat_from_q(x, y, z, w, R.m);
    return R;
}

// GP3:14-27 project_point_to_camera
__device__ __forceinline__ void project_point(const Mat3 &R, const float t[3], const float K[9],
                                              const float p[3], float uv[2], float c[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
        c[i] = ((R.m[3 * i] * p[0] + R.m[3 * i + 1] * p[1]) + R.m[3 * i + 2] * p[2]) + t[i] * 1.f;
    float u1 = (K[0] * c[0] + K[1] * c[1]) + K[2] * c[2];
    float v1 = (K[3] * c[0] + K[4] * c[1]) + K[5] * c[2];
    uv[0] = u1 / c[2];
    uv[1] = v1 / c[2];
}

// C(m x n) = A(m x k) @ B(k x n), left-to-right sums (Taichi's unrolled matmul order)
template <int M, int Kd, int N>
__device__ __forceinline__ void matmul(const float *A, const float *B, float *C) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float s = A[i * Kd] * B[j];
#pragma unroll
            for (int l = 1; l < Kd; ++l) s = s + A[i * Kd + l] * B[l * N + j];
            C[i * N + j] = s;
        }
}

__device__ __forceinline__ void tile_box(float u, float v, float r, int tw, int th, int &t0u, int &t1u,
                                         int &t0v, int &t1v) {
    gs_tile_box(u, v, r, tw, th, t0u, t1u, t0v, t1v);
}

// Tile-row ownership of one GPU (image-space sharding): rows {begin + k*step, k >= 0} below `end`.
struct RowOwner { int begin, step, end; };
// number of owned tile rows in [lo, hi)
__device__ __forceinline__ int owned_rows(int lo, int hi, RowOwner ow) {
    hi = min(hi, ow.end);
    if (ow.step == 1) return max(hi - max(lo, ow.begin), 0);   // contiguous band (and the single-GPU case): no division
    int f = ow.begin;
    if (lo > ow.begin) f = ow.begin + ((lo - ow.begin + ow.step - 1) / ow.step) * ow.step;
    return f < hi ? (hi - 1 - f) / ow.step + 1 : 0;
}

// Sort keys are emitted per BIN = (1 << bin_shift)^2 tiles; bin_shift 0 (default) = per tile as the reference
// (RAS:131-172).  With bin_shift > 0 the blend kernels walk their bin's depth-sorted list and stage the entries that
// belong to their own tile (box test + exact contribution test, gs_entry_in_tile): the per-tile sequence is unchanged --
// the bin list filtered by tile membership is the tile's list in the same (depth, index) order -- but 3-16x fewer keys
// are generated, sorted and ranged.  A Gaussian emits a key for bin (bu, bv) iff the bin holds a tile of its box
// (RAS:81-103) in a tile row this GPU owns and (with the exact cull) the Gaussian can reach alpha >= 1/255 somewhere
// on those tiles.  for_each_emitting_bin visits exactly those bins; gs_preprocess counts them, gs_make_keys writes them:
// the same function on the same stored values, so count and keys agree.
// (a) per-lane walk: every lane visits the bins of its own Gaussian.  Cheapest per pair (~45 instructions), but a wave
//     waits for its largest box.
template <typename Emit>
__device__ __forceinline__ void for_each_emitting_bin(int t0u, int t1u, int t0v, int t1v, int bin_shift, RowOwner ow,
                                                      int cull, float ux, float uy, float A, float B, float C, float qmax,
                                                      Emit emit) {
    if (t1u <= t0u || t1v <= t0v) return;
    const int b0u = t0u >> bin_shift, b1u = ((t1u - 1) >> bin_shift) + 1;
    const int b0v = t0v >> bin_shift, b1v = ((t1v - 1) >> bin_shift) + 1;
    // slopes of the conic's conjugate diameters: per-Gaussian constants of the contribution test
    const float sy = -B * __builtin_amdgcn_rcpf(C), sx = -B * __builtin_amdgcn_rcpf(A);
    // tile_u outer, tile_v inner: the generation order of RAS:161-166 (with bin_shift 0 the unsorted keys are the
    // reference's, position by position)
    for (int bu = b0u; bu < b1u; ++bu) {
        const int u_lo = max(t0u, bu << bin_shift), u_hi = min(t1u, (bu + 1) << bin_shift);   // box columns in this bin
        const float x0 = (float)(u_lo * GS_TILE_WIDTH) + 0.5f, x1 = (float)(u_hi * GS_TILE_WIDTH) - 0.5f;
        for (int bv = b0v; bv < b1v; ++bv) {
            const int v_lo = max(t0v, bv << bin_shift), v_hi = min(t1v, (bv + 1) << bin_shift);
            if (owned_rows(v_lo, v_hi, ow) == 0) continue;
            if (cull && !gs_rect_may_contribute(ux, uy, A, B, C, sx, sy, qmax, x0, x1,
                                                (float)(v_lo * GS_TILE_HEIGHT) + 0.5f,
                                                (float)(v_hi * GS_TILE_HEIGHT) - 0.5f))
                continue;
            emit(bu, bv);
        }
    }
}

// (b) balanced walk (gs_make_keys).  A Gaussian's box holds 1 .. several thousand bins: a lane that walks its own box
// alone makes the other 63 lanes wait for the largest box of the wave, and writes its keys with scattered 8-B stores.
// Here the (bin, Gaussian) pairs of the 64 Gaussians of a wave are numbered consecutively -- Gaussians in lane order, a
// Gaussian's bins in the generation order of RAS:161-166 (tile_u outer, tile_v inner) -- and dealt to the lanes 64 at a
// time.  Surviving pairs are then visited in exactly the order in which their keys are laid out, so key generation
// becomes a stream compaction with consecutive lanes writing consecutive keys.  Both walks visit the same pairs in the
// same order and take the same decisions.
struct BinWalkRec {           // one Gaussian, 64 B in LDS
    float u, v, A, B;
    float C, qmax, sx, sy;    // sx = -B/A, sy = -B/C: slopes of the conic's conjugate diameters (contribution test)
    int t0u, t1u, t0v, t1v;   // tile box (RAS:81-103)
    int b0u, b0v, nbv;        // first bin column / row of the walk, bin rows per column
    unsigned magic;           // floor(2^32 / nbv) + 1: pair index -> (column, row) without an integer division
};
// -> number of (bin, Gaussian) pairs to visit.  Bin rows without a tile row of this GPU are left out up front when the
// GPU owns a contiguous band (row_step 1), so that a Gaussian outside the band costs nothing.
__device__ __forceinline__ int make_walk_rec(BinWalkRec &r, bool active, float u, float v, float A, float B, float C,
                                             float qmax, int t0u, int t1u, int t0v, int t1v, int bin_shift, RowOwner ow) {
    r.u = u; r.v = v; r.A = A; r.B = B; r.C = C; r.qmax = qmax;
    r.sx = -B * __builtin_amdgcn_rcpf(A); r.sy = -B * __builtin_amdgcn_rcpf(C);
    r.t0u = t0u; r.t1u = t1u; r.t0v = t0v; r.t1v = t1v;
    int b0u = 0, b1u = 0, b0v = 0, b1v = 0;
    if (active && t1u > t0u && t1v > t0v) {
        b0u = t0u >> bin_shift; b1u = ((t1u - 1) >> bin_shift) + 1;
        int v_lo = t0v, v_hi = t1v;
        if (ow.step == 1) { v_lo = max(v_lo, ow.begin); v_hi = min(v_hi, ow.end); }
        if (v_hi > v_lo) { b0v = v_lo >> bin_shift; b1v = ((v_hi - 1) >> bin_shift) + 1; }
    }
    r.b0u = b0u; r.b0v = b0v; r.nbv = b1v - b0v;
    r.magic = r.nbv > 1 ? 0xffffffffu / (unsigned)r.nbv + 1u : 0u;
    return (b1u - b0u) * (b1v - b0v);
}
// Which walk a wave takes (gs_preprocess and gs_make_keys ask the same question of the same numbers): per lane when at
// least half of its Gaussians are heavy (> 256 pairs: the wave is balanced as it is and the per-lane walk costs ~2.5x
// less per pair), shared among the lanes otherwise.
constexpr int HEAVY_PAIRS = 256;
__device__ __forceinline__ bool gs_mostly_heavy_wave(int npairs) {
    return __popcll(__builtin_amdgcn_ballot_w64(npairs > HEAVY_PAIRS)) >= GS_WAVE / 2;
}
// visit(owner_lane, bin_u, bin_v, survives) is called by every lane once per round of 64 pairs (wave-convergent, so that
// it may use ballots); `survives` is false for the padding lanes of the last round.  recs: the 64 records of this wave.
template <typename Visit>
__device__ __forceinline__ void walk_bins_balanced(const BinWalkRec *__restrict__ recs, int npairs, int bin_shift,
                                                   RowOwner ow, int cull, Visit visit) {
    const int incl = gs_wave_incl_scan(npairs), excl = incl - npairs;
    const int total = __builtin_amdgcn_readlane(incl, GS_WAVE - 1), lane = gs_lane();
    for (int base = 0; base < total; base += GS_WAVE) {
        const int q = base + lane;
        const bool valid = q < total;
        int owner = 0;   // number of lanes whose pairs all lie before q: binary search over the inclusive scan
#pragma unroll
        for (int step = GS_WAVE / 2; step > 0; step >>= 1)
            if (__shfl(incl, owner + step - 1, GS_WAVE) <= q) owner += step;
        owner = min(owner, GS_WAVE - 1);
        const int p = q - __shfl(excl, owner, GS_WAVE);
        const BinWalkRec r = recs[owner];
        bool survives = false;
        int bu = 0, bv = 0;
        if (valid) {
            const int col = r.nbv > 1 ? (int)__umulhi((unsigned)p, r.magic) : p;
            bu = r.b0u + col;
            bv = r.b0v + (p - col * r.nbv);
            const int u_lo = max(r.t0u, bu << bin_shift), u_hi = min(r.t1u, (bu + 1) << bin_shift);   // box tiles in the bin
            const int v_lo = max(r.t0v, bv << bin_shift), v_hi = min(r.t1v, (bv + 1) << bin_shift);
            survives = owned_rows(v_lo, v_hi, ow) > 0;
            if (survives && cull)
                survives = gs_rect_may_contribute(r.u, r.v, r.A, r.B, r.C, r.sx, r.sy, r.qmax,
                                                  (float)(u_lo * GS_TILE_WIDTH) + 0.5f, (float)(u_hi * GS_TILE_WIDTH) - 0.5f,
                                                  (float)(v_lo * GS_TILE_HEIGHT) + 0.5f, (float)(v_hi * GS_TILE_HEIGHT) - 0.5f);
        }
        visit(owner, bu, bv, survives);
    }
}

// ------------------------------------------------------------------ pose inverse
// UTL:396-432 inverse_SE3_qt_torch: q_inv = conj(q) (not renormalised),
// t_inv = -rot(normalise(q_inv), t) with the Hamilton products of UTL:402-412.
__device__ __forceinline__ void quat_mul(const float a[4], const float b[4], float o[4]) {
    float x0 = a[0], y0 = a[1], z0 = a[2], w0 = a[3];
    float x1 = b[0], y1 = b[1], z1 = b[2], w1 = b[3];
    o[0] = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1;
    o[1] = w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1;
    o[2] = w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1;
    o[3] = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
}

// one pose: (q, t)_pointcloud<-camera -> (q, t)_camera<-pointcloud
__device__ __forceinline__ void pose_inverse_one(const float *__restrict__ q, const float *__restrict__ t, int i,
                                                 float qi[4], float ti[3]) {
    qi[0] = -q[4 * i]; qi[1] = -q[4 * i + 1]; qi[2] = -q[4 * i + 2]; qi[3] = q[4 * i + 3];
    float nrm = sqrtf(((qi[0] * qi[0] + qi[1] * qi[1]) + qi[2] * qi[2]) + qi[3] * qi[3]);
    float qn[4] = {qi[0] / nrm, qi[1] / nrm, qi[2] / nrm, qi[3] / nrm};
    float v[4] = {t[3 * i], t[3 * i + 1], t[3 * i + 2], 0.f};
    float qc[4] = {-qn[0], -qn[1], -qn[2], qn[3]};
    float tmp[4], out[4];
    quat_mul(qn, v, tmp);
    quat_mul(tmp, qc, out);
#pragma unroll
    for (int k = 0; k < 3; ++k) ti[k] = -out[k];
}

__global__ void pose_inverse_kernel(const float *__restrict__ q, const float *__restrict__ t,
                                    float *__restrict__ q_inv, float *__restrict__ t_inv, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float qi[4], ti[3];
    pose_inverse_one(q, t, i, qi, ti);
#pragma unroll
    for (int k = 0; k < 4; ++k) q_inv[4 * i + k] = qi[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t_inv[3 * i + k] = ti[k];
}

// ------------------------------------------------------------------ filter + ordered compaction
// One 256-thread workgroup handles FILTER_ITEMS consecutive points in rounds of 256 so that
// the xyz reads are coalesced and the output order is the input order.
constexpr int FILTER_ROUNDS = 4;
constexpr int FILTER_ITEMS = GS_BLOCK * FILTER_ROUNDS;

// RAS:31-78 filter_point_in_camera
__global__ __launch_bounds__(GS_BLOCK) void filter_kernel(
    const float *__restrict__ xyz, const int8_t *__restrict__ invalid, const int32_t *__restrict__ obj,
    const float *__restrict__ Kmat, const float *__restrict__ q_cp, const float *__restrict__ t_cp, int n,
    float near_plane, float far_plane, int width, int height, int8_t *__restrict__ mask,
    int32_t *__restrict__ block_counts, int32_t *__restrict__ counters, const float *__restrict__ q_pc,
    const float *__restrict__ t_pc, int n_obj, float *__restrict__ q_cp_out, float *__restrict__ t_cp_out) {
    __shared__ int s_count;
    if (threadIdx.x == 0) s_count = 0;
    // the frame's counters start at zero (the first kernel of a frame does it: no separate fill launch)
    if (blockIdx.x == 0 && threadIdx.x < GS_NUM_COUNTERS) counters[threadIdx.x] = 0;
    // Fused pose inverse (q_pc != null; UTL:426-432): every lane inverts the pose of its point's object itself -- the same
    // device function as gs_pose_inverse, hence the same bits -- and workgroup 0 leaves the inverted poses in
    // q_cp_out / t_cp_out for the later stages: one launch (and one dependent boundary) less per frame
    const bool fused_pose = q_pc != nullptr;
    if (fused_pose && blockIdx.x == 0)
        for (int o = threadIdx.x; o < n_obj; o += GS_BLOCK) {
            float qi[4], ti[3];
            pose_inverse_one(q_pc, t_pc, o, qi, ti);
#pragma unroll
            for (int k = 0; k < 4; ++k) q_cp_out[4 * o + k] = qi[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) t_cp_out[3 * o + k] = ti[k];
        }
    int cached_obj = -1;
    float cq[4] = {0.f, 0.f, 0.f, 1.f}, ct[3] = {0.f, 0.f, 0.f};
    __syncthreads();
    float K[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) K[k] = Kmat[k];
    int local = 0;
#pragma unroll
    for (int r = 0; r < FILTER_ROUNDS; ++r) {
        int i = blockIdx.x * FILTER_ITEMS + r * GS_BLOCK + threadIdx.x;
        bool vis = false;
        if (i < n) {
            if (invalid[i] != 1) {
                int o = obj[i];
                if (o != cached_obj) {   // (one object in most scenes: inverted or loaded once per lane)
                    if (fused_pose) {
                        pose_inverse_one(q_pc, t_pc, o, cq, ct);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) cq[k] = q_cp[4 * o + k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) ct[k] = t_cp[3 * o + k];
                    }
                    cached_obj = o;
                }
                Mat3 R = rotmat_from_q(cq[0], cq[1], cq[2], cq[3]);
                float t[3] = {ct[0], ct[1], ct[2]};
                float p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
                float uv[2], c[3];
                project_point(R, t, K, p, uv, c);
                vis = c[2] > near_plane && c[2] < far_plane &&
                      uv[0] >= (float)(-GS_TILE_WIDTH * GS_BOUNDARY_TILES) &&
                      uv[0] < (float)(width + GS_TILE_WIDTH * GS_BOUNDARY_TILES) &&
                      uv[1] >= (float)(-GS_TILE_HEIGHT * GS_BOUNDARY_TILES) &&
                      uv[1] < (float)(height + GS_TILE_HEIGHT * GS_BOUNDARY_TILES);
            }
            mask[i] = vis ? 1 : 0;
        }
        unsigned long long b = __ballot(vis);
        if (gs_lane() == 0) local += __popcll(b);
    }
    if (gs_lane() == 0) atomicAdd(&s_count, local);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_count;
}

// single-workgroup exclusive scan over n ints (in place); total -> *total_out (saturating).
// blockIdx.x == 1 scans the optional second array (two independent scans in one launch).
__global__ __launch_bounds__(GS_BLOCK) void scan_single_block_kernel(int32_t *__restrict__ data, int n,
                                                                    int32_t *__restrict__ total_out,
                                                                    int32_t *__restrict__ data2,
                                                                    int32_t *__restrict__ total_out2,
                                                                    const int32_t *__restrict__ counters,
                                                                    int32_t *__restrict__ host_mirror, uint32_t stamp) {
    __shared__ long long lds[GS_BLOCK / GS_WAVE];
    int mirror_slot = GS_COUNTER_NUM_KEYS;
    if (blockIdx.x == 1) { data = data2; total_out = total_out2; mirror_slot = GS_COUNTER_NUM_SLOTS; }
    // every thread owns SCAN_ITEMS consecutive values: all loads of a 4096-value chunk are in flight together (the
    // one-value-per-thread form paid one L2 round trip per 256 values: 12 us for the 3.9 k block sums of 1e6 points)
    constexpr int SCAN_ITEMS = 16;
    long long carry = 0;
    for (int base = 0; base < n; base += GS_BLOCK * SCAN_ITEMS) {
        const int first = base + threadIdx.x * SCAN_ITEMS;
        int v[SCAN_ITEMS];
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) v[k] = first + k < n ? data[first + k] : 0;
        long long mine = 0;   // 64-bit throughout: the total saturates at INT32_MAX instead of wrapping
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) mine += v[k];
        long long incl = mine;
#pragma unroll
        for (int d = 1; d < GS_WAVE; d <<= 1) {
            const long long o = __shfl_up(incl, d, GS_WAVE);
            if (gs_lane() >= d) incl += o;
        }
        const int w = threadIdx.x >> 6;
        if (gs_lane() == GS_WAVE - 1) lds[w] = incl;
        __syncthreads();
        long long before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < GS_BLOCK / GS_WAVE; ++i) {
            const long long t = lds[i];
            if (i < w) before += t;
            total += t;
        }
        __syncthreads();
        long long run = carry + before + incl - mine;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            if (first + k < n) data[first + k] = run > 0x7fffffffLL ? 0x7fffffff : (int)run;
            run += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        const int total = carry > 0x7fffffffLL ? 0x7fffffff : (int)carry;
        *total_out = total;
        // host_mirror (pinned, host-coherent memory mapped into the device's address space): the frame's sizes go to the
        // host straight from this kernel -- visible when the kernel has completed (the event recorded behind it) -- instead
        // of through a copy launch of their own (~6 us of a frame that is a chain of such launches)
        if (host_mirror != nullptr && stamp == 0u) {
            host_mirror[mirror_slot] = total;
            if (blockIdx.x == 0) {
                host_mirror[GS_COUNTER_NUM_VISIBLE] = counters[GS_COUNTER_NUM_VISIBLE];
                host_mirror[GS_COUNTER_MAX_DEPTH_KEY] = counters[GS_COUNTER_MAX_DEPTH_KEY];
            }
        } else if (host_mirror != nullptr) {
            // stamped words (gs_scan_block_sums2_stamped): {stamp, value} in one 8-byte system-scope store each -- written
            // through to the host's memory, valid or not on its own, no event and no fence behind them
            unsigned long long *words = reinterpret_cast<unsigned long long *>(host_mirror);
            auto put = [&](int slot, int value) {
                __hip_atomic_store(&words[slot], ((unsigned long long)stamp << 32) | (unsigned)value, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            };
            put(mirror_slot, total);
            if (blockIdx.x == 0) {
                put(GS_COUNTER_NUM_VISIBLE, counters[GS_COUNTER_NUM_VISIBLE]);
                put(GS_COUNTER_MAX_DEPTH_KEY, counters[GS_COUNTER_MAX_DEPTH_KEY]);
            }
        }
    }
}

// RAS:861-870: point_id[mask] -- order-preserving compaction with wave ballots.  Every workgroup first sums the
// counts of the workgroups before it (at most a few thousand ints, L2-resident): cheaper than a separate scan
// launch.  The last workgroup publishes the total (saturating) as the visible count.
__global__ __launch_bounds__(GS_BLOCK) void compact_kernel(const int8_t *__restrict__ mask, int n,
                                                          const int32_t *__restrict__ block_counts,
                                                          int32_t *__restrict__ ids, int32_t *__restrict__ total_out) {
    __shared__ int s_wave[GS_BLOCK / GS_WAVE];
    __shared__ int s_before[GS_BLOCK / GS_WAVE];
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
    for (int r = 0; r < FILTER_ROUNDS; ++r) {
        int i = blockIdx.x * FILTER_ITEMS + r * GS_BLOCK + threadIdx.x;
        bool vis = i < n && mask[i] != 0;
        unsigned long long b = __ballot(vis);
        int rank = gs_mbcnt(b);
        if (gs_lane() == 0) s_wave[w] = __popcll(b);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < GS_BLOCK / GS_WAVE; ++k) {
            int c = s_wave[k];
            if (k < w) before += c;
            total += c;
        }
        if (vis) ids[running + before + rank] = i;
        running += total;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = running;
}

// The colours of a wave's Gaussians (RAS:280-282,302-310; ray origin = (-R^T) t, UTL:495-510), wave-convergent.  The 192 B
// of SH coefficients are read COOPERATIVELY: every lane leaves the sixteen basis values of its Gaussian in LDS (Yw: 64 x
// 16 floats of this wave), then sixteen lanes serve one Gaussian -- twelve of them load one 16-byte quarter of a colour
// channel each (one coalesced 192-byte read per Gaussian, four Gaussians per load instruction) and form that quarter's
// part of SH . Y, two exchanges add the quarters in the tree of gs_view_colour, the sigmoid follows.  (One lane gathering
// its own row with twelve scattered 16-byte loads kept 64 partially used lines per instruction in flight: 1.64x the
// algorithmic HBM bytes, round 3.)  Shared by gs_preprocess and gs_view_colours: the same bits from either.
__device__ __forceinline__ void gs_wave_view_colours(bool wants, int id, const float *__restrict__ xyz,
                                                     const float *__restrict__ feat, const int32_t *__restrict__ obj,
                                                     const float *__restrict__ q_cp, const float *__restrict__ t_cp,
                                                     float *Yw, float rgb[3]) {
    const unsigned long long need = __builtin_amdgcn_ballot_w64(wants);
    if (need == 0ull) return;   // wave-uniform
    const int lane = gs_lane();
    __builtin_amdgcn_wave_barrier();
    if (wants) {
        const int o = obj[id];
        const Mat3 W = rotmat_from_q(q_cp[4 * o], q_cp[4 * o + 1], q_cp[4 * o + 2], q_cp[4 * o + 3]);
        const float t[3] = {t_cp[3 * o], t_cp[3 * o + 1], t_cp[3 * o + 2]};
        const float p[3] = {xyz[3 * (size_t)id], xyz[3 * (size_t)id + 1], xyz[3 * (size_t)id + 2]};
        float Y[16];
        gs_view_basis(W.m, t, p, Y);
        float4 *dst = reinterpret_cast<float4 *>(Yw + 16 * lane);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_float4(Y[4 * k], Y[4 * k + 1], Y[4 * k + 2], Y[4 * k + 3]);
    }
    __builtin_amdgcn_wave_barrier();
    const int sub = lane & 15, quarter = sub & 3;
    constexpr int IN_FLIGHT = 8;   // Gaussians-of-a-group whose coefficient loads are in flight together
#pragma unroll
    for (int it0 = 0; it0 < 16; it0 += IN_FLIGHT) {
        float4 c4[IN_FLIGHT];
#pragma unroll
        for (int k = 0; k < IN_FLIGHT; ++k) {
            const int g = (it0 + k) * 4 + (lane >> 4);                 // the Gaussian (lane of this wave) served
            const int gid = __shfl(id, g, GS_WAVE);
            const bool on = ((need >> g) & 1ull) != 0ull && sub < 12;
            c4[k] = on ? gs_load_stream(reinterpret_cast<const float4 *>(feat + (size_t)GS_FEATURE_DIM * gid) + 2 + sub)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < IN_FLIGHT; ++k) {
            const int g = (it0 + k) * 4 + (lane >> 4);
            const float4 y = reinterpret_cast<const float4 *>(Yw + 16 * g)[quarter];
            const float part = gs_sh_quarter(c4[k].x, c4[k].y, c4[k].z, c4[k].w, y.x, y.y, y.z, y.w);
            const float pair = part + __shfl_xor(part, 1, GS_WAVE);    // q0 + q1 | q2 + q3
            const float sum = pair + __shfl_xor(pair, 2, GS_WAVE);     // (q0 + q1) + (q2 + q3)
            // the channel's first lane leaves the SUM in the (now consumed) basis slot of the Gaussian; the sigmoid is applied
            // once per Gaussian and channel below (round 6: applied here it was evaluated by all 64 lanes in each of the sixteen
            // iterations -- 1,024 exponentials and divisions per wave for the 192 that are kept)
            if (quarter == 0 && sub < 12 && ((need >> g) & 1ull) != 0ull) Yw[16 * g + (sub >> 2)] = sum;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (wants) {
        rgb[0] = gs_colour_from_sum(Yw[16 * lane]);
        rgb[1] = gs_colour_from_sum(Yw[16 * lane + 1]);
        rgb[2] = gs_colour_from_sum(Yw[16 * lane + 2]);
    }
}

// ------------------------------------------------------------------ per-visible-point projection
// RAS:239-315 generate_point_attributes_in_camera_plane + RAS:106-128 generate_num_overlap_tiles.
// One lane per visible point; the 224-B feature row is read as 14 x 16-B loads (the lines are reused by
// the 14 loads out of L1; measured: the kernel is bound by its ~2.7 k VALU instructions per wave -- IEEE
// divisions, expf, the cull loop -- and an LDS-staged coalesced gather was 6 % slower: lower occupancy).
template <bool COLOUR>
__global__ __launch_bounds__(GS_BLOCK, 6) void preprocess_kernel(
    const float *__restrict__ xyz, float *__restrict__ feat, const int32_t *__restrict__ obj,
    const float *__restrict__ Kmat, const float *__restrict__ q_cp, const float *__restrict__ t_cp,
    const int32_t *__restrict__ ids, int m_capacity, int use_device_count, int width, int height, RowOwner ow,
    int bin_shift, int cull, int always_store_q, float depth_scale, int32_t *__restrict__ counters,
    float *__restrict__ attrs, int32_t *__restrict__ ntiles_full, int32_t *__restrict__ nkeys,
    int32_t *__restrict__ block_sums, int32_t *__restrict__ block_sums_full) {
    __shared__ int s_sum, s_sum_full, s_dq;
    __shared__ BinWalkRec s_rec[GS_BLOCK];   // the key count shares its Gaussians' bins among the lanes of a wave
    __shared__ int s_cnt[GS_BLOCK];
    if (threadIdx.x == 0) { s_sum = 0; s_sum_full = 0; s_dq = 0; }
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    // the number of visible points may still be on its way to the host: read it on the device
    const int m = use_device_count ? min(counters[GS_COUNTER_NUM_VISIBLE], m_capacity) : m_capacity;
    const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
    int owned = 0, full = 0, dq = 0;
    const bool live = i < m;
    // quantities that outlive the geometry phase (the count phase below runs wave-convergent, outside any branch)
    float u_ = 0.f, v_ = 0.f, z_ = 0.f, cA = 0.f, cB = 0.f, cC = 0.f, radius = 0.f, opacity = 0.f, rescale = 0.f, qmax = 0.f;
    int t0u = 0, t1u = 0, t0v = 0, t1v = 0, id = 0;
    if (live) {
        id = ids[i];
        float4 *row4 = reinterpret_cast<float4 *>(feat + (size_t)GS_FEATURE_DIM * id);
        // q, log-scale and the opacity logit now (32 B); the 192 B of SH coefficients only if this Gaussian emits a key on
        // this GPU (under tile-row sharding most visible Gaussians do not)
        float f[8];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float4 v = row4[k];
            f[4 * k] = v.x; f[4 * k + 1] = v.y; f[4 * k + 2] = v.z; f[4 * k + 3] = v.w;
        }
        // RAS:196-205: q <- q/|q|, written back in place
        float nrm = sqrtf(((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]) + f[3] * f[3]);
        const float4 q_in = make_float4(f[0], f[1], f[2], f[3]);
        f[0] = f[0] / nrm; f[1] = f[1] / nrm; f[2] = f[2] / nrm; f[3] = f[3] / nrm;
        // the store is skipped when it would not change the row (an already normalised quaternion: every frame but
        // the first of a static scene) -- same memory contents, 64 B of HBM write granule per Gaussian less.  A training
        // iteration always pays it (the optimiser has moved q); always_store_q lets a benchmark on a static scene pay it too
        if (always_store_q || f[0] != q_in.x || f[1] != q_in.y || f[2] != q_in.z || f[3] != q_in.w)
            row4[0] = make_float4(f[0], f[1], f[2], f[3]);

        float K[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) K[k] = Kmat[k];
        const int o = obj[id];
        const Mat3 W = rotmat_from_q(q_cp[4 * o], q_cp[4 * o + 1], q_cp[4 * o + 2], q_cp[4 * o + 3]);
        const float t[3] = {t_cp[3 * o], t_cp[3 * o + 1], t_cp[3 * o + 2]};
        const float p[3] = {xyz[3 * (size_t)id], xyz[3 * (size_t)id + 1], xyz[3 * (size_t)id + 2]};
        float uv[2], c[3];
        project_point(W, t, K, p, uv, c);
        dq = (int32_t)(c[2] * depth_scale);  // the quantised depth of RAS:159-160 (for the key width)

        // GP3:161-191 project_to_camera_covariance: cov = J W Sigma W^T J^T, left to right
        float J[6] = {K[0] / c[2], 0.f, -(K[0] * c[0]) / (c[2] * c[2]),
                      0.f, K[4] / c[2], -(K[4] * c[1]) / (c[2] * c[2])};
        const Mat3 R = rotmat_from_q(f[0], f[1], f[2], f[3]);
        // The scale activation is the one transcendental on the way to the INTEGER outputs (covariance -> radius -> tile
        // box -> counts, keys, slots).  Evaluated in double and rounded once it is the correctly rounded fp32 exponential
        // (up to double rounding, 2^-29) -- the oracle does the same, so the two sides agree on the radius to the last bit
        // whatever their libms; with expf on both sides 6 % of the radii were one ulp apart and a tile-box edge moved
        // across a tile boundary once in a few thousand random frames (tests/test_fuzz_gpu.py, case 6102).
        float S[9] = {gs_exp_cr(f[4]), 0.f, 0.f, 0.f, gs_exp_cr(f[5]), 0.f, 0.f, 0.f, gs_exp_cr(f[6])};
        float RS[9], RSS[9], Rt[9], Sigma[9];
        matmul<3, 3, 3>(R.m, S, RS);
        matmul<3, 3, 3>(RS, S, RSS);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Rt[b * 3 + a] = R.m[a * 3 + b];
        matmul<3, 3, 3>(RSS, Rt, Sigma);
        float JW[6], JWS[6], Wt[9], JWSW[6], Jt[6], cov[4];
        matmul<2, 3, 3>(J, W.m, JW);
        matmul<2, 3, 3>(JW, Sigma, JWS);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Wt[b * 3 + a] = W.m[a * 3 + b];
        matmul<2, 3, 3>(JWS, Wt, JWSW);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Jt[b * 2 + a] = J[a * 3 + b];
        matmul<2, 3, 2>(JWSW, Jt, cov);

        // UTL:257-272 get_point_conic_and_rescale (+0.3 low-pass only inside the conic)
        float det0 = cov[0] * cov[3] - cov[1] * cov[2];
        float ca = cov[0] + 0.3f, cd = cov[3] + 0.3f;
        float det = ca * cd - cov[1] * cov[2];
        rescale = sqrtf(fmaxf(0.0f, det0 / det));
        float inv = 1.0f / det;

        // RAS:311-315 radius from the un-filtered covariance
        float dd = cov[0] - cov[3];
        float lam = (cov[0] + cov[3] + sqrtf(dd * dd + 4.0f * cov[1] * cov[2])) / 2.0f;
        radius = sqrtf(lam) * 3.0f;
        // (the correctly rounded exponential here too: the opacity is an INPUT of every 1/255 decision of the blend passes,
        // which are taken exactly as the reference takes them only if both sides hold the same opacity bits)
        opacity = 1.f / (1.f + gs_exp_cr(-f[7]));  // RAS:299-300
        cA = inv * cd; cB = inv * (-cov[1]); cC = inv * ca;
        u_ = uv[0]; v_ = uv[1]; z_ = c[2];

        tile_box(uv[0], uv[1], radius, width / GS_TILE_WIDTH, height / GS_TILE_HEIGHT, t0u, t1u, t0v, t1v);
        full = (t1u - t0u) * (t1v - t0v);
        ntiles_full[i] = full;
        // the exact-cull bound of this Gaussian; +inf (never culled) when the cull is off.  Stored in the record: the
        // blend kernels apply the same test per tile.
        qmax = cull ? gs_cull_qmax(opacity * rescale) : __builtin_inff();
    }
    // number of sort keys = bins reached on this GPU: the walk of gs_make_keys on the same values (count and keys agree).
    // The (bin, Gaussian) pairs of a wave's 64 Gaussians are dealt to its lanes 64 at a time, so that ONE screen-filling
    // Gaussian (a floater close to the camera: thousands of bins) is counted by the whole wave instead of stalling the
    // frame on a single lane (measured: +0.35 ms per frame here and +0.45 ms in gs_make_keys for one such Gaussian among
    // the 1e6 of the headline scene; ordinary waves: 0.139 vs 0.143 ms, the row gathers hide the count).  A wave made
    // of mostly heavy Gaussians (the reference's stress distribution) keeps the per-lane loop: it is already balanced.
    {
        BinWalkRec *recs = s_rec + (threadIdx.x & ~(GS_WAVE - 1));
        int *cnts = s_cnt + (threadIdx.x & ~(GS_WAVE - 1));
        int w0u = t0u, w1u = t1u, w0v = t0v, w1v = t1v;   // the part of the box the walk looks at (gs_common.h)
        if (live && cull) gs_cull_box(u_, v_, cA, cB, cC, qmax, w0u, w1u, w0v, w1v);
        const int npairs = make_walk_rec(s_rec[threadIdx.x], live, u_, v_, cA, cB, cC, qmax, w0u, w1u, w0v, w1v,
                                         bin_shift, ow);
        if (gs_mostly_heavy_wave(npairs)) {
            if (live)
                for_each_emitting_bin(w0u, w1u, w0v, w1v, bin_shift, ow, cull, u_, v_, cA, cB, cC, qmax,
                                      [&](int, int) { ++owned; });
        } else {
            walk_bins_balanced(recs, npairs, bin_shift, ow, cull, [&](int owner, int, int, bool survives) {
                if (survives) atomicAdd(&cnts[owner], 1);
            });
            owned = cnts[gs_lane()];
        }
    }
    // Colour -- only for Gaussians that emit at least one key on this GPU: nothing else is ever gathered by the blend
    // kernels (tile-row sharding: most Gaussians touch the rows of one or two of the G GPUs).  The basis values go through
    // the wave's slice of s_rec, free after the key count.
    float rgb[3] = {0.f, 0.f, 0.f};
    if (COLOUR)   // (else: gs_view_colours fills rgb in behind this kernel, beside the list stages)
        gs_wave_view_colours(live && owned > 0, id, xyz, feat, obj, q_cp, t_cp,
                             reinterpret_cast<float *>(s_rec + (threadIdx.x & ~(GS_WAVE - 1))), rgb);
    if (live) {
        float4 *out = reinterpret_cast<float4 *>(attrs + (size_t)GS_ATTR_STRIDE * i);
        out[0] = make_float4(u_, v_, z_, qmax);  // always: the hook exposes uv and depth of every
                                                 // visible point (RAS:1138-1139)
        if (owned > 0) {
            out[1] = make_float4(cA, cB, cC, radius);
            out[2] = make_float4(rgb[0], rgb[1], rgb[2], opacity);
            // what the blend kernels want next to the conic (gs_blend.hip): amp = opacity * rescale (alpha = amp * exp(e)), the
            // Gaussian's stop-bracket weight (gs_common.h, "threshold decisions") -- and the rescale factor on its own (the
            // opacity is in row 2): the reference multiplies exp(e) by them one after the other (UTL:284, RAS:447), and so does
            // the exact re-evaluation of an alpha next to 1/255; float 14: the exponent below which the reference skips the pair
            // for certain (gs_common.h, "the 1/255 decision in the exponent's domain")
            const float amp = opacity * rescale;
            out[3] = make_float4(amp, gs_stop_weight(amp, 0.0001f), gs_hit_exponent_lo(amp), rescale);
        }
        nkeys[i] = owned;
    }
    // per-block partial sums for the two scans (wave reduce, then one LDS atomic per wave) and the
    // largest quantised depth on screen (lets the host sort only the key bits that are in use)
    int s = owned, sf = full;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_xor(s, d, GS_WAVE);
        sf += __shfl_xor(sf, d, GS_WAVE);
        dq = max(dq, __shfl_xor(dq, d, GS_WAVE));
    }
    if (gs_lane() == 0) {
        if (s != 0) atomicAdd(&s_sum, s);
        if (sf != 0) atomicAdd(&s_sum_full, sf);
        if (dq > 0) atomicMax(&s_dq, dq);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = s_sum;
        block_sums_full[blockIdx.x] = s_sum_full;
        // one global atomic per workgroup, and only while it can still raise the maximum (a stale read
        // merely causes a redundant atomic)
        if (counters != nullptr && s_dq > __hip_atomic_load(&counters[GS_COUNTER_MAX_DEPTH_KEY], __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&counters[GS_COUNTER_MAX_DEPTH_KEY], s_dq);
    }
}


// ------------------------------------------------------------------ colours of the visible points, on their own
// The colour half of gs_preprocess (RAS:280-282,302-310) as a kernel of its own: gs_frame_forward runs it on a second
// stream BESIDE key generation, sort and ranges (none of which read a colour) -- it is a plain streaming pass over the
// 192 B of SH coefficients per Gaussian, the list stages are latency-bound launches that leave the HBM idle.
// Writes floats 8..10 of the record (row 2 = r, g, b, opacity: the opacity is gs_preprocess_geometry's).
__global__ __launch_bounds__(GS_BLOCK) void view_colours_kernel(
    const float *__restrict__ xyz, const float *__restrict__ feat, const int32_t *__restrict__ obj,
    const float *__restrict__ q_cp, const float *__restrict__ t_cp, const int32_t *__restrict__ ids, int m_capacity,
    int use_device_count, const int32_t *__restrict__ counters, const int32_t *__restrict__ nkeys,
    float *__restrict__ attrs) {
    __shared__ float s_basis[GS_BLOCK * 16];
    const int m = use_device_count ? min(counters[GS_COUNTER_NUM_VISIBLE], m_capacity) : m_capacity;
    if (blockIdx.x * GS_BLOCK >= m) return;   // (workgroup-uniform)
    const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
    const bool wants = i < m && nkeys[i] > 0;
    const int id = wants ? ids[i] : 0;
    float rgb[3] = {0.f, 0.f, 0.f};
    gs_wave_view_colours(wants, id, xyz, feat, obj, q_cp, t_cp, s_basis + 16 * (threadIdx.x & ~(GS_WAVE - 1)), rgb);
    if (wants) {
        float *out = attrs + (size_t)GS_ATTR_STRIDE * i + 8;
        *reinterpret_cast<float2 *>(out) = make_float2(rgb[0], rgb[1]);
        out[2] = rgb[2];
    }
}


// ------------------------------------------------------------------ key count of RECEIVED records (owner-sharded Gaussians)
// A band that receives projected records from the other ranks (gs_route_scatter; chunks of `chunk` 64-B slots per sending
// rank, slot 0 of a chunk = header {number of valid records as int32 bits}) needs what gs_preprocess produces next to a
// record: the number of sort keys it emits on THIS rank's tile rows (the walk of gs_make_keys on the same stored values),
// the reference's box count (RAS:106-128: its scan gives the backward slots) and the largest quantised depth.  Header
// slots and the unused tail of a chunk count as Gaussians without keys: the chunked buffer is used as the `attrs` array of
// every later stage as it lies, in (sending rank, position) order = ascending global point id.
__global__ __launch_bounds__(GS_BLOCK) void count_keys_kernel(
    const float *__restrict__ records, int n_slots, int chunk, int width, int height, RowOwner ow, int bin_shift, int cull,
    float depth_scale, int32_t *__restrict__ counters, int32_t *__restrict__ ntiles_full, int32_t *__restrict__ nkeys,
    int32_t *__restrict__ block_sums, int32_t *__restrict__ block_sums_full) {
    __shared__ int s_sum, s_sum_full, s_dq;
    __shared__ BinWalkRec s_rec[GS_BLOCK];
    __shared__ int s_cnt[GS_BLOCK];
    if (threadIdx.x == 0) { s_sum = 0; s_sum_full = 0; s_dq = 0; }
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
    bool live = false;
    if (i < n_slots) {
        const int src = i / chunk, j = i - src * chunk;
        const int valid = __builtin_bit_cast(int, records[(size_t)GS_ATTR_STRIDE * src * chunk]);
        live = j >= 1 && j <= valid;
    }
    int owned = 0, full = 0, dq = 0;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    int t0u = 0, t1u = 0, t0v = 0, t1v = 0;
    if (live) {
        a0 = reinterpret_cast<const float4 *>(records + (size_t)GS_ATTR_STRIDE * i)[0];
        a1 = reinterpret_cast<const float4 *>(records + (size_t)GS_ATTR_STRIDE * i)[1];
        tile_box(a0.x, a0.y, a1.w, width / GS_TILE_WIDTH, height / GS_TILE_HEIGHT, t0u, t1u, t0v, t1v);
        full = (t1u - t0u) * (t1v - t0v);
        dq = (int32_t)(a0.z * depth_scale);
        if (cull) gs_cull_box(a0.x, a0.y, a1.x, a1.y, a1.z, a0.w, t0u, t1u, t0v, t1v);   // as gs_preprocess / gs_make_keys
    }
    {
        BinWalkRec *recs = s_rec + (threadIdx.x & ~(GS_WAVE - 1));
        int *cnts = s_cnt + (threadIdx.x & ~(GS_WAVE - 1));
        const int npairs = make_walk_rec(s_rec[threadIdx.x], live, a0.x, a0.y, a1.x, a1.y, a1.z, a0.w, t0u, t1u, t0v, t1v,
                                         bin_shift, ow);
        if (gs_mostly_heavy_wave(npairs)) {
            if (live)
                for_each_emitting_bin(t0u, t1u, t0v, t1v, bin_shift, ow, cull, a0.x, a0.y, a1.x, a1.y, a1.z, a0.w,
                                      [&](int, int) { ++owned; });
        } else {
            walk_bins_balanced(recs, npairs, bin_shift, ow, cull, [&](int owner, int, int, bool survives) {
                if (survives) atomicAdd(&cnts[owner], 1);
            });
            owned = cnts[gs_lane()];
        }
    }
    if (i < n_slots) {
        ntiles_full[i] = full;
        nkeys[i] = owned;
    }
    int s = owned, sf = full;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s += __shfl_xor(s, d, GS_WAVE);
        sf += __shfl_xor(sf, d, GS_WAVE);
        dq = max(dq, __shfl_xor(dq, d, GS_WAVE));
    }
    if (gs_lane() == 0) {
        if (s != 0) atomicAdd(&s_sum, s);
        if (sf != 0) atomicAdd(&s_sum_full, sf);
        if (dq > 0) atomicMax(&s_dq, dq);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = s_sum;
        block_sums_full[blockIdx.x] = s_sum_full;
        if (s_dq > __hip_atomic_load(&counters[GS_COUNTER_MAX_DEPTH_KEY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&counters[GS_COUNTER_MAX_DEPTH_KEY], s_dq);
        if (blockIdx.x == 0) counters[GS_COUNTER_NUM_VISIBLE] = n_slots;   // what the later stages index by
    }
}

// ------------------------------------------------------------------ key generation
// RAS:131-172 generate_point_sort_key_by_num_overlap_tiles
// KeyT = uint64_t: reference layout (tile << 32) + int32 depth.  KeyT = uint32_t: compressed layout
// (tile << key_depth_bits) | depth, used when the quantised depth is known to be non-negative and
// tile and depth fit 32 bits together (same order, half the sort traffic).
template <typename KeyT>
__global__ __launch_bounds__(GS_BLOCK) void make_keys_kernel(
    const float *__restrict__ attrs, const int32_t *__restrict__ nkeys,
    const int32_t *__restrict__ block_offsets, int m_capacity, const int32_t *__restrict__ counters,
    long long n_keys_capacity, int width, int height, RowOwner ow, int bin_shift,
    int cull, int key_depth_bits, float depth_scale, KeyT *__restrict__ keys, int32_t *__restrict__ payload,
    const int32_t *__restrict__ ntiles_full, const int32_t *__restrict__ block_offsets_full,
    int32_t *__restrict__ slot_offsets) {
    __shared__ int lds[4];
    __shared__ BinWalkRec s_rec[GS_BLOCK];
    __shared__ int s_dq[GS_BLOCK];
    // sizes may still be on their way to the host: the visible count is read on the device, writes stop at the capacity
    const int m = counters ? min(counters[GS_COUNTER_NUM_VISIBLE], m_capacity) : m_capacity;
    const int i = blockIdx.x * GS_BLOCK + threadIdx.x;
    int cnt = i < m ? nkeys[i] : 0;
    int total;
    int offset = block_offsets[blockIdx.x] + gs_block_excl_scan(cnt, &total, lds);
    if (slot_offsets != nullptr) {   // exclusive scan of the reference's box counts = slot base of every Gaussian (RAS:913-922)
        const int full = i < m ? ntiles_full[i] : 0;
        const int so = block_offsets_full[blockIdx.x] + gs_block_excl_scan(full, &total, lds);
        if (i < m) slot_offsets[i] = so;
    }
    const int tw = width / GS_TILE_WIDTH;
    const int bins_u = (tw + (1 << bin_shift) - 1) >> bin_shift;
    const uint32_t depth_mask = key_depth_bits > 0 ? (1u << key_depth_bits) - 1u : 0xffffffffu;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    int t0u = 0, t1u = 0, t0v = 0, t1v = 0;
    if (cnt > 0) {
        a0 = reinterpret_cast<const float4 *>(attrs + (size_t)GS_ATTR_STRIDE * i)[0];
        a1 = reinterpret_cast<const float4 *>(attrs + (size_t)GS_ATTR_STRIDE * i)[1];
        tile_box(a0.x, a0.y, a1.w, tw, height / GS_TILE_HEIGHT, t0u, t1u, t0v, t1v);
        if (cull) gs_cull_box(a0.x, a0.y, a1.x, a1.y, a1.z, a0.w, t0u, t1u, t0v, t1v);   // as gs_preprocess
    }
    // The same walk on the same stored values as gs_preprocess: both kernels agree on which pairs survive.  A wave of
    // ordinary Gaussians (a few dozen bins each) deals its pairs to its lanes 64 at a time -- survivors are then met in
    // the order of their keys, the wave's first key sits at its lane 0's offset, and consecutive lanes write consecutive
    // keys (0.070 -> 0.056 ms at the headline size against one scattered-store loop per lane) -- and a single
    // screen-filling Gaussian is written by its whole wave.  A wave of mostly heavy Gaussians (> 256 bins each) keeps
    // the per-lane loop: the balanced walk costs ~2.5x per pair (owner search, record fetch) and such waves are
    // balanced as they are (the reference's stress scene: 0.15 vs 0.34 ms).
    BinWalkRec *recs = s_rec + (threadIdx.x & ~(GS_WAVE - 1));
    int *dqs = s_dq + (threadIdx.x & ~(GS_WAVE - 1));
    s_dq[threadIdx.x] = (int32_t)(a0.z * depth_scale);  // truncation toward zero, RAS:159-160
    const int npairs = make_walk_rec(s_rec[threadIdx.x], cnt > 0, a0.x, a0.y, a1.x, a1.y, a1.z, a0.w, t0u, t1u, t0v, t1v,
                                     bin_shift, ow);
    if (gs_mostly_heavy_wave(npairs)) {   // per-lane walk: every lane writes its own keys from its own offset
        if (cnt == 0) return;
        const int32_t dq = (int32_t)(a0.z * depth_scale);
        long long k = offset;
        for_each_emitting_bin(t0u, t1u, t0v, t1v, bin_shift, ow, cull, a0.x, a0.y, a1.x, a1.y, a1.z, a0.w,
                              [&](int bu, int bv) {
            const int32_t bin = bu + bv * bins_u;
            if (k < n_keys_capacity) {   // (an overflowing frame is detected by the host from the counters and redone)
                if (sizeof(KeyT) == 8)
                    keys[k] = (KeyT)((int64_t)dq + ((int64_t)bin << 32));
                else   // (a speculative frame whose depth range outgrew the field is redone by the host: the masked bits only
                       //  keep its discarded keys inside the bin range, so that gs_tile_ranges stays inside its arrays)
                    keys[k] = (KeyT)(((uint32_t)bin << key_depth_bits) | ((uint32_t)dq & depth_mask));
                payload[k] = i;
            }
            ++k;
        });
        return;
    }
    long long next = __builtin_amdgcn_readfirstlane(offset);
    const int first_point = blockIdx.x * GS_BLOCK + (threadIdx.x & ~(GS_WAVE - 1));
    walk_bins_balanced(recs, npairs, bin_shift, ow, cull, [&](int owner, int bu, int bv, bool survives) {
        const unsigned long long alive = __builtin_amdgcn_ballot_w64(survives);
        const long long k = next + gs_mbcnt(alive);
        if (survives && k < n_keys_capacity) {   // (an overflowing frame is detected by the host from the counters and redone)
            const int32_t bin = bu + bv * bins_u, dq = dqs[owner];
            if (sizeof(KeyT) == 8)
                keys[k] = (KeyT)((int64_t)dq + ((int64_t)bin << 32));
            else
                keys[k] = (KeyT)(((uint32_t)bin << key_depth_bits) | ((uint32_t)dq & depth_mask));
            payload[k] = first_point + owner;
        }
        next += __popcll(alive);
    });
}

// RAS:175-193 find_tile_start_and_end (arrays pre-zeroed by the caller entry point)
template <typename KeyT>
__device__ __forceinline__ int32_t tile_of_key(KeyT key, int key_depth_bits) {
    if (sizeof(KeyT) == 8) return (int32_t)((int64_t)key >> 32);
    return (int32_t)((uint32_t)key >> key_depth_bits);
}
constexpr int RANGES_PER_THREAD = 4;   // consecutive keys per thread (one 16-B / 32-B run): 4x fewer, fatter waves
template <typename KeyT>
__global__ void tile_ranges_kernel(const KeyT *__restrict__ keys, long long n, const int32_t *__restrict__ n_device,
                                   int key_depth_bits, int n_tiles, int32_t *__restrict__ tile_start,
                                   int32_t *__restrict__ tile_end) {
    if (n_device) n = min((long long)*n_device, n);
    const long long first = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * RANGES_PER_THREAD;
    if (first >= n) return;
    int32_t t = tile_of_key<KeyT>(keys[first], key_depth_bits);
#pragma unroll
    for (int k = 0; k < RANGES_PER_THREAD; ++k) {
        const long long i = first + k;
        if (i >= n) break;
        if (i + 1 < n) {
            const int32_t tn = tile_of_key<KeyT>(keys[i + 1], key_depth_bits);
            if (t != tn) {   // (bin ids outside the arrays -- keys the caller did not generate -- are never written)
                if ((unsigned)tn < (unsigned)n_tiles) tile_start[tn] = (int32_t)(i + 1);
                if ((unsigned)t < (unsigned)n_tiles) tile_end[t] = (int32_t)(i + 1);
            }
            t = tn;
        } else if ((unsigned)t < (unsigned)n_tiles) {
            tile_end[t] = (int32_t)n;
        }
    }
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int gs_pose_inverse(const float *q_pc, const float *t_pc, float *q_cp, float *t_cp, int n_obj, void *stream) {
    GS_REQUIRE(n_obj > 0, "n_obj must be positive");
    hipLaunchKernelGGL(pose_inverse_kernel, dim3(gs_div_up(n_obj, 64)), dim3(64), 0, (hipStream_t)stream,
                       q_pc, t_pc, q_cp, t_cp, n_obj);
    GS_CHECK_LAUNCH();
    return 0;
}

size_t gs_filter_workspace_bytes(int n_points) {
    return sizeof(int32_t) * ((size_t)gs_div_up(n_points > 0 ? n_points : 1, FILTER_ITEMS) + 64);
}

int gs_filter_compact(const float *xyz, const int8_t *invalid_mask, const int32_t *object_id,
                      const float *intrinsics, const float *q_cp, const float *t_cp, int n_points,
                      float near_plane, float far_plane, int width, int height, int8_t *mask, int32_t *ids,
                      int32_t *counters, void *workspace, void *stream) {
    return gs_filter_compact_from_poses(xyz, invalid_mask, object_id, intrinsics, nullptr, nullptr, 0, (float *)q_cp,
                                        (float *)t_cp, n_points, near_plane, far_plane, width, height, mask, ids, counters,
                                        workspace, stream);
}

int gs_filter_compact_from_poses(const float *xyz, const int8_t *invalid_mask, const int32_t *object_id,
                                 const float *intrinsics, const float *q_pointcloud_camera,
                                 const float *t_pointcloud_camera, int n_objects, float *q_cp, float *t_cp, int n_points,
                                 float near_plane, float far_plane, int width, int height, int8_t *mask, int32_t *ids,
                                 int32_t *counters, void *workspace, void *stream) {
    GS_REQUIRE(n_points >= 0, "n_points");
    GS_REQUIRE(q_pointcloud_camera == nullptr || (t_pointcloud_camera != nullptr && n_objects > 0), "poses");
    GS_REQUIRE(width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0, "image size must be a multiple of 16");
    hipStream_t s = (hipStream_t)stream;
    int32_t *block_counts = (int32_t *)workspace;
    if (n_points == 0) {
        GS_CHECK_HIP(hipMemsetAsync(counters, 0, sizeof(int32_t) * GS_NUM_COUNTERS, s));
        if (q_pointcloud_camera != nullptr)   // the later stages of the frame still expect the inverted poses
            return gs_pose_inverse(q_pointcloud_camera, t_pointcloud_camera, q_cp, t_cp, n_objects, stream);
        return 0;
    }
    const int nblk = gs_div_up(n_points, FILTER_ITEMS);
    hipLaunchKernelGGL(filter_kernel, dim3(nblk), dim3(GS_BLOCK), 0, s, xyz, invalid_mask, object_id, intrinsics,
                       q_cp, t_cp, n_points, near_plane, far_plane, width, height, mask, block_counts, counters,
                       q_pointcloud_camera, t_pointcloud_camera, n_objects, q_cp, t_cp);
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(compact_kernel, dim3(nblk), dim3(GS_BLOCK), 0, s, mask, n_points, block_counts, ids,
                       counters + GS_COUNTER_NUM_VISIBLE);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_read_counters(const int32_t *counters, int32_t *host_counters, int n, void *stream) {
    GS_REQUIRE(n > 0 && n <= GS_NUM_COUNTERS, "n");
    GS_CHECK_HIP(hipMemcpyAsync(host_counters, counters, sizeof(int32_t) * n, hipMemcpyDeviceToHost,
                                (hipStream_t)stream));
    GS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int gs_read_counters_async(const int32_t *counters, int32_t *host_counters_pinned, int n, void *stream) {
    GS_REQUIRE(n > 0 && n <= GS_NUM_COUNTERS, "n");
    GS_CHECK_HIP(hipMemcpyAsync(host_counters_pinned, counters, sizeof(int32_t) * n, hipMemcpyDeviceToHost,
                                (hipStream_t)stream));
    return 0;
}

static int gs_preprocess_launch(bool colour, const float *xyz, float *features, const int32_t *object_id,
                                const float *intrinsics, const float *q_cp, const float *t_cp, const int32_t *ids,
                                int n_visible, int n_visible_on_device, int width, int height, int tile_row_begin,
                                int tile_row_step, int tile_row_end, int bin_shift, int exact_tile_cull,
                                int always_store_rotation, float depth_scale, int32_t *counters, float *attrs,
                                int32_t *num_overlap_tiles, int32_t *num_keys, int32_t *block_sums,
                                int32_t *block_sums_full, void *stream) {
    GS_REQUIRE(n_visible >= 0, "n_visible");
    GS_REQUIRE(tile_row_step >= 1 && tile_row_begin >= 0 && tile_row_end >= 0, "tile row ownership");
    GS_REQUIRE(bin_shift >= 0 && bin_shift <= 4, "bin_shift");
    GS_REQUIRE(width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0, "image size must be a multiple of 16");
    GS_REQUIRE(!n_visible_on_device || counters != nullptr, "device-side count needs counters");
    if (n_visible == 0) return 0;
    const dim3 grid(gs_div_up(n_visible, GS_BLOCK)), block(GS_BLOCK);
    const RowOwner ow{tile_row_begin, tile_row_step, tile_row_end};
    if (colour)
        hipLaunchKernelGGL(preprocess_kernel<true>, grid, block, 0, (hipStream_t)stream, xyz, features, object_id,
                           intrinsics, q_cp, t_cp, ids, n_visible, n_visible_on_device, width, height, ow, bin_shift,
                           exact_tile_cull, always_store_rotation, depth_scale, counters, attrs, num_overlap_tiles,
                           num_keys, block_sums, block_sums_full);
    else
        hipLaunchKernelGGL(preprocess_kernel<false>, grid, block, 0, (hipStream_t)stream, xyz, features, object_id,
                           intrinsics, q_cp, t_cp, ids, n_visible, n_visible_on_device, width, height, ow, bin_shift,
                           exact_tile_cull, always_store_rotation, depth_scale, counters, attrs, num_overlap_tiles,
                           num_keys, block_sums, block_sums_full);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_preprocess(const float *xyz, float *features, const int32_t *object_id, const float *intrinsics,
                  const float *q_cp, const float *t_cp, const int32_t *ids, int n_visible, int n_visible_on_device,
                  int width, int height, int tile_row_begin, int tile_row_step, int tile_row_end, int bin_shift,
                  int exact_tile_cull, int always_store_rotation, float depth_scale, int32_t *counters,
                  float *attrs, int32_t *num_overlap_tiles, int32_t *num_keys, int32_t *block_sums,
                  int32_t *block_sums_full, void *stream) {
    return gs_preprocess_launch(true, xyz, features, object_id, intrinsics, q_cp, t_cp, ids, n_visible,
                                n_visible_on_device, width, height, tile_row_begin, tile_row_step, tile_row_end, bin_shift,
                                exact_tile_cull, always_store_rotation, depth_scale, counters, attrs, num_overlap_tiles,
                                num_keys, block_sums, block_sums_full, stream);
}

int gs_preprocess_geometry(const float *xyz, float *features, const int32_t *object_id, const float *intrinsics,
                           const float *q_cp, const float *t_cp, const int32_t *ids, int n_visible,
                           int n_visible_on_device, int width, int height, int tile_row_begin, int tile_row_step,
                           int tile_row_end, int bin_shift, int exact_tile_cull, int always_store_rotation,
                           float depth_scale, int32_t *counters, float *attrs, int32_t *num_overlap_tiles,
                           int32_t *num_keys, int32_t *block_sums, int32_t *block_sums_full, void *stream) {
    return gs_preprocess_launch(false, xyz, features, object_id, intrinsics, q_cp, t_cp, ids, n_visible,
                                n_visible_on_device, width, height, tile_row_begin, tile_row_step, tile_row_end, bin_shift,
                                exact_tile_cull, always_store_rotation, depth_scale, counters, attrs, num_overlap_tiles,
                                num_keys, block_sums, block_sums_full, stream);
}

int gs_view_colours(const float *xyz, const float *features, const int32_t *object_id, const float *q_cp,
                    const float *t_cp, const int32_t *ids, int n_visible, int n_visible_on_device,
                    const int32_t *counters, const int32_t *num_keys, float *attrs, void *stream) {
    GS_REQUIRE(n_visible >= 0, "n_visible");
    GS_REQUIRE(!n_visible_on_device || counters != nullptr, "device-side count needs counters");
    if (n_visible == 0) return 0;
    hipLaunchKernelGGL(view_colours_kernel, dim3(gs_div_up(n_visible, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       xyz, features, object_id, q_cp, t_cp, ids, n_visible, n_visible_on_device, counters, num_keys, attrs);
    GS_CHECK_LAUNCH();
    return 0;
}


int gs_count_keys(const float *records, int n_slots, int chunk_slots, int width, int height, int tile_row_begin,
                  int tile_row_step, int tile_row_end, int bin_shift, int exact_tile_cull, float depth_scale,
                  int32_t *counters, int32_t *num_overlap_tiles, int32_t *num_keys, int32_t *block_sums,
                  int32_t *block_sums_full, void *stream) {
    GS_REQUIRE(n_slots >= 0 && chunk_slots >= 1 && n_slots % chunk_slots == 0, "n_slots must be a whole number of chunks");
    GS_REQUIRE(tile_row_step >= 1 && tile_row_begin >= 0 && tile_row_end >= 0, "tile row ownership");
    GS_REQUIRE(bin_shift >= 0 && bin_shift <= 4, "bin_shift");
    GS_REQUIRE(width % GS_TILE_WIDTH == 0 && height % GS_TILE_HEIGHT == 0, "image size must be a multiple of 16");
    GS_REQUIRE(counters != nullptr, "counters");
    if (n_slots == 0) return 0;
    hipLaunchKernelGGL(count_keys_kernel, dim3(gs_div_up(n_slots, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, records,
                       n_slots, chunk_slots, width, height, RowOwner{tile_row_begin, tile_row_step, tile_row_end}, bin_shift,
                       exact_tile_cull, depth_scale, counters, num_overlap_tiles, num_keys, block_sums, block_sums_full);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_scan_block_sums(int32_t *block_sums, int n_blocks, int32_t *counters, int counter_slot, void *stream) {
    GS_REQUIRE(n_blocks >= 0, "n_blocks");
    GS_REQUIRE(counter_slot >= 0 && counter_slot < GS_NUM_COUNTERS, "counter_slot");
    if (n_blocks == 0) {
        GS_CHECK_HIP(hipMemsetAsync(counters + counter_slot, 0, sizeof(int32_t), (hipStream_t)stream));
        return 0;
    }
    hipLaunchKernelGGL(scan_single_block_kernel, dim3(1), dim3(GS_BLOCK), 0, (hipStream_t)stream, block_sums,
                       n_blocks, counters + counter_slot, (int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr,
                       (int32_t *)nullptr, 0u);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_scan_block_sums2(int32_t *block_sums, int32_t *block_sums_full, int n_blocks, int32_t *counters,
                        void *stream) {
    return gs_scan_block_sums2_to_host(block_sums, block_sums_full, n_blocks, counters, nullptr, stream);
}

int gs_scan_block_sums2_to_host(int32_t *block_sums, int32_t *block_sums_full, int n_blocks, int32_t *counters,
                                int32_t *host_counters_mapped, void *stream) {
    GS_REQUIRE(n_blocks >= 0, "n_blocks");
    if (n_blocks == 0) {
        GS_CHECK_HIP(hipMemsetAsync(counters + GS_COUNTER_NUM_KEYS, 0, 2 * sizeof(int32_t), (hipStream_t)stream));
        if (host_counters_mapped != nullptr)
            GS_CHECK_HIP(hipMemcpyAsync(host_counters_mapped, counters, sizeof(int32_t) * GS_NUM_COUNTERS, hipMemcpyDeviceToHost,
                                        (hipStream_t)stream));
        return 0;
    }
    hipLaunchKernelGGL(scan_single_block_kernel, dim3(2), dim3(GS_BLOCK), 0, (hipStream_t)stream, block_sums,
                       n_blocks, counters + GS_COUNTER_NUM_KEYS, block_sums_full, counters + GS_COUNTER_NUM_SLOTS, counters,
                       host_counters_mapped, 0u);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_scan_block_sums2_stamped(int32_t *block_sums, int32_t *block_sums_full, int n_blocks, int32_t *counters,
                                void *host_words, uint32_t stamp, void *stream) {
    GS_REQUIRE(n_blocks >= 0, "n_blocks");
    GS_REQUIRE(host_words != nullptr && (reinterpret_cast<uintptr_t>(host_words) & 7u) == 0, "host_words: 8-byte aligned host memory");
    GS_REQUIRE(stamp != 0u, "stamp must not be 0");
    // (also with no block at all: the kernel then stores the zero totals and the stamped words)
    hipLaunchKernelGGL(scan_single_block_kernel, dim3(2), dim3(GS_BLOCK), 0, (hipStream_t)stream, block_sums,
                       n_blocks, counters + GS_COUNTER_NUM_KEYS, block_sums_full, counters + GS_COUNTER_NUM_SLOTS, counters,
                       reinterpret_cast<int32_t *>(host_words), stamp);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_wait_stamped_sizes(const void *host_words, uint32_t stamp, int64_t timeout_us, int32_t *sizes4) {
    GS_REQUIRE(host_words != nullptr && sizes4 != nullptr && stamp != 0u, "gs_wait_stamped_sizes: arguments");
    const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(host_words);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        unsigned long long v[4];
        bool all = true;
        for (int k = 0; k < 4; ++k) {
            v[k] = w[k];
            all = all && (uint32_t)(v[k] >> 32) == stamp;
        }
        if (all) {
            for (int k = 0; k < 4; ++k) sizes4[k] = (int32_t)(uint32_t)v[k];
            return 0;
        }
        if ((spins & 63u) == 63u) {
            const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
            if (us >= timeout_us) return 1;
            if (us > 2000) std::this_thread::yield();   // a frame's sizes arrive within ~0.2 ms; be polite after 2 ms
        }
    }
}

int gs_host_alloc_coherent(int64_t bytes, void **out) {
    GS_REQUIRE(bytes > 0 && out != nullptr, "gs_host_alloc_coherent: arguments");
    *out = nullptr;
    GS_CHECK_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocCoherent | hipHostMallocMapped));
    memset(*out, 0, (size_t)bytes);
    return 0;
}

int gs_host_free(void *p) {
    if (p != nullptr) GS_CHECK_HIP(hipHostFree(p));
    return 0;
}

int gs_make_keys(const float *attrs, const int32_t *num_keys, const int32_t *block_offsets, int n_visible,
                 const int32_t *counters, int64_t n_keys_capacity, int width, int height, int tile_row_begin,
                 int tile_row_step, int tile_row_end, int bin_shift, int exact_tile_cull, int key_depth_bits,
                 float depth_scale, void *keys, int32_t *payload, const int32_t *num_overlap_tiles,
                 const int32_t *block_offsets_full, int32_t *slot_offsets, void *stream) {
    GS_REQUIRE(n_visible >= 0 && n_keys_capacity >= 0, "n_visible / n_keys_capacity");
    GS_REQUIRE(tile_row_step >= 1 && tile_row_begin >= 0 && tile_row_end >= 0, "tile row ownership");
    GS_REQUIRE(bin_shift >= 0 && bin_shift <= 4, "bin_shift");
    GS_REQUIRE(key_depth_bits >= 0 && key_depth_bits < 32, "key_depth_bits");
    GS_REQUIRE(slot_offsets == nullptr || (num_overlap_tiles != nullptr && block_offsets_full != nullptr),
               "slot_offsets needs num_overlap_tiles and its scanned block sums");
    if (n_visible == 0) return 0;
    const dim3 grid(gs_div_up(n_visible, GS_BLOCK)), block(GS_BLOCK);
    const RowOwner ow{tile_row_begin, tile_row_step, tile_row_end};
    if (key_depth_bits == 0)
        hipLaunchKernelGGL(make_keys_kernel<uint64_t>, grid, block, 0, (hipStream_t)stream, attrs, num_keys,
                           block_offsets, n_visible, counters, (long long)n_keys_capacity, width, height, ow, bin_shift,
                           exact_tile_cull, 0, depth_scale, (uint64_t *)keys, payload, num_overlap_tiles,
                           block_offsets_full, slot_offsets);
    else
        hipLaunchKernelGGL(make_keys_kernel<uint32_t>, grid, block, 0, (hipStream_t)stream, attrs, num_keys,
                           block_offsets, n_visible, counters, (long long)n_keys_capacity, width, height, ow, bin_shift,
                           exact_tile_cull, key_depth_bits, depth_scale, (uint32_t *)keys, payload, num_overlap_tiles,
                           block_offsets_full, slot_offsets);
    GS_CHECK_LAUNCH();
    return 0;
}

int gs_tile_ranges(const void *keys_sorted, int64_t n_keys, const int32_t *n_keys_device, int key_depth_bits,
                   int32_t *tile_start, int32_t *tile_end, int n_tiles /* number of bins */, void *stream) {
    return gs_tile_ranges_prezeroed(keys_sorted, n_keys, n_keys_device, key_depth_bits, tile_start, tile_end, n_tiles, 0,
                                    stream);
}

int gs_tile_ranges_prezeroed(const void *keys_sorted, int64_t n_keys, const int32_t *n_keys_device, int key_depth_bits,
                             int32_t *tile_start, int32_t *tile_end, int n_tiles /* number of bins */,
                             int ranges_are_zeroed, void *stream) {
    GS_REQUIRE(n_keys >= 0 && n_tiles > 0, "sizes");
    GS_REQUIRE(key_depth_bits >= 0 && key_depth_bits < 32, "key_depth_bits");
    hipStream_t s = (hipStream_t)stream;
    if (ranges_are_zeroed) {
        // (gs_sort_pairs_and_zero filled them with its first launch)
    } else if (tile_end == tile_start + n_tiles) {  // adjacent halves of one buffer: a single fill
        GS_CHECK_HIP(hipMemsetAsync(tile_start, 0, 2 * sizeof(int32_t) * n_tiles, s));
    } else {
        GS_CHECK_HIP(hipMemsetAsync(tile_start, 0, sizeof(int32_t) * n_tiles, s));
        GS_CHECK_HIP(hipMemsetAsync(tile_end, 0, sizeof(int32_t) * n_tiles, s));
    }
    if (n_keys == 0) return 0;
    const dim3 grid(gs_div_up(n_keys, (int64_t)GS_BLOCK * RANGES_PER_THREAD)), block(GS_BLOCK);
    if (key_depth_bits == 0)
        hipLaunchKernelGGL(tile_ranges_kernel<uint64_t>, grid, block, 0, s, (const uint64_t *)keys_sorted,
                           (long long)n_keys, n_keys_device, 0, n_tiles, tile_start, tile_end);
    else
        hipLaunchKernelGGL(tile_ranges_kernel<uint32_t>, grid, block, 0, s, (const uint32_t *)keys_sorted,
                           (long long)n_keys, n_keys_device, key_depth_bits, n_tiles, tile_start, tile_end);
    GS_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
