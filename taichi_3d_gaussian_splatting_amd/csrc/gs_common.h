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

__device__ __forceinline__ int gs_lane() { return threadIdx.x & (GS_WAVE - 1); }

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
// gfx950 cross-half / cross-row swaps (v_permlane32_swap / v_permlane16_swap).  Written as inline
// asm because ROCm 7.2's clang returns element 0 for BOTH results of the __builtin_amdgcn_permlane*_swap
// builtins.  The s_nop's cover the VALU-write -> permlane-read and permlane-write -> DPP-read wait states,
// which the hazard recogniser cannot see through an asm statement.
// (semantics verified on hardware: v_permlane32_swap exchanges vdst[63:32] with src0[31:0];
// v_permlane16_swap exchanges the odd rows of vdst with the even rows of src0)
// fold32: returns r with  r[lane<32] = x[l]+x[l+32],  r[lane>=32] = y[l-32]+y[l]
__device__ __forceinline__ float gs_fold32(float x, float y) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
// fold16: even rows of the result hold x[row]+x[row+1], odd rows hold y[row-1]+y[row] (rows of 16 lanes)
__device__ __forceinline__ float gs_fold16(float x, float y) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float gs_readlane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

#endif  // __HIPCC__
