// Shared device/host helpers for libgflow_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gflow_hip.h"

namespace gfl {

extern thread_local int g_last_hip_error;

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return GFL_ERR_HIP;
    }
    return GFL_OK;
}
inline int check(hipError_t e) {
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return GFL_ERR_HIP;
    }
    return GFL_OK;
}

constexpr int WAVE = 64;

// Camera in wave-uniform registers: the 16 floats are read with scalar loads.
struct Cam {
    float fx, fy, cx, cy;
    float r00, r01, r02, t0;
    float r10, r11, r12, t1;
    float r20, r21, r22, t2;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ intr, const float* __restrict__ extr) {
    Cam c;
    c.fx = intr[0]; c.fy = intr[1]; c.cx = intr[2]; c.cy = intr[3];
    c.r00 = extr[0]; c.r01 = extr[1]; c.r02 = extr[2]; c.t0 = extr[3];
    c.r10 = extr[4]; c.r11 = extr[5]; c.r12 = extr[6]; c.t1 = extr[7];
    c.r20 = extr[8]; c.r21 = extr[9]; c.r22 = extr[10]; c.t2 = extr[11];
    return c;
}

// ---- wave64 reductions on DPP (no LDS traffic) -------------------------------
// After the six steps lane 63 holds the sum of all 64 lanes.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    // quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
    // row_mirror = 0x140, row_bcast:15 = 0x142, row_bcast:31 = 0x143
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, true));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = wave_sum_to_lane63(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Deterministic block-level reduction of NV values -> one partial row per block.
// partial[blockIdx.x * NV + k]; a second tiny kernel folds the rows in order.
template <int NV, int BLOCK>
__device__ __forceinline__ void block_reduce_store(float (&vals)[NV], float* __restrict__ partial) {
    __shared__ float red[BLOCK / WAVE][NV];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x / WAVE;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float s = wave_sum_to_lane63(vals[k]);
        if (lane == 63) red[wid][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; ++w) s += red[w][threadIdx.x];
        partial[(size_t)blockIdx.x * NV + threadIdx.x] = s;
    }
}

template <int NV>
__global__ void __launch_bounds__(256) fold_partials_kernel(const float* __restrict__ partial, int rows,
                                                            float* __restrict__ out) {
    // one block; thread t folds rows t, t+256, ... in order, then a fixed-shape
    // wave/LDS tree combines the 256 partials: bitwise reproducible run to run.
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] += partial[(size_t)r * NV + k];
    }
    __shared__ float red[4][NV];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wid = threadIdx.x / WAVE;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float s = wave_sum_to_lane63(acc[k]);
        if (lane == 63) red[wid][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) out[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

constexpr int REDUCE_BLOCK = 256;
inline int reduce_rows(int N) { return (N + REDUCE_BLOCK - 1) / REDUCE_BLOCK; }

}  // namespace gfl
