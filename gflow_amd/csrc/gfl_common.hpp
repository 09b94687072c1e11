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

// where pixel p is sampled (include/gflow_hip.h: GFL_PIXEL_CENTER).  With the default 0 this is (float)p and the kernels are
// instruction for instruction what they were without the constant.
constexpr float PIXEL_CENTER = GFL_PIXEL_CENTER;
__host__ __device__ __forceinline__ float pixf(int p) {
    if constexpr (PIXEL_CENTER != 0.0f) return (float)p + PIXEL_CENTER;
    return (float)p;
}
// a splat centre as the pixel GRID sees it (tests of a centre against boxes of pixel indices)
__host__ __device__ __forceinline__ float gridf(float u) {
    if constexpr (PIXEL_CENTER != 0.0f) return u - PIXEL_CENTER;
    return u;
}
// tile sort: lists longer than SORT_SPLIT_MIN keys are cut at a pivot and sorted by two workgroups; at most SORT_MAX_SPLIT
// tiles per launch (gfl_tile_sort.hpp; the list of such tiles is the trailer of the sort's order: gfl_fused.hip build_sort_order)
#ifndef GFL_SORT_SPLIT_MIN
#define GFL_SORT_SPLIT_MIN 768
#endif
constexpr int SORT_SPLIT_MIN = GFL_SORT_SPLIT_MIN;
constexpr int SORT_MAX_SPLIT = 64;
constexpr int SORT_ORDER_TRAILER = SORT_MAX_SPLIT + 4;      // ints behind order[T][4]: count, positions
constexpr int SLOT_MAX = 32;   // fused backward: list positions kept per splat (tile rect <= 32 tiles)

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

// XCD-aware block order.  The dispatcher is observed to place workgroup b on XCD b % 8
// (MI355X_MICROARCH.md, "Workgroup dispatch"); each XCD has its own 4 MB L2.  Kernels whose
// neighbouring blocks share data (stencil halos, splat records of adjacent tiles) therefore walk
// their work in this order: XCD x owns one contiguous range of logical block ids.  Speed only --
// the map is a bijection on [0, nb) whatever the real placement is.
__device__ __forceinline__ int xcd_logical_block(int b, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int x = b & 7, i = b >> 3;
    return x * q + min(x, r) + i;
}

// Reduce-scatter of ten per-lane values over the wave (blend backward: five conic/centre moments, do, df0..3).
// gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves / rows between two
// registers, so "swap + add" halves the number of live values while summing lane pairs:
//   stage 1  10 -> 5 values (lanes i, i+32),  stage 2  5 -> 3 values (rows 2k, 2k+1),
//   stage 3  the three values folded over the 16 lanes of a row (9 ops, below).
//   ~27 VALU ops instead of 10 x 6 dependent DPP adds.
// Afterwards, in row r = (hi, odd) = (lane>>5, (lane>>4)&1) every lane holds
//   x0 = sum of component 2*odd + hi,  x1 = sum of component 4 + 2*odd + hi,
//   x2 = sum of component 8 + hi (even rows only).
// Returns the value this lane ends up with; reduce_scatter10_component(lane) says which component it is.
// NOTE (ROCm 7.2 hipcc): __builtin_amdgcn_permlane{16,32}_swap returns the updated vdst in BOTH
// elements of its result (verified on hardware), so the swaps are issued as inline asm; the
// leading s_nop covers the VALU-write -> permlane-read hazard that hipcc does not pad inside asm.
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// five (three) independent swaps behind ONE hazard nop: the operands of all of them are written
// before the block, and no swap reads what another one wrote
__device__ __forceinline__ void permlane32_swap_x5(float (&a)[5], float (&b)[5]) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %5\n\tv_permlane32_swap_b32 %1, %6\n\tv_permlane32_swap_b32 %2, %7\n\t"
        "v_permlane32_swap_b32 %3, %8\n\tv_permlane32_swap_b32 %4, %9"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]));
}
__device__ __forceinline__ void permlane16_swap_x3(float (&a)[3], float (&b)[3]) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %3\n\tv_permlane16_swap_b32 %1, %4\n\tv_permlane16_swap_b32 %2, %5"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
}

__device__ __forceinline__ float wave_reduce_scatter10(const float (&v)[10], int lane) {
    float w[5];
    {
        float a[5] = {v[0], v[2], v[4], v[6], v[8]}, b[5] = {v[1], v[3], v[5], v[7], v[9]};
        permlane32_swap_x5(a, b);     // a = {a_lo, b_lo}, b = {a_hi, b_hi}
#pragma unroll
        for (int m = 0; m < 5; ++m) w[m] = a[m] + b[m];   // lanes 0-31: component 2m, lanes 32-63: component 2m+1
    }
    float x[3];
    {
        float a[3] = {w[0], w[2], w[4]}, b[3] = {w[1], w[3], 0.f};
        permlane16_swap_x3(a, b);     // a = {a_r0, b_r0, a_r2, b_r2}, b = {a_r1, b_r1, a_r3, b_r3}
        x[0] = a[0] + b[0];
        x[1] = a[1] + b[1];
        x[2] = a[2] + b[2];
    }
    // stage 3: the three values of a 16-lane row.  Instead of three independent 4-step reductions
    // (12 DPP adds) the values are folded onto lane classes on the way: after the xor-1 step the
    // even lanes carry x0 and the odd lanes x1, after the xor-2 step lanes with bit 1 set carry x2;
    // two rotations by 4 and 8 lanes (which keep the low two lane bits) finish the sums: 9 ops.
#define GFL_DPP(val, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (val)), ctrl, 0xF, 0xF, true))
    const bool b0 = lane & 1, b1 = lane & 2;
    const float keep01 = b0 ? x[1] : x[0], send01 = b0 ? x[0] : x[1];
    const float y = keep01 + GFL_DPP(send01, 0xB1);          // quad_perm [1,0,3,2]
    const float z = x[2] + GFL_DPP(x[2], 0xB1);
    const float keep = b1 ? z : y, send = b1 ? y : z;
    float t = keep + GFL_DPP(send, 0x4E);                    // quad_perm [2,3,0,1]
    t += GFL_DPP(t, 0x124);                                  // row_ror:4
    t += GFL_DPP(t, 0x128);                                  // row_ror:8
#undef GFL_DPP
    // lane & 3 == 0: x0 total, == 1: x1 total, >= 2: x2 total
    return t;
}

// which of the ten components the value returned to this lane belongs to (-1: a duplicate, not to
// be used).  Depends on the lane only: callers evaluate it ONCE, outside their loops (inside, the
// compiler turned the nested selects into a dozen exec-mask instructions per call).
__device__ __forceinline__ int reduce_scatter10_component(int lane) {
    const int hi = lane >> 5, odd = (lane >> 4) & 1, sel = lane & 15;
    return sel == 0 ? 2 * odd + hi : (sel == 1 ? 4 + 2 * odd + hi : ((sel == 2 && odd == 0) ? 8 + hi : -1));
}

// The same for SIX values (camera-only stage of the backward blend: the five moments and the depth feature's gradient --
// nobody reads the opacity / colour gradients of frozen splats): 6 -> 3 -> 2 values, then the two folded over the 16 lanes
// of a row: 16 VALU ops instead of 27.  v = {c0, c1, c2, c3, c4, c5}; afterwards, in row (hi, odd), lanes with bit 0
// clear hold the sum of component 2 * odd + hi, lanes with bit 0 set the sum of component 4 + hi (even rows only).
__device__ __forceinline__ void permlane32_swap_x3(float (&a)[3], float (&b)[3]) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %3\n\tv_permlane32_swap_b32 %1, %4\n\tv_permlane32_swap_b32 %2, %5"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
}
__device__ __forceinline__ void permlane16_swap_x2(float (&a)[2], float (&b)[2]) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3"
        : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
}
__device__ __forceinline__ float wave_reduce_scatter6(const float (&v)[6], int lane) {
    float w[3];
    {
        float a[3] = {v[0], v[2], v[4]}, b[3] = {v[1], v[3], v[5]};
        permlane32_swap_x3(a, b);
#pragma unroll
        for (int m = 0; m < 3; ++m) w[m] = a[m] + b[m];   // lanes 0-31: component 2m, lanes 32-63: component 2m+1
    }
    float x[2];
    {
        float a[2] = {w[0], w[2]}, b[2] = {w[1], 0.f};
        permlane16_swap_x2(a, b);
        x[0] = a[0] + b[0];                                // even rows: w0, odd rows: w1
        x[1] = a[1] + b[1];                                // even rows: w2
    }
#define GFL_DPP(val, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (val)), ctrl, 0xF, 0xF, true))
    const bool b0 = lane & 1;
    const float keep = b0 ? x[1] : x[0], send = b0 ? x[0] : x[1];
    float t = keep + GFL_DPP(send, 0xB1);                  // quad_perm [1,0,3,2]
    t += GFL_DPP(t, 0x4E);                                 // quad_perm [2,3,0,1]
    t += GFL_DPP(t, 0x124);                                // row_ror:4
    t += GFL_DPP(t, 0x128);                                // row_ror:8
#undef GFL_DPP
    return t;
}
__device__ __forceinline__ int reduce_scatter6_component(int lane) {
    const int hi = lane >> 5, odd = (lane >> 4) & 1, sel = lane & 15;
    return sel == 0 ? 2 * odd + hi : ((sel == 1 && odd == 0) ? 4 + hi : -1);
}

// ... and for SEVEN (later frames: the colours are frozen, trainer.py:537-540 -- the moments, the opacity's and the depth
// feature's gradient remain): eight slots {c0 .. c6, 0}, 8 -> 4 -> 2 values, then the two folded over a row: 18 VALU ops.
// Afterwards lanes with bit 0 clear hold component 2 * odd + hi, lanes with bit 0 set component 4 + hi in even rows and
// component 6 in row (hi = 0, odd = 1).
__device__ __forceinline__ void permlane32_swap_x4(float (&a)[4], float (&b)[4]) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
        "v_permlane32_swap_b32 %3, %7"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ float wave_reduce_scatter7(const float (&v)[7], int lane) {
    float w[4];
    {
        float a[4] = {v[0], v[2], v[4], v[6]}, b[4] = {v[1], v[3], v[5], 0.f};
        permlane32_swap_x4(a, b);
#pragma unroll
        for (int m = 0; m < 4; ++m) w[m] = a[m] + b[m];   // lanes 0-31: component 2m, lanes 32-63: component 2m+1
    }
    float x[2];
    {
        float a[2] = {w[0], w[2]}, b[2] = {w[1], w[3]};
        permlane16_swap_x2(a, b);
        x[0] = a[0] + b[0];                                // even rows: w0, odd rows: w1
        x[1] = a[1] + b[1];                                // even rows: w2, odd rows: w3
    }
#define GFL_DPP(val, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (val)), ctrl, 0xF, 0xF, true))
    const bool b0 = lane & 1;
    const float keep = b0 ? x[1] : x[0], send = b0 ? x[0] : x[1];
    float t = keep + GFL_DPP(send, 0xB1);                  // quad_perm [1,0,3,2]
    t += GFL_DPP(t, 0x4E);                                 // quad_perm [2,3,0,1]
    t += GFL_DPP(t, 0x124);                                // row_ror:4
    t += GFL_DPP(t, 0x128);                                // row_ror:8
#undef GFL_DPP
    return t;
}
__device__ __forceinline__ int reduce_scatter7_component(int lane) {
    const int hi = lane >> 5, odd = (lane >> 4) & 1, sel = lane & 15;
    return sel == 0 ? 2 * odd + hi : (sel == 1 ? (odd == 0 ? 4 + hi : (hi == 0 ? 6 : -1)) : -1);
}

// Deterministic block-level reduction of NV values -> one partial row per block.
// partial[blockIdx.x * NV + k]; a second tiny kernel folds the rows in order.
// AGENT: the row is stored with agent-scope atomic stores (write-through), for a reader in ANOTHER workgroup of the
// same launch (camera_tail of gfl_fused.hip).
template <int NV, int BLOCK, bool AGENT = false>
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
        if (AGENT) __hip_atomic_store(&partial[(size_t)blockIdx.x * NV + threadIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partial[(size_t)blockIdx.x * NV + threadIdx.x] = s;
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
