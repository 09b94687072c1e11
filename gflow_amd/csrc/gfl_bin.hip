// Tile binning and per-tile depth sort (msplat.sort_gaussian, render.py:52-54).
//
// MI355X-first design: no global radix sort.  A frame has ~1.6k tiles holding ~100
// splats each, so the (tile, depth) order is produced by
//   1. count   : one lane per splat, one L2 atomic per covered tile;
//   2. scan    : one workgroup, exclusive scan of the T tile counters;
//   3. scatter : one lane per splat, slot = atomic cursor of the tile, writes a
//                64-bit key (depth bits << 32 | splat id) into the tile's segment;
//   4. sort    : one workgroup per tile, bitonic network on the segment staged in
//                LDS (global memory fallback for segments over 4096 keys).
// Keys are unique, so the result is independent of the atomic arrival order
// (bitwise reproducible) and equals a stable sort by (tile, depth) of id-ordered
// pairs.  Every launch is sized by N or T, never by the data-dependent K, so the
// whole sequence is graph-capturable with no host read-back.
#include "gfl_math.hpp"

namespace gfl {


__device__ __forceinline__ bool tile_hit(float u, float v, float cutoff, int tx, int ty) {
    // exact-disc test: distance from the splat centre to the tile's pixel-centre
    // box [16tx, 16tx+15] x [16ty, 16ty+15]
    const float x_lo = (float)(tx * GFL_TILE), x_hi = x_lo + (float)(GFL_TILE - 1);
    const float y_lo = (float)(ty * GFL_TILE), y_hi = y_lo + (float)(GFL_TILE - 1);
    u = gridf(u); v = gridf(v);
    const float ddx = fmaxf(fmaxf(x_lo - u, u - x_hi), 0.f);
    const float ddy = fmaxf(fmaxf(y_lo - v, v - y_hi), 0.f);
    return ddx * ddx + ddy * ddy <= cutoff;
}

template <bool CUT>
__global__ void __launch_bounds__(256) bin_count_kernel(const float* __restrict__ uv,
                                                        const int32_t* __restrict__ radius,
                                                        const float* __restrict__ cutoff, int N, int gx, int gy,
                                                        int32_t* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int r = radius[i];
    if (r <= 0) return;
    const float u = uv[2 * i], v = uv[2 * i + 1];
    int x0, x1, y0, y1;
    tile_rect(u, v, r, gx, gy, x0, x1, y0, y1);
    const float cut = CUT ? cutoff[i] : 0.f;
    for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) {
            if (CUT && !tile_hit(u, v, cut, tx, ty)) continue;
            atomicAdd(&counts[ty * gx + tx], 1);
        }
}

// In-place exclusive scan of counts[0..T) -> offsets, offsets[T] = total.
__global__ void __launch_bounds__(1024) bin_scan_kernel(int32_t* __restrict__ data, int T) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 1024) {
        const int i = base + tid;
        const int v = (i < T) ? data[i] : 0;
        // wave inclusive scan
        int s = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(s, off);
            if (lane >= off) s += n;
        }
        if (lane == 63) wsum[wid] = s;
        __syncthreads();
        int wprefix = 0;
        for (int w = 0; w < wid; ++w) wprefix += wsum[w];
        const int carry = carry_s;
        if (i < T) data[i] = carry + wprefix + s - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wprefix + s;
        __syncthreads();
    }
    if (tid == 0) data[T] = carry_s;
}

template <bool CUT>
__global__ void __launch_bounds__(256) bin_scatter_kernel(const float* __restrict__ uv,
                                                          const float* __restrict__ depth,
                                                          const int32_t* __restrict__ radius,
                                                          const float* __restrict__ cutoff, int N, int gx, int gy,
                                                          const int32_t* __restrict__ offsets,
                                                          int32_t* __restrict__ cursor, int K_cap,
                                                          unsigned long long* __restrict__ keys,
                                                          int32_t* __restrict__ overflow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int r = radius[i];
    if (r <= 0) return;
    const float u = uv[2 * i], v = uv[2 * i + 1];
    int x0, x1, y0, y1;
    tile_rect(u, v, r, gx, gy, x0, x1, y0, y1);
    const float cut = CUT ? cutoff[i] : 0.f;
    const unsigned long long key =
        ((unsigned long long)__float_as_uint(depth[i]) << 32) | (unsigned long long)(unsigned)i;
    for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) {
            if (CUT && !tile_hit(u, v, cut, tx, ty)) continue;
            const int t = ty * gx + tx;
            const int pos = offsets[t] + atomicAdd(&cursor[t], 1);
            if (pos < K_cap) keys[pos] = key;
            else *overflow = 1;
        }
}

// All-ascending bitonic network over n keys with virtual +inf padding.
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr a, int n) {
    int npow = 1;
    while (npow < n) npow <<= 1;
    const int half = npow >> 1;
    for (int k = 2; k <= npow; k <<= 1) {
        const int hk = k >> 1;
        for (int idx = threadIdx.x; idx < half; idx += blockDim.x) {
            const int blk = idx / hk, off = idx - blk * hk;
            const int i = blk * k + off, j = blk * k + (k - 1 - off);
            if (j < n) {
                const unsigned long long x = a[i], y = a[j];
                if (x > y) { a[i] = y; a[j] = x; }
            }
        }
        __syncthreads();
        for (int s = hk >> 1; s >= 1; s >>= 1) {
            for (int idx = threadIdx.x; idx < half; idx += blockDim.x) {
                const int blk = idx / s, off = idx - blk * s;
                const int i = blk * 2 * s + off, j = i + s;
                if (j < n) {
                    const unsigned long long x = a[i], y = a[j];
                    if (x > y) { a[i] = y; a[j] = x; }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace gfl

#include "gfl_tile_sort.hpp"

#include <stdlib.h>

namespace gfl {
}  // namespace gfl

using namespace gfl;

extern "C" {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t gfl_bin_workspace_bytes(int N, int K_cap, int W, int H) {
    (void)N;
    if (W <= 0 || H <= 0 || K_cap < 0) return 0;
    const size_t T = (size_t)((W + GFL_TILE - 1) / GFL_TILE) * ((H + GFL_TILE - 1) / GFL_TILE);
    return align_up(T * sizeof(int32_t), 256) + align_up((size_t)K_cap * sizeof(unsigned long long), 256) + 256;
}

int gfl_bin_count(const float* uv, const int32_t* radius, const float* cutoff, int N, int W, int H,
                  int32_t* tile_offsets, gfl_stream_t stream) {
    if (N < 0 || W <= 0 || H <= 0 || !tile_offsets) return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (W + GFL_TILE - 1) / GFL_TILE, gy = (H + GFL_TILE - 1) / GFL_TILE, T = gx * gy;
    int rc = check(hipMemsetAsync(tile_offsets, 0, (size_t)(T + 1) * sizeof(int32_t), s));
    if (rc) return rc;
    if (N > 0) {
        if (!uv || !radius) return GFL_ERR_INVALID;
        if (cutoff) bin_count_kernel<true><<<(N + 255) / 256, 256, 0, s>>>(uv, radius, cutoff, N, gx, gy, tile_offsets);
        else bin_count_kernel<false><<<(N + 255) / 256, 256, 0, s>>>(uv, radius, cutoff, N, gx, gy, tile_offsets);
    }
    bin_scan_kernel<<<1, 1024, 0, s>>>(tile_offsets, T);
    return check_launch();
}

int gfl_bin_sort(const float* uv, const float* depth, const int32_t* radius, const float* cutoff, int N, int W,
                 int H, const int32_t* tile_offsets, int K_cap, int32_t* ids, int32_t* tile_range,
                 int32_t* overflow, void* workspace, size_t workspace_bytes, gfl_stream_t stream) {
    if (N < 0 || W <= 0 || H <= 0 || K_cap < 0 || !tile_offsets || !tile_range || !overflow || !workspace)
        return GFL_ERR_INVALID;
    if (workspace_bytes < gfl_bin_workspace_bytes(N, K_cap, W, H)) return GFL_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (W + GFL_TILE - 1) / GFL_TILE, gy = (H + GFL_TILE - 1) / GFL_TILE, T = gx * gy;
    int32_t* cursor = (int32_t*)workspace;
    unsigned long long* keys =
        (unsigned long long*)((char*)workspace + align_up((size_t)T * sizeof(int32_t), 256));
    int rc = check(hipMemsetAsync(cursor, 0, (size_t)T * sizeof(int32_t), s));
    if (rc) return rc;
    rc = check(hipMemsetAsync(overflow, 0, sizeof(int32_t), s));
    if (rc) return rc;
    if (N > 0) {
        if (!uv || !depth || !radius || (K_cap > 0 && !ids)) return GFL_ERR_INVALID;
        if (cutoff)
            bin_scatter_kernel<true><<<(N + 255) / 256, 256, 0, s>>>(uv, depth, radius, cutoff, N, gx, gy, tile_offsets,
                                                                     cursor, K_cap, keys, overflow);
        else
            bin_scatter_kernel<false><<<(N + 255) / 256, 256, 0, s>>>(uv, depth, radius, cutoff, N, gx, gy, tile_offsets,
                                                                      cursor, K_cap, keys, overflow);
    }
    bin_tile_sort_kernel<<<T, SORT_THREADS, 0, s>>>(tile_offsets, K_cap, keys, ids, tile_range, gx, gy, nullptr, nullptr, nullptr, nullptr);
    return check_launch();
}

int gfl_tile_sort_only(const int32_t* tile_offsets, int T, int K_cap, void* keys, int32_t* ids, int32_t* tile_range,
                       gfl_stream_t stream) {
    if (T <= 0 || K_cap < 0 || !tile_offsets || !keys || !tile_range || (K_cap > 0 && !ids)) return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* k64 = (unsigned long long*)keys;
    bin_tile_sort_kernel<<<T, SORT_THREADS, 0, s>>>(tile_offsets, K_cap, k64, ids, tile_range, 1, T, nullptr, nullptr, nullptr, nullptr);
    return check_launch();
}

int gfl_tile_sort_ordered(const int32_t* order, int W, int H, int K_cap, void* keys, int32_t* ids, int32_t* tile_range,
                          gfl_stream_t stream) {
    if (W <= 0 || H <= 0 || K_cap < 0 || !order || !keys || !tile_range || (K_cap > 0 && !ids)) return GFL_ERR_INVALID;
    const int gx = (W + GFL_TILE - 1) / GFL_TILE, gy = (H + GFL_TILE - 1) / GFL_TILE;
    bin_tile_sort_kernel<<<SORT_MAX_SPLIT + gx * gy, SORT_THREADS, 0, (hipStream_t)stream>>>(
        nullptr, K_cap, (unsigned long long*)keys, ids, tile_range, gx, gy,
        reinterpret_cast<const int4*>(order), nullptr, nullptr, nullptr);
    return check_launch();
}

int gfl_tile_sort_reserved(const int32_t* order, const int32_t* fill, int32_t* tile_counts, int32_t* void_words, int W, int H,
                           int K_cap, void* keys, int32_t* ids, int32_t* tile_range, gfl_stream_t stream) {
    if (W <= 0 || H <= 0 || K_cap < 0 || !order || !fill || !tile_counts || !keys || !tile_range || (K_cap > 0 && !ids))
        return GFL_ERR_INVALID;
    const int gx = (W + GFL_TILE - 1) / GFL_TILE, gy = (H + GFL_TILE - 1) / GFL_TILE;
    bin_tile_sort_kernel<<<SORT_MAX_SPLIT + gx * gy, SORT_THREADS, 0, (hipStream_t)stream>>>(
        nullptr, K_cap, (unsigned long long*)keys, ids, tile_range, gx, gy,
        reinterpret_cast<const int4*>(order), fill, tile_counts, void_words);
    return check_launch();
}

#ifdef GFL_TRACE
int gfl_debug_read_sort_trace(long long* out, int n_tiles) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gfl::g_sort_trace), (size_t)n_tiles * 4 * sizeof(long long));
}
#endif

}  // extern "C"
