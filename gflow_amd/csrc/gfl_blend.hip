// Front-to-back alpha compositing and its per-pixel backward
// (msplat.alpha_blending, render.py:58-105).
//
// One workgroup (4 wave64) per 16x16 tile.  A WAVE OWNS AN 8x8 PIXEL BLOCK (lane ->
// (lane&7, lane>>3)), so the wave-level tests below act on a compact footprint.
// The tile's depth-sorted splat list is staged through LDS 256 records at a time
// (three wide LDS reads per splat, all lanes the same address = broadcast).
#include "gfl_common.hpp"

namespace gfl {

constexpr int BLEND_BATCH = 256;

struct SplatRec {            // 40 bytes + pad: what a pixel needs from one splat
    float4 p0;               // u, v, conic a, conic b
    float4 p1;               // conic c, opacity, f0, f1
    float2 p2;               // f2, f3
};

// alpha of one splat at one pixel; identical instruction sequence in the forward
// and the backward (explicit fma, contraction off) so both take the same
// skip/keep decision.  Returns false when the splat is skipped.
__device__ __forceinline__ bool splat_alpha(float u, float v, float A, float B, float C, float o, float fx, float fy,
                                            float& alpha, float& G) {
#pragma clang fp contract(off)
    const float dx = u - fx, dy = v - fy;
    const float q = __builtin_fmaf(A * dx, dx, (C * dy) * dy);
    const float power = __builtin_fmaf(-0.5f, q, -((B * dx) * dy));
    if (power > 0.f) return false;
    G = __expf(power);
    alpha = fminf(GFL_ALPHA_MAX, o * G);
    return alpha >= GFL_ALPHA_MIN;
}

template <int C>
__device__ __forceinline__ void stage_splat(SplatRec* __restrict__ rec, int g, const float* __restrict__ uv,
                                            const float* __restrict__ conic, const float* __restrict__ opacity,
                                            const float* __restrict__ feature, int C_total, int c0) {
    const float2 p = reinterpret_cast<const float2*>(uv)[g];
    const float a = conic[3 * g], b = conic[3 * g + 1], c = conic[3 * g + 2];
    float f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < C; ++k) f[k] = feature[(size_t)g * C_total + c0 + k];
    rec->p0 = make_float4(p.x, p.y, a, b);
    rec->p1 = make_float4(c, opacity[g], f[0], f[1]);
    rec->p2 = make_float2(f[2], f[3]);
}

template <int C>
__global__ void __launch_bounds__(256) blend_fwd_kernel(const float* __restrict__ uv, const float* __restrict__ conic,
                                                        const float* __restrict__ opacity,
                                                        const float* __restrict__ feature, int C_total, int c0,
                                                        const int32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ tile_range, float bg, int W, int H,
                                                        int gx, float* __restrict__ out, float* __restrict__ final_T,
                                                        int32_t* __restrict__ n_contrib) {
    __shared__ SplatRec recs[BLEND_BATCH];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * GFL_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GFL_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = pixf(px), fy = pixf(py);
    const int start = tile_range[2 * tile], end = tile_range[2 * tile + 1];

    float T = 1.f;
    float acc[C];
#pragma unroll
    for (int k = 0; k < C; ++k) acc[k] = 0.f;
    int last = 0;
    bool done = !inside;

    for (int base = start; base < end; base += BLEND_BATCH) {
        if (__syncthreads_and(done)) break;
        const int idx = base + tid;
        if (idx < end) stage_splat<C>(&recs[tid], ids[idx], uv, conic, opacity, feature, C_total, c0);
        __syncthreads();
        const int cnt = min(BLEND_BATCH, end - base);
        if (!done) {
            for (int j = 0; j < cnt; ++j) {
                const float4 p0 = recs[j].p0;
                const float4 p1 = recs[j].p1;
                float alpha, G;
                if (!splat_alpha(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, fx, fy, alpha, G)) continue;
                const float test_T = T * (1.f - alpha);
                if (test_T < GFL_T_MIN) { done = true; break; }
                const float w = alpha * T;
                const float2 p2 = recs[j].p2;
                const float f[4] = {p1.z, p1.w, p2.x, p2.y};
#pragma unroll
                for (int k = 0; k < C; ++k) acc[k] = fmaf(f[k], w, acc[k]);
                T = test_T;
                last = base - start + j + 1;
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px;
#pragma unroll
        for (int k = 0; k < C; ++k) out[(size_t)k * H * W + pix] = fmaf(T, bg, acc[k]);
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// Backward: every pixel walks its contributing splats back to front.
//   T_i = T_{i+1}/(1-a_i);  h = <g, f_i>;  dL/da_i = T_i h - S/(1-a_i);  S += h a_i T_i
// with S initialised to T_final * bg * sum(g).  Per-splat gradients are reduced over
// the wave's 64 pixels on DPP and lane 63 issues one float atomic per value.
template <int C>
__global__ void __launch_bounds__(256) blend_bwd_kernel(const float* __restrict__ uv, const float* __restrict__ conic,
                                                        const float* __restrict__ opacity,
                                                        const float* __restrict__ feature, int C_total, int c0,
                                                        const int32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ tile_range, float bg, int W, int H,
                                                        int gx, const float* __restrict__ final_T,
                                                        const int32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ d_out, float* __restrict__ d_uv,
                                                        float* __restrict__ d_conic, float* __restrict__ d_opacity,
                                                        float* __restrict__ d_feature) {
    __shared__ SplatRec recs[BLEND_BATCH];
    __shared__ int32_t rec_id[BLEND_BATCH];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * GFL_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GFL_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = pixf(px), fy = pixf(py);
    const int start = tile_range[2 * tile], end = tile_range[2 * tile + 1];
    const int total = end - start;

    float g[C];
    float T = 1.f, S = 0.f;
    int last = 0;
    if (inside) {
        const size_t pix = (size_t)py * W + px;
        T = final_T[pix];
        last = n_contrib[pix];
        float gs = 0.f;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            g[k] = d_out[(size_t)k * H * W + pix];
            gs += g[k];
        }
        S = T * bg * gs;
    } else {
#pragma unroll
        for (int k = 0; k < C; ++k) g[k] = 0.f;
    }
    // the tile only needs splats up to the deepest contributor of any pixel
    __shared__ int32_t s_max_last;
    if (tid == 0) s_max_last = 0;
    __syncthreads();
    {
        int m = last;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off));
        if (lane == 0) atomicMax(&s_max_last, m);
    }
    __syncthreads();
    const int depth_n = min(total, (int)s_max_last);

    for (int r0 = 0; r0 < depth_n; r0 += BLEND_BATCH) {
        // stage list positions depth_n-1-r0 ... downwards; slot j <-> position depth_n-1-r0-j
        const int pos_t = depth_n - 1 - r0 - tid;
        __syncthreads();
        if (pos_t >= 0) {
            const int gid = ids[start + pos_t];
            rec_id[tid] = gid;
            stage_splat<C>(&recs[tid], gid, uv, conic, opacity, feature, C_total, c0);
        }
        __syncthreads();
        const int cnt = min(BLEND_BATCH, depth_n - r0);
        for (int j = 0; j < cnt; ++j) {
            const int pos = depth_n - 1 - r0 - j;  // 0-based list position
            const float4 p0 = recs[j].p0;
            const float4 p1 = recs[j].p1;
            float alpha = 0.f, G = 0.f;
            bool valid = (pos < last) && splat_alpha(p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, fx, fy, alpha, G);
            if (__ballot(valid) == 0ull) continue;
            float v_u = 0.f, v_v = 0.f, v_a = 0.f, v_b = 0.f, v_c = 0.f, v_o = 0.f;
            float v_f[C];
#pragma unroll
            for (int k = 0; k < C; ++k) v_f[k] = 0.f;
            if (valid) {
                const float2 p2 = recs[j].p2;
                const float f[4] = {p1.z, p1.w, p2.x, p2.y};
                const float om = 1.f - alpha;
                T = T / om;
                float h = 0.f;
#pragma unroll
                for (int k = 0; k < C; ++k) h = fmaf(g[k], f[k], h);
                const float dalpha = T * h - S / om;
                const float w = alpha * T;
                S = fmaf(h, w, S);
#pragma unroll
                for (int k = 0; k < C; ++k) v_f[k] = w * g[k];
                const float dx = p0.x - fx, dy = p0.y - fy;
                v_o = G * dalpha;
                const float dpow = p1.y * G * dalpha;
                v_a = -0.5f * dx * dx * dpow;
                v_c = -0.5f * dy * dy * dpow;
                v_b = -dx * dy * dpow;
                v_u = -(p0.z * dx + p0.w * dy) * dpow;
                v_v = -(p1.x * dy + p0.w * dx) * dpow;
            }
            v_u = wave_sum_to_lane63(v_u); v_v = wave_sum_to_lane63(v_v);
            v_a = wave_sum_to_lane63(v_a); v_b = wave_sum_to_lane63(v_b); v_c = wave_sum_to_lane63(v_c);
            v_o = wave_sum_to_lane63(v_o);
#pragma unroll
            for (int k = 0; k < C; ++k) v_f[k] = wave_sum_to_lane63(v_f[k]);
            if (lane == 63) {
                const int gid = rec_id[j];
                atomicAdd(&d_uv[2 * gid], v_u); atomicAdd(&d_uv[2 * gid + 1], v_v);
                atomicAdd(&d_conic[3 * gid], v_a); atomicAdd(&d_conic[3 * gid + 1], v_b);
                atomicAdd(&d_conic[3 * gid + 2], v_c);
                atomicAdd(&d_opacity[gid], v_o);
#pragma unroll
                for (int k = 0; k < C; ++k) atomicAdd(&d_feature[(size_t)gid * C_total + c0 + k], v_f[k]);
            }
        }
    }
}

template <int C>
static int launch_fwd(const float* uv, const float* conic, const float* opacity, const float* feature, int C_total,
                      int c0, const int32_t* ids, const int32_t* tile_range, float bg, int W, int H, float* out,
                      float* final_T, int32_t* n_contrib, hipStream_t s) {
    const int gx = (W + GFL_TILE - 1) / GFL_TILE, gy = (H + GFL_TILE - 1) / GFL_TILE;
    blend_fwd_kernel<C><<<gx * gy, 256, 0, s>>>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, gx,
                                                out, final_T, n_contrib);
    return check_launch();
}

template <int C>
static int launch_bwd(const float* uv, const float* conic, const float* opacity, const float* feature, int C_total,
                      int c0, const int32_t* ids, const int32_t* tile_range, float bg, int W, int H,
                      const float* final_T, const int32_t* n_contrib, const float* d_out, float* d_uv, float* d_conic,
                      float* d_opacity, float* d_feature, hipStream_t s) {
    const int gx = (W + GFL_TILE - 1) / GFL_TILE, gy = (H + GFL_TILE - 1) / GFL_TILE;
    blend_bwd_kernel<C><<<gx * gy, 256, 0, s>>>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, gx,
                                                final_T, n_contrib, d_out, d_uv, d_conic, d_opacity, d_feature);
    return check_launch();
}

}  // namespace gfl

using namespace gfl;

extern "C" {

int gfl_blend_fwd(const float* uv, const float* conic, const float* opacity, const float* feature, int C_total,
                  int c0, int C, const int32_t* ids, const int32_t* tile_range, float bg, int W, int H, float* out,
                  float* final_T, int32_t* n_contrib, gfl_stream_t stream) {
    if (W <= 0 || H <= 0 || C < 1 || C > GFL_MAX_BLEND_CHANNELS || c0 < 0 || c0 + C > C_total) return GFL_ERR_INVALID;
    if (!tile_range || !out || !final_T || !n_contrib) return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 1: return launch_fwd<1>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, out, final_T, n_contrib, s);
        case 2: return launch_fwd<2>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, out, final_T, n_contrib, s);
        case 3: return launch_fwd<3>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, out, final_T, n_contrib, s);
        default: return launch_fwd<4>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, out, final_T, n_contrib, s);
    }
}

int gfl_blend_bwd(const float* uv, const float* conic, const float* opacity, const float* feature, int C_total,
                  int c0, int C, const int32_t* ids, const int32_t* tile_range, float bg, int W, int H,
                  const float* final_T, const int32_t* n_contrib, const float* d_out, int N, float* d_uv,
                  float* d_conic, float* d_opacity, float* d_feature, int zero_first, gfl_stream_t stream) {
    if (W <= 0 || H <= 0 || N < 0 || C < 1 || C > GFL_MAX_BLEND_CHANNELS || c0 < 0 || c0 + C > C_total)
        return GFL_ERR_INVALID;
    if (!tile_range || !final_T || !n_contrib || !d_out) return GFL_ERR_INVALID;
    if (N == 0) return GFL_OK;
    if (!d_uv || !d_conic || !d_opacity || !d_feature) return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (zero_first) {
        int rc = check(hipMemsetAsync(d_uv, 0, (size_t)N * 2 * sizeof(float), s));
        if (!rc) rc = check(hipMemsetAsync(d_conic, 0, (size_t)N * 3 * sizeof(float), s));
        if (!rc) rc = check(hipMemsetAsync(d_opacity, 0, (size_t)N * sizeof(float), s));
        if (!rc) rc = check(hipMemsetAsync(d_feature, 0, (size_t)N * C_total * sizeof(float), s));
        if (rc) return rc;
    }
    switch (C) {
        case 1: return launch_bwd<1>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, final_T, n_contrib, d_out, d_uv, d_conic, d_opacity, d_feature, s);
        case 2: return launch_bwd<2>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, final_T, n_contrib, d_out, d_uv, d_conic, d_opacity, d_feature, s);
        case 3: return launch_bwd<3>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, final_T, n_contrib, d_out, d_uv, d_conic, d_opacity, d_feature, s);
        default: return launch_bwd<4>(uv, conic, opacity, feature, C_total, c0, ids, tile_range, bg, W, H, final_T, n_contrib, d_out, d_uv, d_conic, d_opacity, d_feature, s);
    }
}

}  // extern "C"
