// Photometric (per-pixel MSE) + SSIM + scale/shift depth loss, forward AND backward, in two
// stencil kernels (trainer.py:452-488, utils/pytorch_ssim.py:17-37).
//
// SSIM: the reference convolves x, y, x^2, y^2, xy with an 11x11 Gaussian window (zero padding)
// and differentiates through five depthwise conv2d calls.  Here
//   pass 1 (ssim_stats): 16x16 output tile + 5-pixel halo of ALL THREE channels staged in LDS
//          (the HWC ground truth is read as whole pixels), separable 11-tap filter of the five
//          maps, SSIM value and its three partial derivatives (wrt mu1, E[x^2], E[xy]) written out;
//   pass 2 (loss_grad) : the same separable filter over the nine derivative maps (the window is
//          symmetric, so the adjoint of the conv is the conv):
//          dL/dx = conv(dmu1) + 2x conv(de11) + y conv(de12), fused with the MSE and depth-term
//          gradients into d_render[4,H,W]; the fourth wave does the depth plane and the error map.
// LDS traffic is what bounds a stencil like this, so both 1-D passes use a REGISTER SLIDING
// WINDOW: a lane produces four adjacent outputs from fourteen inputs fetched once (16-byte LDS
// reads in the row pass) instead of 4 x 11 single reads -- ~4x fewer LDS instructions than the
// first version of these kernels.  Scalar sums go block-partial -> ordered fold (reproducible).
#include "gfl_common.hpp"

#include <math.h>

namespace gfl {

constexpr int SW = 11;           // SSIM window
constexpr int SR = SW / 2;       // halo
constexpr int ST = 16;           // output tile edge
constexpr int SI = ST + 2 * SR;  // staged tile edge (26)
constexpr int SP = 48;           // staged row pitch (floats).  The row pass reads 16-byte chunks, four lanes per
                                 // row: with a pitch of 48 floats four consecutive rows fall on disjoint banks
                                 // (28 gave a 2-way conflict on every read: 60 % of the LDS cycles, rocprofv3 PMC)
constexpr int HP = 24;           // row-filtered pitch (floats): the column pass's four row pairs on disjoint banks
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

struct Win { float w[SW]; };

__device__ __forceinline__ void load16(const float* __restrict__ row, float (&v)[16]) {
    const float4* p = reinterpret_cast<const float4*>(row);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
}

// 12 floats starting at a 16-byte aligned LDS address
__device__ __forceinline__ void load12(const float* __restrict__ row, float (&v)[12]) {
    const float4* p = reinterpret_cast<const float4*>(row);
    const float4 a = p[0], b = p[1], c = p[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}

// grid gx*gy*3 blocks in XCD order, the three channel blocks of a tile adjacent (they share the
// HWC ground-truth lines); dmaps[3 channels][3 maps][H][W]; partial[block] = sum of S
typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: all five maps from scratch.
// MODE 1: mu2 = conv(y) and E[y^2] = conv(y^2) come from gt_stats[3 channels][2][H][W], computed once
//         per ground-truth image by MODE 2 with the same arithmetic (the target does not change
//         during the hundreds of iterations of a fit): three maps instead of five through LDS and
//         the filter passes.
// MODE 2: write gt_stats, nothing else.
// KEEP: an occlusion mask is given (a second template parameter instead of a pointer test per load: the staging loads
// below have to be straight-line code to be issued together).
template <int MODE, bool KEEP>
__global__ void __launch_bounds__(256) ssim_stats_kernel(const float* __restrict__ render,
                                                         const float* __restrict__ gt_rgb,
                                                         const uint8_t* __restrict__ keep, int W, int H, int gx,
                                                         int gy, Win win, float scale /* dL/dS per element */,
                                                         float* __restrict__ dmaps, float* __restrict__ partial,
                                                         float* __restrict__ gt_stats) {
    // The kernel was bound by VALU issue (62 % busy, rocprofv3 PMC), so the maps are staged as
    // packed pairs with the products formed ONCE per staged pixel, and both filter passes run
    // v_pk_fma_f32 on the pairs.  pair = (x, y) and (x^2, y^2) in MODE 0, (x, x^2) in MODE 1,
    // (y, y^2) in MODE 2; the fifth map xy is a plain float (MODE 0 and 1).
    __shared__ v2f s_a[SI][SI + 1];
    __shared__ v2f s_b[MODE == 0 ? SI : 1][SI + 1];
    __shared__ float s_x_y[MODE == 2 ? 1 : SI][SI + 1];
    __shared__ v2f h_a[SI][ST + 1];
    __shared__ v2f h_b[MODE == 0 ? SI : 1][ST + 1];
    __shared__ float h_xy[MODE == 2 ? 1 : SI][ST + 1];
    const int lb = xcd_logical_block(blockIdx.x, gx * gy * 3);
    const int c = lb % 3, tile = lb / 3;
    const int bx = tile % gx, by = tile / gx;
    const int x0 = bx * ST - SR, y0 = by * ST - SR;
    const int tid = threadIdx.x;
    // MODE 1: the cached statistics of this lane's own pixel, requested before anything else (read where they are
    // used, after the column pass, the two loads were a full memory round trip at the end of the workgroup's life)
    float mu2_own = 0.f, e22_own = 0.f;
    if (MODE == 1) {
        const int opx = bx * ST + (tid & 15), opy = by * ST + (tid >> 4);
        if (opx < W && opy < H) {
            const size_t opix = (size_t)opy * W + opx;
            mu2_own = gt_stats[(size_t)(2 * c) * H * W + opix];
            e22_own = gt_stats[(size_t)(2 * c + 1) * H * W + opix];
        }
    }
    // Staging: ALL of a lane's loads (three staged pixels x render, target, mask) are issued before the first is used.
    // Written as a loop with a load, a test and a store per pass, the compiler waited for memory in every pass: six
    // dependent round trips in the life of a workgroup, which is what these kernels' duration was made of.  Addresses
    // outside the image are clamped to pixel 0 and the value replaced afterwards.
    constexpr int NS = (SI * SI + 255) / 256;
    const size_t plane_s = (size_t)H * W;
    float xs[NS], ys[NS];
    int kb[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int i = min(tid + 256 * j, SI * SI - 1);
        const int r = i / SI, q = i - r * SI;
        const int x = x0 + q, y = y0 + r;
        const bool in = x >= 0 && y >= 0 && x < W && y < H;
        const size_t pix = in ? (size_t)y * W + x : 0;
        xs[j] = MODE == 2 ? 0.f : render[(size_t)c * plane_s + pix];
        ys[j] = gt_rgb[pix * 3 + c];
        kb[j] = KEEP ? (int)keep[pix] : 1;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        // (no branch here either, or the last pass's loads sink into it: lanes past the end write the pad column)
        const int i = tid + 256 * j;
        int r = i / SI, q = i - r * SI;
        const int x = x0 + q, y = y0 + r;
        const bool k = x >= 0 && y >= 0 && x < W && y < H && kb[j] != 0;
        const float xv = k ? xs[j] : 0.f, yv = k ? ys[j] : 0.f;
        if (i >= SI * SI) { r = tid & 15; q = SI; }
        if (MODE == 0) {
            s_a[r][q] = (v2f){xv, yv};
            s_b[r][q] = (v2f){xv * xv, yv * yv};
        } else if (MODE == 1) {
            s_a[r][q] = (v2f){xv, xv * xv};
        } else {
            s_a[r][q] = (v2f){yv, yv * yv};
        }
        if (MODE != 2) s_x_y[r][q] = xv * yv;
    }
    __syncthreads();
    for (int i = tid; i < SI * ST; i += 256) {
        const int r = i / ST, q = i - r * ST;
        v2f a_a = {0.f, 0.f}, a_b = {0.f, 0.f};
        float a_xy = 0.f;
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const float w = win.w[k];
            const v2f w2 = {w, w};
            a_a = __builtin_elementwise_fma(w2, s_a[r][q + k], a_a);
            if (MODE == 0) a_b = __builtin_elementwise_fma(w2, s_b[r][q + k], a_b);
            if (MODE != 2) a_xy = fmaf(w, s_x_y[r][q + k], a_xy);
        }
        h_a[r][q] = a_a;
        if (MODE == 0) h_b[r][q] = a_b;
        if (MODE != 2) h_xy[r][q] = a_xy;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int px = bx * ST + lx, py = by * ST + ly;
    float sval = 0.f;
    if (px < W && py < H) {
        v2f fa = {0.f, 0.f}, fb = {0.f, 0.f};
        float e12 = 0.f;
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const float w = win.w[k];
            const v2f w2 = {w, w};
            fa = __builtin_elementwise_fma(w2, h_a[ly + k][lx], fa);
            if (MODE == 0) fb = __builtin_elementwise_fma(w2, h_b[ly + k][lx], fb);
            if (MODE != 2) e12 = fmaf(w, h_xy[ly + k][lx], e12);
        }
        const size_t plane = (size_t)H * W, pix = (size_t)py * W + px;
        if (MODE == 2) {
            gt_stats[(size_t)(2 * c) * plane + pix] = fa.x;          // mu2
            gt_stats[(size_t)(2 * c + 1) * plane + pix] = fa.y;      // E[y^2]
            return;
        }
        float mu1, mu2, e11, e22;
        if (MODE == 0) {
            mu1 = fa.x; mu2 = fa.y; e11 = fb.x; e22 = fb.y;
        } else {
            mu1 = fa.x; e11 = fa.y;
            mu2 = mu2_own;
            e22 = e22_own;
        }
        const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
        const float A1 = 2.f * mu12 + SSIM_C1, A2 = 2.f * s12 + SSIM_C2;
        const float B1 = mu1s + mu2s + SSIM_C1, B2 = s1 + s2 + SSIM_C2;
        const float inv = 1.f / (B1 * B2);
        sval = A1 * A2 * inv;
        const float d_e11 = -sval / B2;
        const float d_e12 = 2.f * A1 * inv;
        const float d_mu1 = 2.f * mu2 * (A2 - A1) * inv - 2.f * mu1 * sval * (1.f / B1 - 1.f / B2);
        float* base = dmaps + (size_t)c * 3 * plane;
        base[pix] = scale * d_mu1;
        base[plane + pix] = scale * d_e11;
        base[2 * plane + pix] = scale * d_e12;
    }
    if (MODE == 2) return;
    float v[1] = {sval};
    const int bid = c * gx * gy + tile;
    // block_reduce_store indexes by blockIdx.x only -> reduce by hand here
    __shared__ float red[4];
    const float s = wave_sum_to_lane63(v[0]);
    if ((tid & 63) == 63) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) partial[bid] = red[0] + red[1] + red[2] + red[3];
}

// grid gx*gy*4 blocks in XCD order, the four blocks of a tile adjacent: 0..2 -> gradient of one
// rgb plane; 3 -> depth plane + per-pixel mse.
// partial rows: [block][4] = {sum mse_px, sum depth term, d/d depth_a, d/d depth_b}
template <bool KEEP>
__global__ void __launch_bounds__(256) loss_grad_kernel(
    const float* __restrict__ render, const float* __restrict__ gt_rgb, const float* __restrict__ gt_depth,
    const uint8_t* __restrict__ keep, const float* __restrict__ depth_ab, const float* __restrict__ dmaps, int W, int H,
    int gx, int gy, Win win, float mse_scale /* lambda_rgb * 2/(3HW) */, float depth_scale /* lambda_depth/(HW) */,
    float* __restrict__ d_render, float* __restrict__ err_px, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float sm[3][SI][SP];
    __shared__ __attribute__((aligned(16))) float hz[3][SI][HP];
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W;
    const int lb = xcd_logical_block(blockIdx.x, gx * gy * 4);
    const int ch = lb & 3, tile = lb >> 2;
    const int bx = tile % gx, by = tile / gx;
    if (ch == 3) {
        const int lx = tid & 15, ly = tid >> 4;
        const int px = bx * ST + lx, py = by * ST + ly;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (px < W && py < H) {
            // every load first (see the staging below)
            const size_t pix = (size_t)py * W + px;
            const int kb = KEEP ? (int)keep[pix] : 1;
            float rr[3], gg[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rr[c] = render[c * plane + pix];
                gg[c] = gt_rgb[pix * 3 + c];
            }
            const bool with_depth = depth_scale != 0.f;
            const float D = with_depth ? render[3 * plane + pix] : 0.f;
            const float gt = with_depth ? gt_depth[pix] : 0.f;
            const bool k = kb != 0;
            float e = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = k ? (rr[c] - gg[c]) : 0.f;
                e = fmaf(d, d, e);
            }
            e *= (1.f / 3.f);
            err_px[pix] = e;
            v[0] = e;
            float gD = 0.f;
            if (with_depth) {
                const float a = depth_ab[0], b = depth_ab[1];
                const float d = fmaf(a, D, b);
                const float diff = d - gt, sum = d + gt;
                if (k) {
                    v[1] = diff * diff / sum;
                    const float dl = depth_scale * diff * (d + 3.f * gt) / (sum * sum);
                    gD = dl * a;
                    v[2] = dl * D;
                    v[3] = dl;
                }
            }
            d_render[3 * plane + pix] = gD;
        }
        __shared__ float red[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float s = wave_sum_to_lane63(v[q]);
            if ((tid & 63) == 63) red[tid >> 6][q] = s;
        }
        __syncthreads();
        if (tid < 4) partial[(size_t)tile * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        return;
    }
    const int x0 = bx * ST - SR, y0 = by * ST - SR;
    const float* base = dmaps + (size_t)ch * 3 * plane;
    // the column-pass lanes fetch their own pixels' x, y now: issued last, the two loads would sit
    // at the end of the workgroup's life with nothing left to overlap them
    float own_x[2] = {0.f, 0.f}, own_y[2] = {0.f, 0.f};
    int own_kb[2] = {0, 0};
    if (tid < 128) {
        const int r0 = (tid >> 4) * 2, col = tid & 15;
        const int px = bx * ST + col;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int py = by * ST + r0 + o;
            if (px < W && py < H) {
                const size_t pix = (size_t)py * W + px;
                own_kb[o] = KEEP ? (int)keep[pix] : 1;
                own_x[o] = render[ch * plane + pix];
                own_y[o] = gt_rgb[pix * 3 + ch];
            }
        }
    }
    // staging: 26 rows x 28 columns (columns 26, 27 are zero padding for the 16-byte row reads) of the three maps, all
    // nine loads of a lane in flight together (ssim_stats_kernel explains why)
    constexpr int SQ = 28, NG = (SI * SQ + 255) / 256;
    float dm[NG][3];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int i = min(tid + 256 * j, SI * SQ - 1);
        const int r = i / SQ, q = i - r * SQ;
        const int x = x0 + q, y = y0 + r;
        const bool in = q < SI && x >= 0 && y >= 0 && x < W && y < H;
        const size_t p = in ? (size_t)y * W + x : 0;
        dm[j][0] = base[p];
        dm[j][1] = base[plane + p];
        dm[j][2] = base[2 * plane + p];
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int i = tid + 256 * j;
        int r = i / SQ, q = i - r * SQ;
        const int x = x0 + q, y = y0 + r;
        const bool in = q < SI && x >= 0 && y >= 0 && x < W && y < H;
        if (i >= SI * SQ) { r = tid & 15; q = SQ + 4; }      // lanes past the end: a pad column nobody reads
        sm[0][r][q] = in ? dm[j][0] : 0.f;
        sm[1][r][q] = in ? dm[j][1] : 0.f;
        sm[2][r][q] = in ? dm[j][2] : 0.f;
    }
    __syncthreads();
    // row pass: item = (map, row, group of 4 columns): 312 items
    for (int it = tid; it < 3 * SI * 4; it += 256) {
        const int m = it / (SI * 4), rem = it - m * (SI * 4);
        const int r = rem >> 2, q = (rem & 3) * 4;
        float X[16];
        load16(&sm[m][r][q], X);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const float w = win.w[k];
#pragma unroll
            for (int o = 0; o < 4; ++o) a[o] = fmaf(w, X[o + k], a[o]);
        }
        *reinterpret_cast<float4*>(&hz[m][r][q]) = make_float4(a[0], a[1], a[2], a[3]);
    }
    __syncthreads();
    // column pass: item = (pair of rows, column): 128 lanes
    if (tid < 128) {
        const int r0 = (tid >> 4) * 2, col = tid & 15;
        const int px = bx * ST + col;
        float g[3][2];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float v[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = hz[m][r0 + k][col];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < SW; ++k) acc = fmaf(win.w[k], v[o + k], acc);
                g[m][o] = acc;
            }
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int py = by * ST + r0 + o;
            if (px < W && py < H) {
                const size_t pix = (size_t)py * W + px;
                float out = 0.f;
                if (own_kb[o] != 0) {
                    const float x = own_x[o], y = own_y[o];
                    out = g[0][o] + 2.f * x * g[1][o] + y * g[2][o] + mse_scale * (x - y);
                }
                d_render[ch * plane + pix] = out;
            }
        }
    }
}

// sums[8] from the two partial arrays (single block, fixed-shape tree)
__global__ void __launch_bounds__(1024) loss_fold_kernel(const float* __restrict__ p_ssim, int n_ssim,
                                                         const float* __restrict__ p_grad, int n_grad,
                                                         float* __restrict__ sums) {
    // one block of 1024 lanes, at most a couple of loads deep per lane (a 256-lane version spent
    // 7 us waiting on a chain of ~25 dependent L2 round trips); fixed-shape tree -> reproducible
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = threadIdx.x; r < n_ssim; r += 1024) acc[1] += p_ssim[r];
    for (int r = threadIdx.x; r < n_grad; r += 1024) {
        const float4 q = reinterpret_cast<const float4*>(p_grad)[r];
        acc[0] += q.x; acc[2] += q.y; acc[3] += q.z; acc[4] += q.w;
    }
    __shared__ float red[16][5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float s = wave_sum_to_lane63(acc[q]);
        if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6][q] = s;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        float t = 0.f;
        if (threadIdx.x < 5)
            for (int w = 0; w < 16; ++w) t += red[w][threadIdx.x];
        sums[threadIdx.x] = t;
    }
}

static Win make_window() {
    // utils/pytorch_ssim.py:7-9: python-float exp cast to float32, float32 sum, divide
    Win w;
    float s = 0.f;
    for (int i = 0; i < SW; ++i) {
        const double d = (double)(i - SR);
        w.w[i] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
        s += w.w[i];
    }
    for (int i = 0; i < SW; ++i) w.w[i] /= s;
    return w;
}

}  // namespace gfl

using namespace gfl;

extern "C" {

static inline size_t align_up256(size_t v) { return (v + 255) / 256 * 256; }

size_t gfl_loss_workspace_bytes(int W, int H) {
    if (W <= 0 || H <= 0) return 0;
    const size_t gx = (W + ST - 1) / ST, gy = (H + ST - 1) / ST;
    return align_up256((size_t)9 * W * H * sizeof(float)) + align_up256(gx * gy * 3 * sizeof(float)) +
           align_up256(gx * gy * 4 * sizeof(float));
}

static int loss_launch(const float* render, const float* gt_rgb, const float* gt_depth, const uint8_t* keep,
                       const float* depth_ab, float lambda_rgb, float lambda_depth, int W, int H, float* d_render,
                       float* err_px, float* sums, void* workspace, size_t workspace_bytes, gfl_stream_t stream,
                       const float** p_ssim_out, int* n_ssim, const float** p_grad_out, int* n_grad,
                       const float* gt_stats = nullptr) {
    if (W <= 0 || H <= 0 || !render || !gt_rgb || !d_render || !err_px || !workspace) return GFL_ERR_INVALID;
    if (lambda_depth != 0.f && (!gt_depth || !depth_ab)) return GFL_ERR_INVALID;
    if (workspace_bytes < gfl_loss_workspace_bytes(W, H)) return GFL_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (W + ST - 1) / ST, gy = (H + ST - 1) / ST;
    float* dmaps = (float*)workspace;
    float* p_ssim = (float*)((char*)workspace + align_up256((size_t)9 * W * H * sizeof(float)));
    float* p_grad = (float*)((char*)p_ssim + align_up256((size_t)gx * gy * 3 * sizeof(float)));
    static const Win win = make_window();
    const float hw = (float)W * (float)H;
    // L = lambda_rgb * (mean mse + 1 - mean S)  ->  dL/dS = -lambda_rgb / (3HW)
    const float s_scale = -lambda_rgb / (3.f * hw);
    float* st = const_cast<float*>(gt_stats);
    auto stats = gt_stats ? (keep ? ssim_stats_kernel<1, true> : ssim_stats_kernel<1, false>)
                          : (keep ? ssim_stats_kernel<0, true> : ssim_stats_kernel<0, false>);
    stats<<<gx * gy * 3, 256, 0, s>>>(render, gt_rgb, keep, W, H, gx, gy, win, s_scale, dmaps, p_ssim, st);
    auto grad = keep ? loss_grad_kernel<true> : loss_grad_kernel<false>;
    grad<<<gx * gy * 4, 256, 0, s>>>(render, gt_rgb, gt_depth, keep, depth_ab, dmaps, W, H, gx, gy, win,
                                     lambda_rgb * 2.f / (3.f * hw), lambda_depth / hw, d_render, err_px, p_grad);
    if (sums) loss_fold_kernel<<<1, 1024, 0, s>>>(p_ssim, gx * gy * 3, p_grad, gx * gy, sums);
    if (p_ssim_out) { *p_ssim_out = p_ssim; *n_ssim = gx * gy * 3; *p_grad_out = p_grad; *n_grad = gx * gy; }
    return check_launch();
}

int gfl_loss_fwd_bwd(const float* render, const float* gt_rgb, const float* gt_depth, const uint8_t* keep,
                     const float* depth_ab, float lambda_rgb, float lambda_depth, int W, int H, float* d_render,
                     float* err_px, float* sums, void* workspace, size_t workspace_bytes, gfl_stream_t stream) {
    if (!sums) return GFL_ERR_INVALID;
    return loss_launch(render, gt_rgb, gt_depth, keep, depth_ab, lambda_rgb, lambda_depth, W, H, d_render, err_px, sums,
                       workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, nullptr);
}

int gfl_loss_fwd_bwd_partials(const float* render, const float* gt_rgb, const float* gt_depth, const uint8_t* keep,
                              const float* depth_ab, float lambda_rgb, float lambda_depth, int W, int H,
                              float* d_render, float* err_px, void* workspace, size_t workspace_bytes,
                              const float** p_ssim, int* n_ssim, const float** p_grad, int* n_grad,
                              gfl_stream_t stream) {
    if (!p_ssim || !n_ssim || !p_grad || !n_grad) return GFL_ERR_INVALID;
    return loss_launch(render, gt_rgb, gt_depth, keep, depth_ab, lambda_rgb, lambda_depth, W, H, d_render, err_px,
                       nullptr, workspace, workspace_bytes, stream, p_ssim, n_ssim, p_grad, n_grad);
}

int gfl_loss_prepare_gt(const float* gt_rgb, const uint8_t* keep, int W, int H, float* gt_stats, gfl_stream_t stream) {
    if (W <= 0 || H <= 0 || !gt_rgb || !gt_stats) return GFL_ERR_INVALID;
    const int gx = (W + ST - 1) / ST, gy = (H + ST - 1) / ST;
    static const Win win = make_window();
    auto stats = keep ? ssim_stats_kernel<2, true> : ssim_stats_kernel<2, false>;
    stats<<<gx * gy * 3, 256, 0, (hipStream_t)stream>>>(nullptr, gt_rgb, keep, W, H, gx, gy, win, 0.f, nullptr, nullptr,
                                                        gt_stats);
    return check_launch();
}

int gfl_loss_fwd_bwd_partials_cached(const float* render, const float* gt_rgb, const float* gt_depth,
                                     const uint8_t* keep, const float* depth_ab, float lambda_rgb, float lambda_depth,
                                     int W, int H, float* d_render, float* err_px, void* workspace,
                                     size_t workspace_bytes, const float* gt_stats, const float** p_ssim, int* n_ssim,
                                     const float** p_grad, int* n_grad, gfl_stream_t stream) {
    if (!p_ssim || !n_ssim || !p_grad || !n_grad) return GFL_ERR_INVALID;
    return loss_launch(render, gt_rgb, gt_depth, keep, depth_ab, lambda_rgb, lambda_depth, W, H, d_render, err_px,
                       nullptr, workspace, workspace_bytes, stream, p_ssim, n_ssim, p_grad, n_grad, gt_stats);
}

}  // extern "C"
