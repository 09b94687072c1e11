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

// a float at a 32-bit BYTE offset from a wave-uniform base: the address costs no 64-bit vector arithmetic
__device__ __forceinline__ float ld_f32(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
// y * W + x as ONE full-rate 24-bit multiply-add (the launcher checks W, H < 2^24): a plain 32-bit multiply is a
// quarter-rate instruction, and the 64-bit multiply-add the compiler picked instead dragged a register pair with a
// don't-care high half along -- which happened to be the destination of a load in flight, a wait for memory in the
// middle of the address arithmetic
__device__ __forceinline__ unsigned pixel_index(int y, int W, int x) { return __umul24((unsigned)y, (unsigned)W) + (unsigned)x; }
__device__ __forceinline__ void st_f32(float* __restrict__ base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
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
    // The kernel is bound by VALU issue, so the maps are staged as packed pairs with the products formed ONCE per
    // staged pixel, and both filter passes run v_pk_fma_f32 on the pairs.  pair = (x, y) and (x^2, y^2) in MODE 0,
    // (x, x^2) in MODE 1, (y, y^2) in MODE 2; the fifth map xy is a plain float (MODE 0 and 1).
    // Pitches: a half-wave of the row pass reads two staged rows (8-byte reads: pitch = 16 pairs mod 32 keeps them on
    // disjoint banks), a wave four rows of the xy map (pitch = 16 floats mod 64 likewise); the row-filtered maps are read
    // by the column pass the same way with 16 columns, so they are not padded at all.
    constexpr int APITCH = 48, ROWS = SI + 1;          // row SI: where the lanes past the last staged row write
    __shared__ __attribute__((aligned(16))) v2f s_a[ROWS][APITCH];
    __shared__ __attribute__((aligned(16))) v2f s_b[MODE == 0 ? ROWS : 1][APITCH];
    __shared__ __attribute__((aligned(16))) float s_x_y[MODE == 2 ? 1 : ROWS][APITCH];
    __shared__ __attribute__((aligned(16))) v2f h_a[SI][ST];
    __shared__ __attribute__((aligned(16))) v2f h_b[MODE == 0 ? SI : 1][ST];
    __shared__ __attribute__((aligned(16))) float h_xy[MODE == 2 ? 1 : SI][ST];
    const int lb = __builtin_amdgcn_readfirstlane(xcd_logical_block(blockIdx.x, gx * gy * 3));
    const int c = lb % 3, tile = lb / 3;
    const int bx = tile % gx, by = tile / gx;
    const int x0 = bx * ST - SR, y0 = by * ST - SR;
    const int tid = threadIdx.x;
    const unsigned plane_b = (unsigned)H * (unsigned)W * 4u;       // bytes of a plane (the launcher checks 36 HW < 2^32)
    // MODE 1: the cached statistics of this lane's own pixel, requested before anything else (read where they are
    // used, after the column pass, the two loads were a full memory round trip at the end of the workgroup's life)
    float mu2_own = 0.f, e22_own = 0.f;
    if (MODE == 1) {
        const int opx = bx * ST + (tid & 15), opy = by * ST + (tid >> 4);
        if (opx < W && opy < H) {
            const unsigned ob = (pixel_index(opy, W, opx)) * 4u + 2u * c * plane_b;
            mu2_own = ld_f32(gt_stats, ob);
            e22_own = ld_f32(gt_stats, ob + plane_b);
        }
    }
    // Staging.  A lane owns ONE staged column (32 lanes across, 26 in use) and every eighth row: four passes whose
    // addresses differ by a constant, with 32-bit byte offsets from uniform bases.  ALL of a lane's loads are issued
    // before the first is used and nothing below branches: written as a loop with a load, a test and a store per pass,
    // the compiler waited for memory in every pass -- six dependent round trips in the life of a workgroup, which is what
    // this kernel's duration was made of.  Addresses outside the image are clamped to pixel 0 and the value replaced.
    constexpr int NS = 4;
    const int sq = tid & 31, srg = tid >> 5;
    const int sx = x0 + sq;
    const bool in_x = sq < SI && sx >= 0 && sx < W;
    const unsigned c_plane_b = (unsigned)c * plane_b;
    float xs[NS], ys[NS];
    int kb[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int y = y0 + srg + 8 * j;
        const bool in = in_x && (srg + 8 * j) < SI && (unsigned)y < (unsigned)H;
        const unsigned pix = in ? pixel_index(y, W, sx) : 0u;
        const unsigned pix4 = pix << 2;
        xs[j] = MODE == 2 ? 0.f : ld_f32(render, pix4 + c_plane_b);
        ys[j] = ld_f32(gt_rgb, pix4 + (pix4 << 1) + 4u * c);
        kb[j] = KEEP ? (int)keep[pix] : 1;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int y = y0 + srg + 8 * j;
        const bool k = in_x && (srg + 8 * j) < SI && (unsigned)y < (unsigned)H && kb[j] != 0;
        const float xv = k ? xs[j] : 0.f, yv = k ? ys[j] : 0.f;
        const int r = min(srg + 8 * j, SI);              // rows past the end -> the spare row (columns >= 26: padding)
        if (MODE == 0) {
            s_a[r][sq] = (v2f){xv, yv};
            s_b[r][sq] = (v2f){xv * xv, yv * yv};
        } else if (MODE == 1) {
            s_a[r][sq] = (v2f){xv, xv * xv};
        } else {
            s_a[r][sq] = (v2f){yv, yv * yv};
        }
        if (MODE != 2) s_x_y[r][sq] = xv * yv;
    }
    __syncthreads();
    // row pass: a lane produces TWO adjacent outputs of a row from twelve staged pixels read once (six 16-byte reads of
    // pairs, six 8-byte reads of xy) -- 26 x 8 = 208 lanes, one pass, a quarter of the LDS reads of one output per lane
    if (tid < SI * 8) {
        const int r = tid >> 3, q = (tid & 7) * 2;
        v2f A[12], B[12];
        float X[12];
#pragma unroll
        for (int k = 0; k < 12; k += 2) {
            const float4 t = *reinterpret_cast<const float4*>(&s_a[r][q + k]);
            A[k] = (v2f){t.x, t.y};
            A[k + 1] = (v2f){t.z, t.w};
            if (MODE == 0) {
                const float4 u = *reinterpret_cast<const float4*>(&s_b[r][q + k]);
                B[k] = (v2f){u.x, u.y};
                B[k + 1] = (v2f){u.z, u.w};
            }
            if (MODE != 2) {
                const float2 u = *reinterpret_cast<const float2*>(&s_x_y[r][q + k]);
                X[k] = u.x;
                X[k + 1] = u.y;
            }
        }
        v2f a_a[2] = {{0.f, 0.f}, {0.f, 0.f}}, a_b[2] = {{0.f, 0.f}, {0.f, 0.f}};
        float a_xy[2] = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const float w = win.w[k];
            const v2f w2 = {w, w};
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                a_a[o] = __builtin_elementwise_fma(w2, A[o + k], a_a[o]);
                if (MODE == 0) a_b[o] = __builtin_elementwise_fma(w2, B[o + k], a_b[o]);
                if (MODE != 2) a_xy[o] = fmaf(w, X[o + k], a_xy[o]);
            }
        }
        *reinterpret_cast<float4*>(&h_a[r][q]) = make_float4(a_a[0].x, a_a[0].y, a_a[1].x, a_a[1].y);
        if (MODE == 0) *reinterpret_cast<float4*>(&h_b[r][q]) = make_float4(a_b[0].x, a_b[0].y, a_b[1].x, a_b[1].y);
        if (MODE != 2) *reinterpret_cast<float2*>(&h_xy[r][q]) = make_float2(a_xy[0], a_xy[1]);
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
        const unsigned pb = (pixel_index(py, W, px)) * 4u;
        if (MODE == 2) {
            st_f32(gt_stats, pb + 2u * c * plane_b, fa.x);                    // mu2
            st_f32(gt_stats, pb + (2u * c + 1u) * plane_b, fa.y);             // E[y^2]
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
        // two hardware reciprocals (1 ulp) instead of three IEEE divisions (~10 VALU instructions each): B1 >= C1,
        // B2 ~ C2 + variances, nowhere near the denormals
        const float r1 = __builtin_amdgcn_rcpf(B1), r2 = __builtin_amdgcn_rcpf(B2);
        const float inv = r1 * r2;
        sval = A1 * A2 * inv;
        const float d_e11 = -sval * r2;
        const float d_e12 = 2.f * A1 * inv;
        const float d_mu1 = 2.f * mu2 * (A2 - A1) * inv - 2.f * mu1 * sval * (r1 - r2);
        const unsigned mb = pb + 3u * c * plane_b;       // dmaps[c][3][H][W]
        st_f32(dmaps, mb, scale * d_mu1);
        st_f32(dmaps, mb + plane_b, scale * d_e11);
        st_f32(dmaps, mb + 2u * plane_b, scale * d_e12);
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
    __shared__ __attribute__((aligned(16))) float sm[3][SI + 1][SP];   // row SI: where lanes past the last row write
    __shared__ __attribute__((aligned(16))) float hz[3][SI][HP];
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W;
    const unsigned plane_b = (unsigned)H * (unsigned)W * 4u;           // (the launcher checks 36 HW < 2^32)
    const int lb = __builtin_amdgcn_readfirstlane(xcd_logical_block(blockIdx.x, gx * gy * 4));
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
                const unsigned pix = pixel_index(py, W, px);
                own_kb[o] = KEEP ? (int)keep[pix] : 1;
                own_x[o] = ld_f32(render, pix * 4u + ch * plane_b);
                own_y[o] = ld_f32(gt_rgb, pix * 12u + ch * 4u);
            }
        }
    }
    // staging: 26 rows x 28 columns (columns 26, 27 are zero padding for the 16-byte row reads) of the three maps.  A
    // lane owns one column (32 across) and every eighth row, four passes, all twelve loads in flight together and no
    // branch below (ssim_stats_kernel explains why)
    constexpr int NG = 4;
    const int sq = tid & 31, srg = tid >> 5;
    const int sx = x0 + sq;
    const bool in_x = sq < SI && sx >= 0 && sx < W;
    const unsigned map_b = 3u * ch * plane_b;
    float dm[NG][3];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int y = y0 + srg + 8 * j;
        const bool in = in_x && (srg + 8 * j) < SI && (unsigned)y < (unsigned)H;
        const unsigned pb = (in ? (pixel_index(y, W, sx)) << 2 : 0u) + map_b;
        dm[j][0] = ld_f32(dmaps, pb);
        dm[j][1] = ld_f32(dmaps, pb + plane_b);
        dm[j][2] = ld_f32(dmaps, pb + 2u * plane_b);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int y = y0 + srg + 8 * j;
        const bool in = in_x && (srg + 8 * j) < SI && (unsigned)y < (unsigned)H;
        const int r = min(srg + 8 * j, SI);               // (columns 28..31 land in the pitch's padding)
        sm[0][r][sq] = in ? dm[j][0] : 0.f;
        sm[1][r][sq] = in ? dm[j][1] : 0.f;
        sm[2][r][sq] = in ? dm[j][2] : 0.f;
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
                float out = 0.f;
                if (own_kb[o] != 0) {
                    const float x = own_x[o], y = own_y[o];
                    out = g[0][o] + 2.f * x * g[1][o] + y * g[2][o] + mse_scale * (x - y);
                }
                st_f32(d_render, (pixel_index(py, W, px)) * 4u + ch * plane_b, out);
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
    if ((uint64_t)W * (uint64_t)H * 36u >= (1ull << 32) || W >= (1 << 24) || H >= (1 << 24))
        return GFL_ERR_INVALID;                                      // the kernels' 32-bit byte offsets
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
    if ((uint64_t)W * (uint64_t)H * 36u >= (1ull << 32) || W >= (1 << 24) || H >= (1 << 24)) return GFL_ERR_INVALID;
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
