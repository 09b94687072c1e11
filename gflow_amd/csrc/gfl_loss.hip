// Photometric (per-pixel MSE) + SSIM + scale/shift depth loss, forward AND backward
// in two stencil kernels (trainer.py:452-488, utils/pytorch_ssim.py:17-37), the
// device-side turbo colour map (color.py:24-44) and the fused Adam step
// (trainer.py:153,554).
//
// SSIM: the reference convolves x, y, x^2, y^2, xy with an 11x11 Gaussian window
// (zero padding) and then differentiates through five depthwise conv2d calls.  Here
//   pass 1 (ssim_stats): 16x16 output tile + 5-pixel halo staged in LDS, separable
//          11-tap filter of the five maps, SSIM map value and its three partial
//          derivatives (wrt mu1, E[x^2], E[xy]) written out;
//   pass 2 (loss_grad) : the same separable filter over the three derivative maps
//          (the window is symmetric, so the adjoint of the conv is the conv) and
//          dL/dx = conv(dmu1) + 2x conv(de11) + y conv(de12), fused with the MSE and
//          depth-term gradients into d_render[4,H,W].
// Every scalar sum goes block-partial -> ordered fold (bitwise reproducible).
#include "gfl_common.hpp"

#include <math.h>

namespace gfl {

constexpr int SW = 11;          // SSIM window
constexpr int SR = SW / 2;      // halo
constexpr int ST = 16;          // output tile edge
constexpr int SI = ST + 2 * SR; // staged tile edge (26)
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;

struct Win { float w[SW]; };

__device__ __forceinline__ float ld_render(const float* __restrict__ render, const uint8_t* __restrict__ keep, int c,
                                           int x, int y, int W, int H) {
    if (x < 0 || y < 0 || x >= W || y >= H) return 0.f;
    const size_t pix = (size_t)y * W + x;
    const float v = render[(size_t)c * H * W + pix];
    return (keep && !keep[pix]) ? 0.f : v;
}
__device__ __forceinline__ float ld_gt(const float* __restrict__ gt_rgb, const uint8_t* __restrict__ keep, int c, int x,
                                       int y, int W, int H) {
    if (x < 0 || y < 0 || x >= W || y >= H) return 0.f;
    const size_t pix = (size_t)y * W + x;
    const float v = gt_rgb[pix * 3 + c];
    return (keep && !keep[pix]) ? 0.f : v;
}

// grid (gx, gy, 3); dmaps[3 channels][3 maps][H][W]; partial[block] = sum of S
__global__ void __launch_bounds__(256) ssim_stats_kernel(const float* __restrict__ render,
                                                         const float* __restrict__ gt_rgb,
                                                         const uint8_t* __restrict__ keep, int W, int H, Win win,
                                                         float scale /* dL/dS per element */,
                                                         float* __restrict__ dmaps, float* __restrict__ partial) {
    __shared__ float sx[SI][SI + 1];
    __shared__ float sy[SI][SI + 1];
    __shared__ float hz[5][SI][ST + 1];
    const int c = blockIdx.z;
    const int x0 = blockIdx.x * ST - SR, y0 = blockIdx.y * ST - SR;
    const int tid = threadIdx.x;
    for (int i = tid; i < SI * SI; i += 256) {
        const int r = i / SI, q = i - r * SI;
        sx[r][q] = ld_render(render, keep, c, x0 + q, y0 + r, W, H);
        sy[r][q] = ld_gt(gt_rgb, keep, c, x0 + q, y0 + r, W, H);
    }
    __syncthreads();
    for (int i = tid; i < SI * ST; i += 256) {
        const int r = i / ST, q = i - r * ST;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const float xv = sx[r][q + k], yv = sy[r][q + k], w = win.w[k];
            a0 = fmaf(w, xv, a0);
            a1 = fmaf(w, yv, a1);
            a2 = fmaf(w, xv * xv, a2);
            a3 = fmaf(w, yv * yv, a3);
            a4 = fmaf(w, xv * yv, a4);
        }
        hz[0][r][q] = a0; hz[1][r][q] = a1; hz[2][r][q] = a2; hz[3][r][q] = a3; hz[4][r][q] = a4;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int px = blockIdx.x * ST + lx, py = blockIdx.y * ST + ly;
    float sval = 0.f;
    if (px < W && py < H) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const float w = win.w[k];
            mu1 = fmaf(w, hz[0][ly + k][lx], mu1);
            mu2 = fmaf(w, hz[1][ly + k][lx], mu2);
            e11 = fmaf(w, hz[2][ly + k][lx], e11);
            e22 = fmaf(w, hz[3][ly + k][lx], e22);
            e12 = fmaf(w, hz[4][ly + k][lx], e12);
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
        const size_t plane = (size_t)H * W, pix = (size_t)py * W + px;
        float* base = dmaps + (size_t)c * 3 * plane;
        base[pix] = scale * d_mu1;
        base[plane + pix] = scale * d_e11;
        base[2 * plane + pix] = scale * d_e12;
    }
    float v[1] = {sval};
    const int bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    // block_reduce_store indexes by blockIdx.x only -> reduce by hand here
    __shared__ float red[4];
    const float s = wave_sum_to_lane63(v[0]);
    if ((tid & 63) == 63) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) partial[bid] = red[0] + red[1] + red[2] + red[3];
}

// grid (gx, gy, 4): z<3 -> rgb channel gradient; z==3 -> depth plane + per-pixel mse.
// partial rows: [block][4] = {sum mse_px, sum depth term, d/d depth_a, d/d depth_b}
__global__ void __launch_bounds__(256) loss_grad_kernel(
    const float* __restrict__ render, const float* __restrict__ gt_rgb, const float* __restrict__ gt_depth,
    const uint8_t* __restrict__ keep, const float* __restrict__ depth_ab, const float* __restrict__ dmaps, int W, int H,
    Win win, float mse_scale /* lambda_rgb * 2/(3HW) */, float depth_scale /* lambda_depth/(HW) */,
    float* __restrict__ d_render, float* __restrict__ err_px, float* __restrict__ partial) {
    __shared__ float sm[3][SI][SI + 1];
    __shared__ float hz[3][SI][ST + 1];
    const int tid = threadIdx.x;
    const int lx = tid & 15, ly = tid >> 4;
    const int px = blockIdx.x * ST + lx, py = blockIdx.y * ST + ly;
    const bool inside = px < W && py < H;
    const size_t plane = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;
    const int bid2 = blockIdx.y * gridDim.x + blockIdx.x;
    if (blockIdx.z == 3) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (inside) {
            const bool k = !(keep && !keep[pix]);
            float e = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = k ? (render[c * plane + pix] - gt_rgb[pix * 3 + c]) : 0.f;
                e = fmaf(d, d, e);
            }
            e *= (1.f / 3.f);
            err_px[pix] = e;
            v[0] = e;
            float gD = 0.f;
            if (depth_scale != 0.f) {
                const float a = depth_ab[0], b = depth_ab[1];
                const float D = render[3 * plane + pix];
                const float d = fmaf(a, D, b), gt = gt_depth[pix];
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
        if (tid < 4) partial[(size_t)bid2 * 4 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        return;
    }
    const int c = blockIdx.z;
    const int x0 = blockIdx.x * ST - SR, y0 = blockIdx.y * ST - SR;
    const float* base = dmaps + (size_t)c * 3 * plane;
    for (int i = tid; i < SI * SI; i += 256) {
        const int r = i / SI, q = i - r * SI;
        const int x = x0 + q, y = y0 + r;
        const bool in = x >= 0 && y >= 0 && x < W && y < H;
        const size_t p = (size_t)y * W + x;
        sm[0][r][q] = in ? base[p] : 0.f;
        sm[1][r][q] = in ? base[plane + p] : 0.f;
        sm[2][r][q] = in ? base[2 * plane + p] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SI * ST; i += 256) {
        const int r = i / ST, q = i - r * ST;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const float w = win.w[k];
            a0 = fmaf(w, sm[0][r][q + k], a0);
            a1 = fmaf(w, sm[1][r][q + k], a1);
            a2 = fmaf(w, sm[2][r][q + k], a2);
        }
        hz[0][r][q] = a0; hz[1][r][q] = a1; hz[2][r][q] = a2;
    }
    __syncthreads();
    if (inside) {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int k = 0; k < SW; ++k) {
            const float w = win.w[k];
            g0 = fmaf(w, hz[0][ly + k][lx], g0);
            g1 = fmaf(w, hz[1][ly + k][lx], g1);
            g2 = fmaf(w, hz[2][ly + k][lx], g2);
        }
        const bool k = !(keep && !keep[pix]);
        float out = 0.f;
        if (k) {
            const float x = render[c * plane + pix], y = gt_rgb[pix * 3 + c];
            out = g0 + 2.f * x * g1 + y * g2 + mse_scale * (x - y);
        }
        d_render[c * plane + pix] = out;
    }
}

// sums[8] from the two partial arrays (single block, fixed-shape tree)
__global__ void __launch_bounds__(256) loss_fold_kernel(const float* __restrict__ p_ssim, int n_ssim,
                                                        const float* __restrict__ p_grad, int n_grad,
                                                        float* __restrict__ sums) {
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = threadIdx.x; r < n_ssim; r += 256) acc[1] += p_ssim[r];
    for (int r = threadIdx.x; r < n_grad; r += 256) {
        acc[0] += p_grad[4 * r]; acc[2] += p_grad[4 * r + 1]; acc[3] += p_grad[4 * r + 2]; acc[4] += p_grad[4 * r + 3];
    }
    __shared__ float red[4][5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float s = wave_sum_to_lane63(acc[q]);
        if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6][q] = s;
    }
    __syncthreads();
    if (threadIdx.x < 8)
        sums[threadIdx.x] = threadIdx.x < 5 ? red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x] : 0.f;
}

// ------------------------------------------------------------------ colour map
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void __launch_bounds__(256) cmap_range_kernel(const float* __restrict__ v, int N, unsigned* __restrict__ mm) {
    unsigned lo = 0xffffffffu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float x = v[i];
        const unsigned k = f2ord(x);
        if (x != 0.f) lo = min(lo, k);
        hi = max(hi, k);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, off));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

__global__ void __launch_bounds__(256) cmap_apply_kernel(const float* __restrict__ v, int N,
                                                         const unsigned* __restrict__ mm,
                                                         const float* __restrict__ lut, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float lo = (mm[0] == 0xffffffffu) ? 0.f : ord2f(mm[0]);
    const float hi = ord2f(mm[1]) - lo;
    float x = (v[i] - lo) / (hi + 1e-5f);
    x = fminf(fmaxf(x, 0.f), 1.f);
    if (x != x) x = 0.f;
    const int idx = (int)(x * 255.f);
    out[3 * i] = lut[3 * idx]; out[3 * i + 1] = lut[3 * idx + 1]; out[3 * i + 2] = lut[3 * idx + 2];
}

// ------------------------------------------------------------------------ Adam
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, int row_len,
                                                   const uint8_t* __restrict__ row_zero, float lr, float b1, float b2,
                                                   float eps, const int32_t* __restrict__ d_step, float lr_end_factor,
                                                   int total_iters) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = *d_step;
    const float t = (float)(e + 1);
    if (total_iters > 0) lr *= 1.f + (lr_end_factor - 1.f) * (float)min(e, total_iters) / (float)total_iters;
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    float g = grad[i];
    if (row_zero && row_zero[i / row_len]) g = 0.f;
    const float mi = fmaf(b1, m[i], (1.f - b1) * g);
    const float vi = fmaf(b2, v[i], (1.f - b2) * g * g);
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

__global__ void step_inc_kernel(int32_t* d_step) { *d_step += 1; }

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

int gfl_loss_fwd_bwd(const float* render, const float* gt_rgb, const float* gt_depth, const uint8_t* keep,
                     const float* depth_ab, float lambda_rgb, float lambda_depth, int W, int H, float* d_render,
                     float* err_px, float* sums, void* workspace, size_t workspace_bytes, gfl_stream_t stream) {
    if (W <= 0 || H <= 0 || !render || !gt_rgb || !d_render || !err_px || !sums || !workspace) return GFL_ERR_INVALID;
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
    ssim_stats_kernel<<<dim3(gx, gy, 3), 256, 0, s>>>(render, gt_rgb, keep, W, H, win, -lambda_rgb / (3.f * hw), dmaps,
                                                      p_ssim);
    loss_grad_kernel<<<dim3(gx, gy, 4), 256, 0, s>>>(render, gt_rgb, gt_depth, keep, depth_ab, dmaps, W, H, win,
                                                     lambda_rgb * 2.f / (3.f * hw), lambda_depth / hw, d_render, err_px,
                                                     p_grad);
    loss_fold_kernel<<<1, 256, 0, s>>>(p_ssim, gx * gy * 3, p_grad, gx * gy, sums);
    return check_launch();
}

int gfl_colormap_nonzero(const float* value, int N, const float* lut, float* out, void* workspace,
                         size_t workspace_bytes, gfl_stream_t stream) {
    if (N < 0 || !lut || !workspace) return GFL_ERR_INVALID;
    if (workspace_bytes < 16) return GFL_ERR_WORKSPACE;
    if (N == 0) return GFL_OK;
    if (!value || !out) return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    unsigned* mm = (unsigned*)workspace;
    int rc = check(hipMemsetAsync(mm, 0xff, 4, s));
    if (!rc) rc = check(hipMemsetAsync(mm + 1, 0, 4, s));
    if (rc) return rc;
    const int blocks = min((N + 255) / 256, 1024);
    cmap_range_kernel<<<blocks, 256, 0, s>>>(value, N, mm);
    cmap_apply_kernel<<<(N + 255) / 256, 256, 0, s>>>(value, N, mm, lut, out);
    return check_launch();
}

int gfl_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, int row_len,
                  const uint8_t* row_zero_grad, float lr, float beta1, float beta2, float eps, const int32_t* d_step,
                  float lr_end_factor, int total_iters, gfl_stream_t stream) {
    if (n < 0 || row_len <= 0 || !d_step) return GFL_ERR_INVALID;
    if (n == 0) return GFL_OK;
    if (!param || !grad || !m || !v) return GFL_ERR_INVALID;
    adam_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(param, grad, m, v, n, row_len, row_zero_grad,
                                                                              lr, beta1, beta2, eps, d_step, lr_end_factor,
                                                                              total_iters);
    return check_launch();
}

int gfl_step_increment(int32_t* d_step, gfl_stream_t stream) {
    if (!d_step) return GFL_ERR_INVALID;
    step_inc_kernel<<<1, 1, 0, (hipStream_t)stream>>>(d_step);
    return check_launch();
}

}  // extern "C"
