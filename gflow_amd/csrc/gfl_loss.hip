// Device-side turbo colour map (color.py:24-44) and the stand-alone fused Adam step
// (trainer.py:153,554).  The image loss kernels live in gfl_ssim.hip.
#include "gfl_common.hpp"

namespace gfl {

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


}  // namespace gfl

using namespace gfl;

extern "C" {

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
    const int blocks = min((N + 255) / 256, 32);   // (a thousand waves hitting the two result words with atomics took 23 us)
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
