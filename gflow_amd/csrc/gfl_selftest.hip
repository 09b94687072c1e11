// Device self-tests of wave-level primitives, exported so the GPU test-suite can pin
// them independently of the kernels that use them.
#include "gfl_math.hpp"

namespace gfl {

// one wave: in[64][10] -> out[10] via wave_reduce_scatter10; out2[10] via ten wave_sum calls
__global__ void __launch_bounds__(64) selftest_reduce10_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               float* __restrict__ out2) {
    const int lane = threadIdx.x;
    float v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = in[lane * 10 + k];
    const int comp = reduce_scatter10_component(lane);
    const float mine = wave_reduce_scatter10(v, lane);
    if (comp >= 0) out[comp] = mine;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) out2[k] = s;
    }
}

// Sigma2 = M Sigma M^T of n splats both ways: m [n][6] = rows m0 | m1 of M, cov [n][6]; out_* [n][3] = a b c
__global__ void __launch_bounds__(256) selftest_cov2d_kernel(const float* __restrict__ m, const float* __restrict__ cov, int n,
                                                             float* __restrict__ out_valu, float* __restrict__ out_mfma) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float m0[3] = {0.f, 0.f, 0.f}, m1[3] = {0.f, 0.f, 0.f}, cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { m0[k] = m[i * 6 + k]; m1[k] = m[i * 6 + 3 + k]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) cv[k] = cov[i * 6 + k];
    }
    float a, b, c, a2, b2, c2;
    cov2d_valu(m0, m1, cv, a, b, c);
    cov2d_mfma(m0, m1, cv, a2, b2, c2);                 // all lanes
    if (i < n) {
        out_valu[i * 3] = a; out_valu[i * 3 + 1] = b; out_valu[i * 3 + 2] = c;
        out_mfma[i * 3] = a2; out_mfma[i * 3 + 1] = b2; out_mfma[i * 3 + 2] = c2;
    }
}

// block_mask of n records (u v A B | C o - - | - - cutoff -) against the four bs x bs boxes at (x0, y0), and, by brute
// force over the boxes' pixels with the blend kernels' own alpha test, which boxes really hold a visible pixel
__global__ void __launch_bounds__(256) selftest_block_mask_kernel(const float* __restrict__ rec, int n, int x0, int y0, int bs,
                                                                  int32_t* __restrict__ mask, int32_t* __restrict__ truth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)i * 12);
    const float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
    mask[i] = (int32_t)block_mask(p0, p1, p2.z, x0, y0, bs);
    int t = 0;
    for (int w = 0; w < 4; ++w)
        for (int y = 0; y < bs; ++y)
            for (int x = 0; x < bs; ++x) {
#pragma clang fp contract(off)
                const float fx = pixf(x0 + (w & 1) * bs + x), fy = pixf(y0 + (w >> 1) * bs + y);
                const float dx = p0.x - fx, dy = p0.y - fy;
                const float q = __builtin_fmaf(p0.z * dx, dx, (p1.x * dy) * dy);
                const float power = __builtin_fmaf(-0.5f, q, -((p0.w * dx) * dy));
                const float alpha = fminf(GFL_ALPHA_MAX, p1.y * __expf(fminf(power, 0.f)));
                if (power <= 0.f && alpha >= GFL_ALPHA_MIN) t |= 1 << w;
            }
    truth[i] = t;
}

}  // namespace gfl

extern "C" int gfl_selftest_block_mask(const float* rec, int n, int x0, int y0, int box, int32_t* mask, int32_t* truth,
                                       gfl_stream_t stream) {
    if (!rec || !mask || !truth || n < 0 || (box != 8 && box != 4)) return GFL_ERR_INVALID;
    if (n == 0) return GFL_OK;
    gfl::selftest_block_mask_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(rec, n, x0, y0, box, mask, truth);
    return gfl::check_launch();
}

extern "C" int gfl_selftest_cov2d(const float* m, const float* cov, int n, float* out_valu, float* out_mfma,
                                  gfl_stream_t stream) {
    if (!m || !cov || !out_valu || !out_mfma || n < 0) return GFL_ERR_INVALID;
    if (n == 0) return GFL_OK;
    gfl::selftest_cov2d_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(m, cov, n, out_valu, out_mfma);
    return gfl::check_launch();
}

extern "C" int gfl_selftest_reduce10(const float* in, float* out_scatter, float* out_dpp, gfl_stream_t stream) {
    if (!in || !out_scatter || !out_dpp) return GFL_ERR_INVALID;
    gfl::selftest_reduce10_kernel<<<1, 64, 0, (hipStream_t)stream>>>(in, out_scatter, out_dpp);
    return gfl::check_launch();
}

// sizes of the plain structs of the fused ABI, so FFI bindings can check their mirrors
extern "C" int gfl_abi_sizes(int* sizeof_fit_state, int* sizeof_fit_hyper) {
    if (!sizeof_fit_state || !sizeof_fit_hyper) return GFL_ERR_INVALID;
    *sizeof_fit_state = (int)sizeof(gfl_fit_state);
    *sizeof_fit_hyper = (int)sizeof(gfl_fit_hyper);
    return GFL_OK;
}
