// A17  real spherical-harmonics colour evaluation, degrees 0..3 (optional operator).
// GFlow itself never calls msplat.compute_sh (its colour is sigmoid(rgb), trainer.py:68); the
// operator exists because the msplat surface has it.  Basis and constants: the real SH basis in
// the sign convention of the 3D Gaussian Splatting code base (Kerbl et al. 2023), i.e. the
// orthonormal real harmonics with the Condon-Shortley phase folded into the constants.
// One lane per splat; K = (degree + 1)^2 coefficients per colour channel, shs[N][K][3].
#include "gfl_common.hpp"

namespace gfl {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

// basis values b[0..K) and, when GRAD, their derivatives wrt x, y, z
template <bool GRAD>
__device__ __forceinline__ void sh_basis(int K, float x, float y, float z, float* b, float* bx, float* by, float* bz) {
    b[0] = SH_C0;
    if (GRAD) { bx[0] = by[0] = bz[0] = 0.f; }
    if (K < 4) return;
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (GRAD) {
        bx[1] = 0.f; by[1] = -SH_C1; bz[1] = 0.f;
        bx[2] = 0.f; by[2] = 0.f; bz[2] = SH_C1;
        bx[3] = -SH_C1; by[3] = 0.f; bz[3] = 0.f;
    }
    if (K < 9) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy); b[7] = SH_C2_3 * xz;
    b[8] = SH_C2_4 * (xx - yy);
    if (GRAD) {
        bx[4] = SH_C2_0 * y; by[4] = SH_C2_0 * x; bz[4] = 0.f;
        bx[5] = 0.f; by[5] = SH_C2_1 * z; bz[5] = SH_C2_1 * y;
        bx[6] = -2.f * SH_C2_2 * x; by[6] = -2.f * SH_C2_2 * y; bz[6] = 4.f * SH_C2_2 * z;
        bx[7] = SH_C2_3 * z; by[7] = 0.f; bz[7] = SH_C2_3 * x;
        bx[8] = 2.f * SH_C2_4 * x; by[8] = -2.f * SH_C2_4 * y; bz[8] = 0.f;
    }
    if (K < 16) return;
    b[9] = SH_C3_0 * y * (3.f * xx - yy);
    b[10] = SH_C3_1 * xy * z;
    b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
    b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
    b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
    b[14] = SH_C3_5 * z * (xx - yy);
    b[15] = SH_C3_6 * x * (xx - 3.f * yy);
    if (GRAD) {
        bx[9] = SH_C3_0 * 6.f * xy; by[9] = SH_C3_0 * (3.f * xx - 3.f * yy); bz[9] = 0.f;
        bx[10] = SH_C3_1 * yz; by[10] = SH_C3_1 * xz; bz[10] = SH_C3_1 * xy;
        bx[11] = SH_C3_2 * (-2.f * xy); by[11] = SH_C3_2 * (4.f * zz - xx - 3.f * yy); bz[11] = SH_C3_2 * 8.f * yz;
        bx[12] = SH_C3_3 * (-6.f * xz); by[12] = SH_C3_3 * (-6.f * yz); bz[12] = SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
        bx[13] = SH_C3_4 * (4.f * zz - 3.f * xx - yy); by[13] = SH_C3_4 * (-2.f * xy); bz[13] = SH_C3_4 * 8.f * xz;
        bx[14] = SH_C3_5 * 2.f * xz; by[14] = SH_C3_5 * (-2.f * yz); bz[14] = SH_C3_5 * (xx - yy);
        bx[15] = SH_C3_6 * (3.f * xx - 3.f * yy); by[15] = SH_C3_6 * (-6.f * xy); bz[15] = 0.f;
    }
}

__global__ void __launch_bounds__(256) sh_fwd_kernel(const float* __restrict__ shs, const float* __restrict__ dirs,
                                                     const uint8_t* __restrict__ visible, int N, int K,
                                                     float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float r = 0.f, g = 0.f, bl = 0.f;
    if (!visible || visible[i]) {
        float b[16];
        sh_basis<false>(K, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], b, nullptr, nullptr, nullptr);
        const float* c = shs + (size_t)i * K * 3;
        for (int k = 0; k < K; ++k) {
            r = fmaf(b[k], c[3 * k], r);
            g = fmaf(b[k], c[3 * k + 1], g);
            bl = fmaf(b[k], c[3 * k + 2], bl);
        }
    }
    out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = bl;
}

__global__ void __launch_bounds__(256) sh_bwd_kernel(const float* __restrict__ shs, const float* __restrict__ dirs,
                                                     const uint8_t* __restrict__ visible,
                                                     const float* __restrict__ d_out, int N, int K,
                                                     float* __restrict__ d_shs, float* __restrict__ d_dirs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float* dc = d_shs + (size_t)i * K * 3;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (!visible || visible[i]) {
        float b[16], bx[16], by[16], bz[16];
        sh_basis<true>(K, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], b, bx, by, bz);
        const float* c = shs + (size_t)i * K * 3;
        const float dr = d_out[3 * i], dg = d_out[3 * i + 1], db = d_out[3 * i + 2];
        for (int k = 0; k < K; ++k) {
            dc[3 * k] = b[k] * dr; dc[3 * k + 1] = b[k] * dg; dc[3 * k + 2] = b[k] * db;
            const float s = c[3 * k] * dr + c[3 * k + 1] * dg + c[3 * k + 2] * db;
            gx = fmaf(bx[k], s, gx); gy = fmaf(by[k], s, gy); gz = fmaf(bz[k], s, gz);
        }
    } else {
        for (int k = 0; k < 3 * K; ++k) dc[k] = 0.f;
    }
    if (d_dirs) { d_dirs[3 * i] = gx; d_dirs[3 * i + 1] = gy; d_dirs[3 * i + 2] = gz; }
}

}  // namespace gfl

using namespace gfl;

extern "C" {

static bool sh_k_ok(int K) { return K == 1 || K == 4 || K == 9 || K == 16; }

int gfl_sh_fwd(const float* shs, const float* dirs, const uint8_t* visible, int N, int K, float* out,
               gfl_stream_t stream) {
    if (N < 0 || !sh_k_ok(K)) return GFL_ERR_INVALID;
    if (N == 0) return GFL_OK;
    if (!shs || !dirs || !out) return GFL_ERR_INVALID;
    sh_fwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(shs, dirs, visible, N, K, out);
    return check_launch();
}

int gfl_sh_bwd(const float* shs, const float* dirs, const uint8_t* visible, const float* d_out, int N, int K,
               float* d_shs, float* d_dirs, gfl_stream_t stream) {
    if (N < 0 || !sh_k_ok(K)) return GFL_ERR_INVALID;
    if (N == 0) return GFL_OK;
    if (!shs || !dirs || !d_out || !d_shs) return GFL_ERR_INVALID;
    sh_bwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(shs, dirs, visible, d_out, N, K, d_shs, d_dirs);
    return check_launch();
}

}  // extern "C"
