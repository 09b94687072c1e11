// Per-splat geometry: projection, 3-D covariance, EWA projection -- forward and
// analytic backward.  One set of inline device functions shared by the stand-alone
// operators (gfl_geom.hip) and the fused preprocess kernels (gfl_fused.hip), so both
// paths produce bit-identical numbers.
#pragma once
#include "gfl_common.hpp"

namespace gfl {

// ------------------------------------------------------------------ project (A4)
struct Proj {
    float px, py, pz;  // camera-space point
    float u, v;        // pixel coordinates (valid only if vis)
    bool vis;
};

__device__ __forceinline__ Proj project_fwd(const Cam& c, float x, float y, float z, int W, int H, float nearest,
                                            float extent) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel these are inlined into, and the oracle's arithmetic)
    Proj p;
    p.px = c.r00 * x + c.r01 * y + c.r02 * z + c.t0;
    p.py = c.r10 * x + c.r11 * y + c.r12 * z + c.t1;
    p.pz = c.r20 * x + c.r21 * y + c.r22 * z + c.t2;
    const bool front = p.pz > nearest;
    const float zs = front ? p.pz : 1.0f;
    p.u = c.fx * p.px / zs + c.cx;
    p.v = c.fy * p.py / zs + c.cy;
    const float lo_u = (1.0f - extent) * 0.5f * (float)W, hi_u = (1.0f + extent) * 0.5f * (float)W;
    const float lo_v = (1.0f - extent) * 0.5f * (float)H, hi_v = (1.0f + extent) * 0.5f * (float)H;
    p.vis = front && p.u >= lo_u && p.u <= hi_u && p.v >= lo_v && p.v <= hi_v;
    return p;
}

// Camera-space gradient from (du, dv, ddepth); pc = camera-space point.
__device__ __forceinline__ void project_bwd_cam(const Cam& c, float px, float py, float pz, float du, float dv,
                                                float dd, float& gx, float& gy, float& gz) {
    const float iz = 1.0f / pz;
    gx = du * c.fx * iz;
    gy = dv * c.fy * iz;
    gz = dd - (du * c.fx * px + dv * c.fy * py) * iz * iz;
}

// Fold a camera-space point gradient g into world xyz gradient and the 12 extr
// gradient accumulators e[] (row-major 3x4: R | t).
__device__ __forceinline__ void cam_grad_to_world(const Cam& c, float x, float y, float z, float gx, float gy,
                                                  float gz, float& dx, float& dy, float& dz, float (&e)[12]) {
    dx += c.r00 * gx + c.r10 * gy + c.r20 * gz;
    dy += c.r01 * gx + c.r11 * gy + c.r21 * gz;
    dz += c.r02 * gx + c.r12 * gy + c.r22 * gz;
    e[0] += gx * x; e[1] += gx * y; e[2] += gx * z; e[3] += gx;
    e[4] += gy * x; e[5] += gy * y; e[6] += gy * z; e[7] += gy;
    e[8] += gz * x; e[9] += gz * y; e[10] += gz * z; e[11] += gz;
}

// ------------------------------------------------------------------- cov3d (A5)
// q = (w,x,y,z); Sigma = R diag(s^2) R^T; out = xx,xy,xz,yy,yz,zz
__device__ __forceinline__ void quat_rot(float w, float x, float y, float z, float (&R)[9]) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel these are inlined into, and the oracle's arithmetic)
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z); R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y); R[7] = 2.f * (y * z + w * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_fwd(const float (&s)[3], const float (&q)[4], float (&cov)[6]) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel these are inlined into, and the oracle's arithmetic)
    float R[9];
    quat_rot(q[0], q[1], q[2], q[3], R);
    const float a = s[0] * s[0], b = s[1] * s[1], c = s[2] * s[2];
    cov[0] = a * R[0] * R[0] + b * R[1] * R[1] + c * R[2] * R[2];
    cov[1] = a * R[0] * R[3] + b * R[1] * R[4] + c * R[2] * R[5];
    cov[2] = a * R[0] * R[6] + b * R[1] * R[7] + c * R[2] * R[8];
    cov[3] = a * R[3] * R[3] + b * R[4] * R[4] + c * R[5] * R[5];
    cov[4] = a * R[3] * R[6] + b * R[4] * R[7] + c * R[5] * R[8];
    cov[5] = a * R[6] * R[6] + b * R[7] * R[7] + c * R[8] * R[8];
}

// g[6] = dL/d(stored entries).  Outputs ds[3], dq[4] (w,x,y,z).
__device__ __forceinline__ void cov3d_bwd(const float (&s)[3], const float (&q)[4], const float (&g)[6],
                                          float (&ds)[3], float (&dq)[4]) {
    float R[9];
    quat_rot(q[0], q[1], q[2], q[3], R);
    // full symmetric gradient: off-diagonals split in two
    const float G00 = g[0], G01 = 0.5f * g[1], G02 = 0.5f * g[2], G11 = g[3], G12 = 0.5f * g[4], G22 = g[5];
    float dR[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float r0 = R[k], r1 = R[3 + k], r2 = R[6 + k];  // column k of R
        const float h0 = G00 * r0 + G01 * r1 + G02 * r2;
        const float h1 = G01 * r0 + G11 * r1 + G12 * r2;
        const float h2 = G02 * r0 + G12 * r1 + G22 * r2;
        ds[k] = 2.f * s[k] * (r0 * h0 + r1 * h1 + r2 * h2);
        const float s2 = 2.f * s[k] * s[k];
        dR[k] = s2 * h0; dR[3 + k] = s2 * h1; dR[6 + k] = s2 * h2;
    }
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    dq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    dq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - w * dR[5] + z * dR[6] + w * dR[7] - 2.f * x * dR[8]);
    dq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + w * dR[2] + x * dR[3] + z * dR[5] - w * dR[6] + z * dR[7] - 2.f * y * dR[8]);
    dq[3] = 2.f * (-2.f * z * dR[0] - w * dR[1] + x * dR[2] + w * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
}

// --------------------------------------------------------------------- EWA (A6)

struct Ewa {
    float a, b, c, det;          // 2-D covariance (+ low-pass) and determinant
    float m0[3], m1[3];          // rows of J*W
    float j00, j02, j11, j12;
    float z;
    bool clamp_x, clamp_y;
    bool ok;                     // det != 0
    float lam;                   // larger eigenvalue (floored discriminant)
};

// ---- the 2x3 . 3x3 . 3x2 contraction Sigma2 = M Sigma M^T (before the low-pass), two ways --------------------
// VALU: 30 multiply-adds per splat in the lane that owns the splat.
__device__ __forceinline__ void cov2d_valu(const float (&m0)[3], const float (&m1)[3], const float (&cov)[6], float& a,
                                           float& b, float& c) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel these are inlined into, and the oracle's arithmetic)
    const float s00 = cov[0] * m0[0] + cov[1] * m0[1] + cov[2] * m0[2];
    const float s01 = cov[1] * m0[0] + cov[3] * m0[1] + cov[4] * m0[2];
    const float s02 = cov[2] * m0[0] + cov[4] * m0[1] + cov[5] * m0[2];
    const float s10 = cov[0] * m1[0] + cov[1] * m1[1] + cov[2] * m1[2];
    const float s11 = cov[1] * m1[0] + cov[3] * m1[1] + cov[4] * m1[2];
    const float s12 = cov[2] * m1[0] + cov[4] * m1[1] + cov[5] * m1[2];
    a = m0[0] * s00 + m0[1] * s01 + m0[2] * s02;
    b = m0[0] * s10 + m0[1] * s11 + m0[2] * s12;
    c = m1[0] * s10 + m1[1] * s11 + m1[2] * s12;
}

// MFMA (north_star: "MFMA only for the 3x3 covariance J Sigma J^T contraction"): v_mfma_f32_4x4x1_16b_f32 multiplies
// sixteen independent 4x4 blocks per instruction -- block q = lanes 4q..4q+3, lane 4q+i supplies A[i], lane 4q+j
// supplies B[j], lane 4q+j receives D[i][j] in register i.  A lane owns ONE splat here, so the sixteen splats of a pass
// (those in lane 4q+P of every quad, P = 0..3) first have to be spread over their quads: one DPP quad broadcast plus
// one select per operand element.  Per wave: 4 passes x (6 MFMA + ~45 DPP / select) against 30 FMAs per lane.
// Must be called by all 64 lanes (lanes without a splat pass zeros).  Exact f32 (an fmaf chain per element).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int P>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), P | (P << 2) | (P << 4) | (P << 6),
                                                                 0xF, 0xF, true));
}
template <int P>
__device__ __forceinline__ void cov2d_mfma_pass(const float (&m0)[3], const float (&m1)[3], const float (&cov)[6], int r,
                                                float& a, float& b, float& c) {
    // rows of the 4x4-padded M of the quad's splat P: lane r holds row r (rows 2, 3 are zero)
    float Mk[3], Sk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x0 = quad_bcast<P>(m0[k]), x1 = quad_bcast<P>(m1[k]);
        Mk[k] = r == 0 ? x0 : (r == 1 ? x1 : 0.f);
    }
    // Sigma row k, column r: (c0 c1 c2 | c1 c3 c4 | c2 c4 c5)
    const float c0 = quad_bcast<P>(cov[0]), c1 = quad_bcast<P>(cov[1]), c2 = quad_bcast<P>(cov[2]), c3 = quad_bcast<P>(cov[3]),
                c4 = quad_bcast<P>(cov[4]), c5 = quad_bcast<P>(cov[5]);
    Sk[0] = r == 0 ? c0 : (r == 1 ? c1 : (r == 2 ? c2 : 0.f));
    Sk[1] = r == 0 ? c1 : (r == 1 ? c3 : (r == 2 ? c4 : 0.f));
    Sk[2] = r == 0 ? c2 : (r == 1 ? c4 : (r == 2 ? c5 : 0.f));
    f32x4_t T = {0.f, 0.f, 0.f, 0.f};                       // T = M Sigma: T[i][j] in register i of lane 4q+j
#pragma unroll
    for (int k = 0; k < 3; ++k) T = __builtin_amdgcn_mfma_f32_4x4x1f32(Mk[k], Sk[k], T, 0, 0, 0);
    f32x4_t D = {0.f, 0.f, 0.f, 0.f};                       // D = T M^T: A[i] = T[i][k] lives in register i of lane 4q+k
    {
        const float t0 = quad_bcast<0>(T[0]), t1 = quad_bcast<0>(T[1]);
        D = __builtin_amdgcn_mfma_f32_4x4x1f32(r == 0 ? t0 : (r == 1 ? t1 : 0.f), Mk[0], D, 0, 0, 0);
    }
    {
        const float t0 = quad_bcast<1>(T[0]), t1 = quad_bcast<1>(T[1]);
        D = __builtin_amdgcn_mfma_f32_4x4x1f32(r == 0 ? t0 : (r == 1 ? t1 : 0.f), Mk[1], D, 0, 0, 0);
    }
    {
        const float t0 = quad_bcast<2>(T[0]), t1 = quad_bcast<2>(T[1]);
        D = __builtin_amdgcn_mfma_f32_4x4x1f32(r == 0 ? t0 : (r == 1 ? t1 : 0.f), Mk[2], D, 0, 0, 0);
    }
    // Sigma2 = [[D00, D01], [., D11]]: D[0][0] in lane 4q+0 register 0, D[0][1] in lane 4q+1 register 0, D[1][1] in
    // lane 4q+1 register 1 -> back to the lane that owns the splat
    const float ra = quad_bcast<0>(D[0]), rb = quad_bcast<1>(D[0]), rc = quad_bcast<1>(D[1]);
    if (r == P) { a = ra; b = rb; c = rc; }
}
__device__ __forceinline__ void cov2d_mfma(const float (&m0)[3], const float (&m1)[3], const float (&cov)[6], float& a,
                                           float& b, float& c) {
    const int r = threadIdx.x & 3;
    a = b = c = 0.f;
    cov2d_mfma_pass<0>(m0, m1, cov, r, a, b, c);
    cov2d_mfma_pass<1>(m0, m1, cov, r, a, b, c);
    cov2d_mfma_pass<2>(m0, m1, cov, r, a, b, c);
    cov2d_mfma_pass<3>(m0, m1, cov, r, a, b, c);
}

__device__ __forceinline__ void ewa_finish(Ewa& e, float a, float b, float c) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel these are inlined into, and the oracle's arithmetic)
    e.a = a + GFL_LOWPASS;
    e.b = b;
    e.c = c + GFL_LOWPASS;
    e.det = e.a * e.c - e.b * e.b;
    e.ok = e.det != 0.0f;
    const float mid = 0.5f * (e.a + e.c);
    e.lam = mid + sqrtf(fmaxf(mid * mid - e.det, GFL_EIG_FLOOR));
}

__device__ __forceinline__ void ewa_jacobian(const Cam& c, float px, float py, float pz, int W, int H, Ewa& e) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel these are inlined into, and the oracle's arithmetic)
    e.z = pz;
    const float limx = GFL_FOV_CLAMP * (float)W / (2.0f * c.fx);
    const float limy = GFL_FOV_CLAMP * (float)H / (2.0f * c.fy);
    const float xz = px / pz, yz = py / pz;
    e.clamp_x = (xz > limx) || (xz < -limx);
    e.clamp_y = (yz > limy) || (yz < -limy);
    const float tx = fmaxf(fminf(xz, limx), -limx) * pz;
    const float ty = fmaxf(fminf(yz, limy), -limy) * pz;
    e.j00 = c.fx / pz;
    e.j02 = -c.fx * tx / (pz * pz);
    e.j11 = c.fy / pz;
    e.j12 = -c.fy * ty / (pz * pz);
    e.m0[0] = e.j00 * c.r00 + e.j02 * c.r20; e.m0[1] = e.j00 * c.r01 + e.j02 * c.r21; e.m0[2] = e.j00 * c.r02 + e.j02 * c.r22;
    e.m1[0] = e.j11 * c.r10 + e.j12 * c.r20; e.m1[1] = e.j11 * c.r11 + e.j12 * c.r21; e.m1[2] = e.j11 * c.r12 + e.j12 * c.r22;
}

__device__ __forceinline__ Ewa ewa_fwd(const Cam& c, float px, float py, float pz, const float (&cov)[6], int W,
                                       int H) {
    Ewa e;
    ewa_jacobian(c, px, py, pz, W, H, e);
    float a, b, cc;
    cov2d_valu(e.m0, e.m1, cov, a, b, cc);
    ewa_finish(e, a, b, cc);
    return e;
}

// the same with the contraction on the matrix cores; ALL 64 lanes of the wave must call it (vis = the lane has a splat)
__device__ __forceinline__ Ewa ewa_fwd_mfma(const Cam& c, bool vis, float px, float py, float pz, const float (&cov_in)[6],
                                            int W, int H) {
    Ewa e;
    ewa_jacobian(c, px, py, vis ? pz : 1.f, W, H, e);
    float m0[3], m1[3], cov[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { m0[k] = vis ? e.m0[k] : 0.f; m1[k] = vis ? e.m1[k] : 0.f; }
#pragma unroll
    for (int k = 0; k < 6; ++k) cov[k] = vis ? cov_in[k] : 0.f;
    float a, b, cc;
    cov2d_mfma(m0, m1, cov, a, b, cc);
    ewa_finish(e, a, b, cc);
    return e;
}

__device__ __forceinline__ int ewa_radius(const Ewa& e) { return (int)ceilf(GFL_RADIUS_SIGMA * sqrtf(e.lam)); }

// 3DGS getRect: C float->int truncation, clamped to the tile grid.
__device__ __forceinline__ void tile_rect(float u, float v, int radius, int gx, int gy, int& x0, int& x1, int& y0,
                                          int& y1) {
    const float r = (float)radius;
    x0 = min(gx, max(0, (int)((u - r) / (float)GFL_TILE)));
    x1 = min(gx, max(0, (int)((u + r + (float)(GFL_TILE - 1)) / (float)GFL_TILE)));
    y0 = min(gy, max(0, (int)((v - r) / (float)GFL_TILE)));
    y1 = min(gy, max(0, (int)((v + r + (float)(GFL_TILE - 1)) / (float)GFL_TILE)));
}

// Backward of conic = [c/det, -b/det, a/det] through Sigma2 = M Sigma M^T + 0.3 I.
// In: dA,dB,dC (true gradients wrt the three conic entries).  Out: gcov[6] (wrt
// stored cov3d entries), camera-space point gradient (gx,gy,gz) and the rotation
// part of the extr gradient accumulated into e[] (entries 0..2,4..6,8..10).
__device__ __forceinline__ void ewa_bwd(const Cam& c, const Ewa& f, float px, float py, const float (&cov)[6],
                                        float dA, float dB, float dC, float (&gcov)[6], float& gx, float& gy,
                                        float& gz, float (&e)[12]) {
    const float inv2 = 1.0f / (f.det * f.det);
    const float da = inv2 * (-f.c * f.c * dA + f.b * f.c * dB + (f.det - f.a * f.c) * dC);
    const float dc = inv2 * (-f.a * f.a * dC + f.a * f.b * dB + (f.det - f.a * f.c) * dA);
    const float db = inv2 * (2.f * f.b * f.c * dA - (f.det + 2.f * f.b * f.b) * dB + 2.f * f.a * f.b * dC);
    const float* m0 = f.m0;
    const float* m1 = f.m1;
    gcov[0] = da * m0[0] * m0[0] + db * m0[0] * m1[0] + dc * m1[0] * m1[0];
    gcov[3] = da * m0[1] * m0[1] + db * m0[1] * m1[1] + dc * m1[1] * m1[1];
    gcov[5] = da * m0[2] * m0[2] + db * m0[2] * m1[2] + dc * m1[2] * m1[2];
    gcov[1] = 2.f * da * m0[0] * m0[1] + db * (m0[0] * m1[1] + m0[1] * m1[0]) + 2.f * dc * m1[0] * m1[1];
    gcov[2] = 2.f * da * m0[0] * m0[2] + db * (m0[0] * m1[2] + m0[2] * m1[0]) + 2.f * dc * m1[0] * m1[2];
    gcov[4] = 2.f * da * m0[1] * m0[2] + db * (m0[1] * m1[2] + m0[2] * m1[1]) + 2.f * dc * m1[1] * m1[2];
    // Sigma*m0, Sigma*m1
    const float s00 = cov[0] * m0[0] + cov[1] * m0[1] + cov[2] * m0[2];
    const float s01 = cov[1] * m0[0] + cov[3] * m0[1] + cov[4] * m0[2];
    const float s02 = cov[2] * m0[0] + cov[4] * m0[1] + cov[5] * m0[2];
    const float s10 = cov[0] * m1[0] + cov[1] * m1[1] + cov[2] * m1[2];
    const float s11 = cov[1] * m1[0] + cov[3] * m1[1] + cov[4] * m1[2];
    const float s12 = cov[2] * m1[0] + cov[4] * m1[1] + cov[5] * m1[2];
    const float d00 = 2.f * da * s00 + db * s10, d01 = 2.f * da * s01 + db * s11, d02 = 2.f * da * s02 + db * s12;
    const float d10 = 2.f * dc * s10 + db * s00, d11 = 2.f * dc * s11 + db * s01, d12 = 2.f * dc * s12 + db * s02;
    const float dj00 = d00 * c.r00 + d01 * c.r01 + d02 * c.r02;
    const float dj02 = d00 * c.r20 + d01 * c.r21 + d02 * c.r22;
    const float dj11 = d10 * c.r10 + d11 * c.r11 + d12 * c.r12;
    const float dj12 = d10 * c.r20 + d11 * c.r21 + d12 * c.r22;
    e[0] += f.j00 * d00; e[1] += f.j00 * d01; e[2] += f.j00 * d02;
    e[4] += f.j11 * d10; e[5] += f.j11 * d11; e[6] += f.j11 * d12;
    e[8] += f.j02 * d00 + f.j12 * d10; e[9] += f.j02 * d01 + f.j12 * d11; e[10] += f.j02 * d02 + f.j12 * d12;
    const float z = f.z, iz = 1.0f / z, iz2 = iz * iz;
    gz = -c.fx * iz2 * dj00 - c.fy * iz2 * dj11;
    if (f.clamp_x) {
        gx = 0.f;
        gz += -f.j02 * iz * dj02;  // j02 = -fx*L/z  ->  d/dz = fx*L/z^2 = -j02/z
    } else {
        gx = -c.fx * iz2 * dj02;
        gz += 2.f * c.fx * px * iz2 * iz * dj02;
    }
    if (f.clamp_y) {
        gy = 0.f;
        gz += -f.j12 * iz * dj12;
    } else {
        gy = -c.fy * iz2 * dj12;
        gz += 2.f * c.fy * py * iz2 * iz * dj12;
    }
}

// ---------------------------------------------------------------- which pixel boxes can a splat reach?
// which of the tile's four 8x8 pixel blocks a splat reaches with alpha >= 1/255.  Exact up to a safety margin: the
// smallest q = A X^2 + 2 B X Y + C Y^2 over the block's box of pixel centres against 2 ln(255 o).  (Round 1 tested the
// bounding DISC of that ellipse, radius^2 = 2 ln(255 o) lambda_max: 16 % of the (splat, block) units it let through
// had no visible pixel at all, tools/lane_efficiency.py.)  q is convex with its minimum at the splat centre, so its
// minimum over a box that does not contain the centre lies on one of the (at most two) faces that look at the centre:
// X = clamp(0) with the best Y, or Y = clamp(0) with the best X.
struct BlockTest {
    float u, v, A, B2, C, bA, bC, tau;        // B2 = 2 B, bA = B / A, bC = B / C
};
__device__ __forceinline__ BlockTest block_test(const float4& p0, const float4& p1, float cutoff) {
    BlockTest t;
    t.u = gridf(p0.x); t.v = gridf(p0.y); t.A = p0.z; t.B2 = 2.f * p0.w; t.C = p1.x;
    t.bA = p0.w * __builtin_amdgcn_rcpf(p0.z);
    t.bC = p0.w * __builtin_amdgcn_rcpf(p1.x);
    const float r = 255.f * p1.y;
    // cutoff < 0: never visible; r < 1.05: too close to the threshold, no culling (as alpha_cutoff)
    t.tau = cutoff < 0.f ? -1.f : (r < 1.05f ? 3.0e38f : fmaf(2.004f, __logf(r), 1e-3f));
    return t;
}
__device__ __forceinline__ bool box_hit(const BlockTest& t, float x_lo, float x_hi, float y_lo, float y_hi) {
    const float x0 = x_lo - t.u, x1 = x_hi - t.u, y0 = y_lo - t.v, y1 = y_hi - t.v;      // the box about the centre
    const float Xc = __builtin_amdgcn_fmed3f(0.f, x0, x1), Yc = __builtin_amdgcn_fmed3f(0.f, y0, y1);
    const float Ys = __builtin_amdgcn_fmed3f(-t.bC * Xc, y0, y1);
    const float Xs = __builtin_amdgcn_fmed3f(-t.bA * Yc, x0, x1);
    const float a1 = t.A * Xc * Xc, c1 = t.C * Ys * Ys, a2 = t.A * Xs * Xs, c2 = t.C * Yc * Yc;
    // rounding of the three terms (|2 B X Y| <= A X^2 + C Y^2 for a positive definite conic): 1e-6 of their size
    const float q1 = fmaf(t.B2 * Xc, Ys, a1 + c1) - 1e-6f * (a1 + c1);
    const float q2 = fmaf(t.B2 * Xs, Yc, a2 + c2) - 1e-6f * (a2 + c2);
    return fminf(q1, q2) <= t.tau;
}
// the four bs x bs boxes at (px0, py0): bs = 8, the blocks of a tile; bs = 4, the quarters of a block
__device__ __forceinline__ unsigned block_mask(const float4& p0, const float4& p1, float cutoff, int px0, int py0, int bs = 8) {
    const BlockTest t = block_test(p0, p1, cutoff);
    unsigned m = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float x_lo = (float)(px0 + (w & 1) * bs), y_lo = (float)(py0 + (w >> 1) * bs);
        if (box_hit(t, x_lo, x_lo + (float)(bs - 1), y_lo, y_lo + (float)(bs - 1))) m |= 1u << w;
    }
    return m;
}


}  // namespace gfl
