// Stand-alone per-splat operators: project_point, compute_cov3d, ewa_project
// (forward + backward).  One lane per splat; each attribute array is a dense run of
// float3/float4 records, so a wave reads 64 consecutive records = one contiguous
// 768/1024-byte span per load instruction group (HBM-coalesced).
#include "gfl_math.hpp"

namespace gfl {

thread_local int g_last_hip_error = 0;

// ---------------------------------------------------------------- project_point
__global__ void __launch_bounds__(256) project_fwd_kernel(const float* __restrict__ xyz,
                                                          const float* __restrict__ intr,
                                                          const float* __restrict__ extr, int N, int W, int H,
                                                          float nearest, float extent, float* __restrict__ uv,
                                                          float* __restrict__ depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const Cam c = load_cam(intr, extr);
    const Proj p = project_fwd(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], W, H, nearest, extent);
    float2 o = make_float2(p.vis ? p.u : 0.f, p.vis ? p.v : 0.f);
    reinterpret_cast<float2*>(uv)[i] = o;
    depth[i] = p.vis ? p.pz : 0.f;
}

__global__ void __launch_bounds__(REDUCE_BLOCK) project_bwd_kernel(
    const float* __restrict__ xyz, const float* __restrict__ intr, const float* __restrict__ extr,
    const float* __restrict__ depth, const float* __restrict__ d_uv, const float* __restrict__ d_depth, int N,
    float* __restrict__ d_xyz, float* __restrict__ partial) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float e[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) e[k] = 0.f;
    if (i < N) {
        const Cam c = load_cam(intr, extr);
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (depth[i] != 0.f) {
            const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            const float px = c.r00 * x + c.r01 * y + c.r02 * z + c.t0;
            const float py = c.r10 * x + c.r11 * y + c.r12 * z + c.t1;
            const float pz = c.r20 * x + c.r21 * y + c.r22 * z + c.t2;
            float gx, gy, gz;
            project_bwd_cam(c, px, py, pz, d_uv[2 * i], d_uv[2 * i + 1], d_depth[i], gx, gy, gz);
            cam_grad_to_world(c, x, y, z, gx, gy, gz, dx, dy, dz, e);
        }
        d_xyz[3 * i] = dx; d_xyz[3 * i + 1] = dy; d_xyz[3 * i + 2] = dz;
    }
    block_reduce_store<12, REDUCE_BLOCK>(e, partial);
}

// --------------------------------------------------------------------- cov3d
__global__ void __launch_bounds__(256) cov3d_fwd_kernel(const float* __restrict__ scale,
                                                        const float* __restrict__ rotate,
                                                        const uint8_t* __restrict__ visible, int N,
                                                        float* __restrict__ cov3d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (visible[i]) {
        const float s[3] = {scale[3 * i], scale[3 * i + 1], scale[3 * i + 2]};
        const float4 qq = reinterpret_cast<const float4*>(rotate)[i];
        const float q[4] = {qq.x, qq.y, qq.z, qq.w};
        cov3d_fwd(s, q, cov);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cov3d[6 * i + k] = cov[k];
}

__global__ void __launch_bounds__(256) cov3d_bwd_kernel(const float* __restrict__ scale,
                                                        const float* __restrict__ rotate,
                                                        const uint8_t* __restrict__ visible,
                                                        const float* __restrict__ d_cov3d, int N,
                                                        float* __restrict__ d_scale, float* __restrict__ d_rotate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    if (visible[i]) {
        const float s[3] = {scale[3 * i], scale[3 * i + 1], scale[3 * i + 2]};
        const float4 qq = reinterpret_cast<const float4*>(rotate)[i];
        const float q[4] = {qq.x, qq.y, qq.z, qq.w};
        float g[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = d_cov3d[6 * i + k];
        cov3d_bwd(s, q, g, ds, dq);
    }
    d_scale[3 * i] = ds[0]; d_scale[3 * i + 1] = ds[1]; d_scale[3 * i + 2] = ds[2];
    reinterpret_cast<float4*>(d_rotate)[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
}

// ------------------------------------------------------------------------ EWA
__global__ void __launch_bounds__(256) ewa_fwd_kernel(const float* __restrict__ xyz, const float* __restrict__ cov3d,
                                                      const float* __restrict__ intr,
                                                      const float* __restrict__ extr, const float* __restrict__ uv,
                                                      const uint8_t* __restrict__ visible, int N, int W, int H,
                                                      float* __restrict__ conic, int32_t* __restrict__ radius,
                                                      int32_t* __restrict__ tiles) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float A = 0.f, B = 0.f, C = 0.f;
    int rad = 0, nt = 0;
    if (visible[i]) {
        const Cam c = load_cam(intr, extr);
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const float px = c.r00 * x + c.r01 * y + c.r02 * z + c.t0;
        const float py = c.r10 * x + c.r11 * y + c.r12 * z + c.t1;
        const float pz = c.r20 * x + c.r21 * y + c.r22 * z + c.t2;
        float cov[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) cov[k] = cov3d[6 * i + k];
        const Ewa e = ewa_fwd(c, px, py, pz, cov, W, H);
        if (e.ok) {
            const int r = ewa_radius(e);
            const int gx = (W + GFL_TILE - 1) / GFL_TILE, gy = (H + GFL_TILE - 1) / GFL_TILE;
            int x0, x1, y0, y1;
            tile_rect(uv[2 * i], uv[2 * i + 1], r, gx, gy, x0, x1, y0, y1);
            nt = (x1 - x0) * (y1 - y0);
            if (nt > 0) {
                rad = r;
                A = e.c / e.det; B = -e.b / e.det; C = e.a / e.det;
            }
        }
    }
    conic[3 * i] = A; conic[3 * i + 1] = B; conic[3 * i + 2] = C;
    radius[i] = rad;
    tiles[i] = nt;
}

__global__ void __launch_bounds__(REDUCE_BLOCK) ewa_bwd_kernel(
    const float* __restrict__ xyz, const float* __restrict__ cov3d, const float* __restrict__ intr,
    const float* __restrict__ extr, const int32_t* __restrict__ radius, const float* __restrict__ d_conic, int N,
    int W, int H, float* __restrict__ d_xyz, float* __restrict__ d_cov3d, float* __restrict__ partial) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float e[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) e[k] = 0.f;
    if (i < N) {
        float dx = 0.f, dy = 0.f, dz = 0.f;
        float gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (radius[i] > 0) {
            const Cam c = load_cam(intr, extr);
            const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            const float px = c.r00 * x + c.r01 * y + c.r02 * z + c.t0;
            const float py = c.r10 * x + c.r11 * y + c.r12 * z + c.t1;
            const float pz = c.r20 * x + c.r21 * y + c.r22 * z + c.t2;
            float cov[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) cov[k] = cov3d[6 * i + k];
            const Ewa f = ewa_fwd(c, px, py, pz, cov, W, H);
            float gx, gy, gz;
            ewa_bwd(c, f, px, py, cov, d_conic[3 * i], d_conic[3 * i + 1], d_conic[3 * i + 2], gcov, gx, gy, gz, e);
            cam_grad_to_world(c, x, y, z, gx, gy, gz, dx, dy, dz, e);
        }
        d_xyz[3 * i] = dx; d_xyz[3 * i + 1] = dy; d_xyz[3 * i + 2] = dz;
#pragma unroll
        for (int k = 0; k < 6; ++k) d_cov3d[6 * i + k] = gcov[k];
    }
    block_reduce_store<12, REDUCE_BLOCK>(e, partial);
}

}  // namespace gfl

using namespace gfl;

extern "C" {

int gfl_version(void) { return GFL_VERSION; }

int gfl_constants(float* out10) {
    if (!out10) return GFL_ERR_INVALID;
    const float c[10] = {(float)GFL_TILE, GFL_NEAREST, GFL_EXTENT, GFL_FOV_CLAMP, GFL_LOWPASS, GFL_EIG_FLOOR,
                         GFL_RADIUS_SIGMA, GFL_ALPHA_MIN, GFL_ALPHA_MAX, GFL_T_MIN};
    for (int k = 0; k < 10; ++k) out10[k] = c[k];
    return GFL_OK;
}

int gfl_constants_n(float* out, int n) {
    if (!out || n < 0) return GFL_ERR_INVALID;
    const float c[GFL_N_CONSTANTS] = {(float)GFL_TILE, GFL_NEAREST, GFL_EXTENT, GFL_FOV_CLAMP, GFL_LOWPASS, GFL_EIG_FLOOR,
                                      GFL_RADIUS_SIGMA, GFL_ALPHA_MIN, GFL_ALPHA_MAX, GFL_T_MIN, GFL_PIXEL_CENTER};
    for (int k = 0; k < n && k < GFL_N_CONSTANTS; ++k) out[k] = c[k];
    return GFL_N_CONSTANTS;
}

const char* gfl_status_string(int status) {
    switch (status) {
        case GFL_OK: return "ok";
        case GFL_ERR_INVALID: return "invalid argument";
        case GFL_ERR_WORKSPACE: return "workspace too small";
        case GFL_ERR_HIP: return "HIP error";
        default: return "unknown status";
    }
}

int gfl_last_hip_error(void) { return g_last_hip_error; }

size_t gfl_reduce_workspace_bytes(int N) { return (size_t)(reduce_rows(N > 0 ? N : 1)) * 12 * sizeof(float); }

int gfl_project_point_fwd(const float* xyz, const float* intr, const float* extr, int N, int W, int H,
                          float nearest, float extent, float* uv, float* depth, gfl_stream_t stream) {
    if (N < 0 || W <= 0 || H <= 0 || !intr || !extr) return GFL_ERR_INVALID;
    if (N == 0) return GFL_OK;
    if (!xyz || !uv || !depth) return GFL_ERR_INVALID;
    project_fwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(xyz, intr, extr, N, W, H, nearest, extent, uv,
                                                                         depth);
    return check_launch();
}

int gfl_project_point_bwd(const float* xyz, const float* intr, const float* extr, const float* depth,
                          const float* d_uv, const float* d_depth, int N, float* d_xyz, float* d_extr,
                          void* workspace, size_t workspace_bytes, gfl_stream_t stream) {
    if (N < 0 || !intr || !extr || !d_extr) return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) return check(hipMemsetAsync(d_extr, 0, 12 * sizeof(float), s));
    if (!xyz || !depth || !d_uv || !d_depth || !d_xyz || !workspace) return GFL_ERR_INVALID;
    if (workspace_bytes < gfl_reduce_workspace_bytes(N)) return GFL_ERR_WORKSPACE;
    const int rows = reduce_rows(N);
    project_bwd_kernel<<<rows, REDUCE_BLOCK, 0, s>>>(xyz, intr, extr, depth, d_uv, d_depth, N, d_xyz,
                                                     (float*)workspace);
    fold_partials_kernel<12><<<1, 256, 0, s>>>((const float*)workspace, rows, d_extr);
    return check_launch();
}

int gfl_cov3d_fwd(const float* scale, const float* rotate, const uint8_t* visible, int N, float* cov3d,
                  gfl_stream_t stream) {
    if (N < 0) return GFL_ERR_INVALID;
    if (N == 0) return GFL_OK;
    if (!scale || !rotate || !visible || !cov3d) return GFL_ERR_INVALID;
    cov3d_fwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(scale, rotate, visible, N, cov3d);
    return check_launch();
}

int gfl_cov3d_bwd(const float* scale, const float* rotate, const uint8_t* visible, const float* d_cov3d, int N,
                  float* d_scale, float* d_rotate, gfl_stream_t stream) {
    if (N < 0) return GFL_ERR_INVALID;
    if (N == 0) return GFL_OK;
    if (!scale || !rotate || !visible || !d_cov3d || !d_scale || !d_rotate) return GFL_ERR_INVALID;
    cov3d_bwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(scale, rotate, visible, d_cov3d, N, d_scale,
                                                                       d_rotate);
    return check_launch();
}

int gfl_ewa_fwd(const float* xyz, const float* cov3d, const float* intr, const float* extr, const float* uv,
                const uint8_t* visible, int N, int W, int H, float* conic, int32_t* radius, int32_t* tiles_touched,
                gfl_stream_t stream) {
    if (N < 0 || W <= 0 || H <= 0 || !intr || !extr) return GFL_ERR_INVALID;
    if (N == 0) return GFL_OK;
    if (!xyz || !cov3d || !uv || !visible || !conic || !radius || !tiles_touched) return GFL_ERR_INVALID;
    ewa_fwd_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(xyz, cov3d, intr, extr, uv, visible, N, W, H,
                                                                     conic, radius, tiles_touched);
    return check_launch();
}

int gfl_ewa_bwd(const float* xyz, const float* cov3d, const float* intr, const float* extr, const int32_t* radius,
                const float* d_conic, int N, int W, int H, float* d_xyz, float* d_cov3d, float* d_extr,
                void* workspace, size_t workspace_bytes, gfl_stream_t stream) {
    if (N < 0 || W <= 0 || H <= 0 || !intr || !extr || !d_extr) return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) return check(hipMemsetAsync(d_extr, 0, 12 * sizeof(float), s));
    if (!xyz || !cov3d || !radius || !d_conic || !d_xyz || !d_cov3d || !workspace) return GFL_ERR_INVALID;
    if (workspace_bytes < gfl_reduce_workspace_bytes(N)) return GFL_ERR_WORKSPACE;
    const int rows = reduce_rows(N);
    ewa_bwd_kernel<<<rows, REDUCE_BLOCK, 0, s>>>(xyz, cov3d, intr, extr, radius, d_conic, N, W, H, d_xyz, d_cov3d,
                                                 (float*)workspace);
    fold_partials_kernel<12><<<1, 256, 0, s>>>((const float*)workspace, rows, d_extr);
    return check_launch();
}

}  // extern "C"
