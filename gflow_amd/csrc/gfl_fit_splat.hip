// Fused fit iteration, stage 5: the per-splat launch -- gather of the pair rows, chain rule to the 14 raw parameters,
// regularisers, gradient masks, Adam on the 64-byte row -- with the scheduling workgroups of the NEXT iteration riding on it
// (tile queues of the two blend launches, reserved tile regions + sort order), and the camera / depth-affine step.
#include "gfl_fit_order.hpp"

namespace gfl {

// OP = true is the differentiable operator's backward (gfl_render_bwd): the rows hold ACTIVATED attributes, the
// camera is the extrinsic `pose` points at (12 floats), the caller's dL/d uv and dL/d depth join the gradient, and
// the 14 gradients are WRITTEN to d_params rows instead of stepping Adam (no regularisers, no masks).
// (Measured and moved out, tools/experiments/: the next iteration's preprocess in the tail of this launch -- neutral --, and the
// camera / depth-affine step as a ticketed tail of this launch -- slower.)
template <bool OP>
__global__ void __launch_bounds__(REDUCE_BLOCK) fused_preprocess_bwd_adam_kernel(
    float* __restrict__ params, float* __restrict__ adam_m, float* __restrict__ adam_v, const float* __restrict__ intr,
    const float* pose, const float* __restrict__ rec, float* __restrict__ d_rec,
    const float* __restrict__ pair_grad, const int32_t* __restrict__ wide_off, long long wide_base,
    const int32_t* __restrict__ stamp_ptr, int gx, int gy, int N, int W, int H,
    const float* __restrict__ flow_target, const float* __restrict__ flow_w, const float* __restrict__ still_target,
    const float* __restrict__ still_w, const uint8_t* __restrict__ row_flags, RegCfg rc, AdamCfg ac,
    const int32_t* d_step, float* partial, const float* __restrict__ d_uv_in,
    const float* __restrict__ d_depth_in, float* __restrict__ d_params, const int32_t* __restrict__ scale_cnt,
    NextSched ns, const int32_t* overflow) {
    constexpr int BLOCK = REDUCE_BLOCK;
    extern __shared__ int32_t sched_scratch[];           // T ints + the block plans: the scheduling workgroups' scratch
    if ((int)blockIdx.x >= ns.rows) {
        __shared__ int32_t sched_wsum[BLOCK / 64];
        GFL_PHASE(3, 0);
        if ((int)blockIdx.x == ns.rows + 2) {
            // ... and a third the next iteration's tile regions, its sort order, and what the column scan of the exact path
            // resets (the slot pool's counter, the blend launches' pull counters: both done with for this iteration)
            build_sort_order<BLOCK, true>(ns.tile_counts, ns.T, ns.order_next, sched_wsum, ns.ro);
            for (int c = threadIdx.x; c < ns.n_pull; c += BLOCK) ns.pull_counters[c] = 0;
            if (threadIdx.x == 0) {
                *ns.pool_counter = 0;
                *ns.regions_valid = 1;
            }
            GFL_PHASE(3, 7);
            return;
        }
        // the two workgroups behind the per-splat ones build the NEXT iteration's tile queues (see fused_scatter_kernel)
        __shared__ SchedLds sched_lds;
        const Sched sc = (int)blockIdx.x == ns.rows ? ns.bwd : ns.fwd;
        schedule_tiles_xcd<BLOCK>(ns.tile_counts, ns.T, sc, sched_scratch, sched_wsum, sched_lds,
                                  reinterpret_cast<uint32_t*>(sched_scratch + ns.T));      // (T <= SCHED_PLAN_TILES: next_sched_ok)
        if (threadIdx.x == 0) *ns.valid = 1;
        GFL_PHASE(3, 7);
        return;
    }
    GFL_PHASE(3, 0);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // the forward dropped pairs (the lists overflowed, [0], or a tile outgrew its reserved region, [2]): no row is stepped
    const bool dropped = !OP && overflow != nullptr && (overflow[0] | overflow[2]) != 0;
    const int e_step = OP ? 0 : *d_step - ((rc.no_pose_grad && !dropped) ? 1 : 0);   // (the camera launch advances it -- or already has: LossTail)
    float scale_w = 0.f;                              // lambda_scale / rows of the scale term
    if (!OP && rc.lambda_scale != 0.f) {
        __shared__ int32_t s_rows;
        if (threadIdx.x == 0) s_rows = 0;
        __syncthreads();
        int c = 0;
        for (int b = threadIdx.x; b < rc.scale_blocks; b += BLOCK) c += scale_cnt[b];
        if (c) atomicAdd(&s_rows, c);
        __syncthreads();
        scale_w = s_rows > 0 ? rc.lambda_scale / (float)s_rows : 0.f;
    }
    float e[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) e[k] = 0.f;
    // gather this splat's rows of pair_grad (one per tile it was binned into), in tile order.
    // The tile sort left each pair's list position in the splat's slot row; splats covering more
    // than SLOT_MAX tiles (a handful per frame) are handled by the whole wave below.
    float4 rp0 = make_float4(0.f, 0.f, 0.f, 0.f), rp2 = rp0;
    float4 d0 = rp0, d1 = rp0, d2 = rp0;   // gathered: s0 s1 s2 s3 | s4 do dr dg | db ddepth (moments, see below)
    float4 d0g = rp0, d1g = rp0, d2g = rp0;
    bool big = false;
    int big_nt = 0;
    // the parameter row and both Adam moments are requested before the gather so that their
    // latency overlaps it (the launch has about one wave per SIMD: nothing else would hide it)
    float4 prow_v[4] = {}, mrow_v[4], vrow_v[4];
    float4 f0[4], f1[4], f2[4];
    const int stamp = *stamp_ptr;
    float rec_C = 0.f;
    unsigned own_flags = 0;
    float own_flow_w = 0.f, own_still_w = 0.f, own_still_t[3] = {0.f, 0.f, 0.f};    // (OP: own_flow_w = dL/d depth,
    float2 own_flow_t = make_float2(0.f, 0.f);                                      //  own_flow_t = dL/d uv of the caller)
    if (i < N) {
        const float4* prow = reinterpret_cast<const float4*>(params + (size_t)i * ROW);
        const float4* mrow = reinterpret_cast<const float4*>(adam_m + (size_t)i * ROW);
        const float4* vrow = reinterpret_cast<const float4*>(adam_v + (size_t)i * ROW);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            prow_v[q] = prow[q];
            // camera-only stage (freeze_all): every gradient is zeroed and the moments were reset at the start of the
            // stage, so Adam leaves row, m and v exactly as they are -- they are neither read nor written (2/3 of this
            // launch's traffic, in a third of a clip's iterations)
            if (!OP && !rc.freeze_all) { mrow_v[q] = mrow[q]; vrow_v[q] = vrow[q]; }
        }
        const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)i * REC);
        rp0 = r4[0]; rp2 = r4[2];
        rec_C = rec[(size_t)i * REC + 4];
        {
            // the splat's first four pair rows, before anybody knows how many it has (their place depends on i alone)
            const float4* g4 = reinterpret_cast<const float4*>(pair_grad + (size_t)i * SLOT_MAX * PG);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f0[j] = g4[3 * j]; f1[j] = g4[3 * j + 1]; f2[j] = g4[3 * j + 2]; }
        }
        // the per-row side inputs of the regularisers too: read where they are used, deep inside the chain rule, each
        // was a round trip of its own (joint stages: chain rule 3.5 us against 2.3 without them, tools/phase_trace.py)
        if (!OP) {
            if (row_flags) own_flags = row_flags[i];
            if (flow_w) {
                own_flow_w = flow_w[i];
                own_flow_t = reinterpret_cast<const float2*>(flow_target)[i];
            }
            if (still_w) {
                own_still_w = still_w[i];
                own_still_t[0] = still_target[3 * i]; own_still_t[1] = still_target[3 * i + 1];
                own_still_t[2] = still_target[3 * i + 2];
            }
        } else {
            if (d_uv_in) own_flow_t = reinterpret_cast<const float2*>(d_uv_in)[i];
            if (d_depth_in) own_flow_w = d_depth_in[i];
        }
#ifdef GFL_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GFL_PHASE(3, 1);
#endif
        {
            const int rad = __float_as_int(rp2.w);
            if (rad > 0) {
                int x0, x1, y0, y1;
                tile_rect(rp0.x, rp0.y, rad, gx, gy, x0, x1, y0, y1);
                const int nt = (x1 - x0) * (y1 - y0);
                if (nt <= SLOT_MAX) {
                    // The splat's pair rows are rows i * SLOT_MAX .. + nt of pair_grad, in the order of its tile rectangle
                    // (the backward blend put them there); the first four were requested at the top with everything else --
                    // nine splats in ten have no more --, the others follow four at a time (twelve loads in flight), summed in
                    // tile order.  A row counts if it carries this forward's stamp.
                    const float4* g4 = reinterpret_cast<const float4*>(pair_grad + (size_t)i * SLOT_MAX * PG);
#pragma unroll
                    for (int k = 0; k < SLOT_MAX / 4; ++k) {
                        if (4 * k >= nt) break;
                        float4 r0[4], r1[4], r2[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (k == 0) { r0[j] = f0[j]; r1[j] = f1[j]; r2[j] = f2[j]; }
                            else { r0[j] = g4[3 * (4 * k + j)]; r1[j] = g4[3 * (4 * k + j) + 1]; r2[j] = g4[3 * (4 * k + j) + 2]; }
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (4 * k + j < nt && __float_as_int(r2[j].z) == stamp) {
                                d0.x += r0[j].x; d0.y += r0[j].y; d0.z += r0[j].z; d0.w += r0[j].w;
                                d1.x += r1[j].x; d1.y += r1[j].y; d1.z += r1[j].z; d1.w += r1[j].w;
                                d2.x += r2[j].x; d2.y += r2[j].y;
                            }
                        }
                    }
                } else {
                    big = true;
                    big_nt = nt;
                }
            }
        }
    }
#ifdef GFL_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GFL_PHASE(3, 2);
#endif
    // ---- wave-cooperative gather for the few splats with more than SLOT_MAX tiles: their rows are a run behind the others
    // (wide_off), wave-summed
    {
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(big);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int si = __shfl(i, src);
            const int nt = __shfl(big_nt, src);
            const int off = wide_off[si];
            float a[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) a[k] = 0.f;
            for (int q = lane; q < nt && off >= 0; q += 64) {
                const float4* g4 = reinterpret_cast<const float4*>(pair_grad + (size_t)(wide_base + off + q) * PG);
                const float4 q0 = g4[0], q1 = g4[1], q2 = g4[2];
                if (__float_as_int(q2.z) == stamp) {
                    a[0] += q0.x; a[1] += q0.y; a[2] += q0.z; a[3] += q0.w; a[4] += q1.x; a[5] += q1.y; a[6] += q1.z;
                    a[7] += q1.w; a[8] += q2.x; a[9] += q2.y;
                }
            }
#pragma unroll
            for (int k = 0; k < 10; ++k) a[k] = wave_sum(a[k]);
            if (lane == src) {
                d0g = make_float4(a[0], a[1], a[2], a[3]);
                d1g = make_float4(a[4], a[5], a[6], a[7]);
                d2g = make_float4(a[8], a[9], 0.f, 0.f);
            }
        }
    }
    GFL_PHASE(3, 3);
    if (i < N) {
        const Cam c = OP ? load_cam(intr, pose) : cam_from_pose(intr, pose);
#ifdef GFL_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GFL_PHASE(3, 4);
#endif
        const Splat s = splat_from_row(prow_v[0], prow_v[1], prow_v[2], prow_v[3], OP);
        if (big) { d0 = d0g; d1 = d1g; d2 = d2g; }
        {
            // moments of the backward blend -> du dv dA dB dC (blend_bwd_terms)
            const float A = rp0.z, B = rp0.w, C = rec_C;
            const float s0 = d0.x, s1 = d0.y;
            d0.x = fmaf(A, s0, B * s1);
            d0.y = fmaf(C, s1, B * s0);
            d0.z *= 0.5f;
            d1.x *= 0.5f;
        }
        if (d_rec) {                                        // (dL/d rec: an output nobody in the fit reads -- written when asked for)
            float4* o4 = reinterpret_cast<float4*>(d_rec + (size_t)i * REC);
            o4[0] = d0; o4[1] = d1; o4[2] = d2;
        }
        float g[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) g[k] = 0.f;
        float scale_g = 0.f;                                // d(scale term) / d|s_k| = scale_g * |s_k|
        const bool vis = rp2.y != 0.f;                      // depth != 0  (render.py:29)
        if (vis) {
            float du = d0.x, dv = d0.y, dd = d2.y;
            if (OP) {                                       // the caller's own use of uv / depth (flow, scale losses)
                if (d_uv_in) { du += own_flow_t.x; dv += own_flow_t.y; }
                if (d_depth_in) dd += own_flow_w;
            }
            if (!OP && scale_w != 0.f && scale_row(rp0.x, rp0.y, W, H, own_flags, rc.freeze_all ? 2 : 1)) {
                // mean over the rows of |scale| / depth (trainer.py:495-502): d/d depth here, d/d scale below
                const float nrm = sqrtf(s.s[0] * s.s[0] + s.s[1] * s.s[1] + s.s[2] * s.s[2]);
                dd -= scale_w * nrm / (rp2.y * rp2.y);
                scale_g = nrm > 0.f ? scale_w / (nrm * rp2.y) : 0.f;
            }
            if (!OP && flow_w) {                            // flow term acts on uv (trainer.py:520-528)
                const float w = rc.lambda_flow * own_flow_w;
                if (w != 0.f) {
                    du += 2.f * w * (rp0.x - own_flow_t.x);
                    dv += 2.f * w * (rp0.y - own_flow_t.y);
                }
            }
            const float px = c.r00 * s.x + c.r01 * s.y + c.r02 * s.z + c.t0;
            const float py = c.r10 * s.x + c.r11 * s.y + c.r12 * s.z + c.t1;
            const float pz = c.r20 * s.x + c.r21 * s.y + c.r22 * s.z + c.t2;
            float gx_, gy_, gz_;
            project_bwd_cam(c, px, py, pz, du, dv, dd, gx_, gy_, gz_);
            if (__float_as_int(rp2.w) > 0) {                // radius > 0: the conic was produced
                float cov[6];
                cov3d_fwd(s.s, s.q, cov);
                const Ewa f = ewa_fwd(c, px, py, pz, cov, W, H);
                float gcov[6], ex, ey, ez;
                ewa_bwd(c, f, px, py, cov, d0.z, d0.w, d1.x, gcov, ex, ey, ez, e);
                gx_ += ex; gy_ += ey; gz_ += ez;
                // (camera-only stage: the pose gradient is complete with ewa_bwd and cam_grad_to_world; what follows --
                //  scale / rotation gradients, the blended attributes, the regularisers -- would be zeroed at the end)
                float ds[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
                if (OP || !rc.freeze_all) cov3d_bwd(s.s, s.q, gcov, ds, dq);
#pragma unroll
                for (int k = 0; k < 3; ++k) g[3 + k] = ds[k];
                if (OP) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[6 + k] = dq[k];
                } else {
                    // through F.normalize: q = raw / n
                    const float dot = s.q[0] * dq[0] + s.q[1] * dq[1] + s.q[2] * dq[2] + s.q[3] * dq[3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[6 + k] = (dq[k] - s.q[k] * dot) / s.qn;
                }
            }
            cam_grad_to_world(c, s.x, s.y, s.z, gx_, gy_, gz_, g[0], g[1], g[2], e);
        }
        if (OP) {
            // gradients wrt the activated attributes, as msplat's operators return them
            float4* o4 = reinterpret_cast<float4*>(d_params + (size_t)i * ROW);
            o4[0] = make_float4(g[0], g[1], g[2], g[3]);
            o4[1] = make_float4(g[4], g[5], g[6], g[7]);
            o4[2] = make_float4(g[8], g[9], d1.y, d1.z);
            o4[3] = make_float4(d1.w, d2.x, 0.f, 0.f);
        } else if (!rc.freeze_all) {
        // blended attributes: opacity = sigmoid(10 x), rgb = sigmoid(x)
        g[10] = d1.y * 10.f * s.o * (1.f - s.o);
        g[11] = d1.z * s.c[0] * (1.f - s.c[0]);
        g[12] = d1.w * s.c[1] * (1.f - s.c[1]);
        g[13] = d2.x * s.c[2] * (1.f - s.c[2]);
        // scale: the two regularisers on |x| (trainer.py:490-502), then the backward of |x|
#pragma unroll
        for (int k = 0; k < 3; ++k) g[3 + k] += scale_g * s.s[k];
        if (rc.lambda_var != 0.f) {
            const float mean = (s.s[0] + s.s[1] + s.s[2]) * (1.f / 3.f);
            const float var = 0.5f * ((s.s[0] - mean) * (s.s[0] - mean) + (s.s[1] - mean) * (s.s[1] - mean) +
                                      (s.s[2] - mean) * (s.s[2] - mean));
            const float sd = sqrtf(var);
            if (sd != 0.f) {                                // torch masks the 0/0 of std's backward to 0
#pragma unroll
                for (int k = 0; k < 3; ++k) g[3 + k] += rc.lambda_var * (s.s[k] - mean) / (2.f * sd);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) g[3 + k] *= (s.raw_s[k] > 0.f) ? 1.f : ((s.raw_s[k] < 0.f) ? -1.f : 0.f);
        if (still_w) {                                      // trainer.py:505-509
            const float w = rc.lambda_still * own_still_w;
            if (w != 0.f) {
                const float ax = s.x - own_still_t[0], ay = s.y - own_still_t[1], az = s.z - own_still_t[2];
                const float n = sqrtf(ax * ax + ay * ay + az * az);
                if (n != 0.f) { g[0] += w * ax / n; g[1] += w * ay / n; g[2] += w * az / n; }
            }
        }
        // gradient control (trainer.py:535-551)
        if (rc.freeze_rgb) { g[11] = 0.f; g[12] = 0.f; g[13] = 0.f; }
        if (own_flags & 1u) { g[0] = 0.f; g[1] = 0.f; g[2] = 0.f; }
        if (rc.freeze_all) {
#pragma unroll
            for (int k = 0; k < 14; ++k) g[k] = 0.f;
        }
        // Adam over the 64-byte row
        GFL_PHASE(3, 5);
        if (!rc.freeze_all && !dropped) {
        float step_size, isb2;
        adam_scalars(ac, e_step, ac.lr, step_size, isb2);
        float4* prow = reinterpret_cast<float4*>(params + (size_t)i * ROW);
        float4* mrow = reinterpret_cast<float4*>(adam_m + (size_t)i * ROW);
        float4* vrow = reinterpret_cast<float4*>(adam_v + (size_t)i * ROW);
        float pv[16], mv[16], vv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = prow_v[q], b = mrow_v[q], d = vrow_v[q];
            pv[4 * q] = a.x; pv[4 * q + 1] = a.y; pv[4 * q + 2] = a.z; pv[4 * q + 3] = a.w;
            mv[4 * q] = b.x; mv[4 * q + 1] = b.y; mv[4 * q + 2] = b.z; mv[4 * q + 3] = b.w;
            vv[4 * q] = d.x; vv[4 * q + 1] = d.y; vv[4 * q + 2] = d.z; vv[4 * q + 3] = d.w;
        }
#pragma unroll
        for (int k = 0; k < 14; ++k) pv[k] = adam_update(pv[k], g[k], mv[k], vv[k], ac, step_size, isb2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            prow[q] = make_float4(pv[4 * q], pv[4 * q + 1], pv[4 * q + 2], pv[4 * q + 3]);
            mrow[q] = make_float4(mv[4 * q], mv[4 * q + 1], mv[4 * q + 2], mv[4 * q + 3]);
            vrow[q] = make_float4(vv[4 * q], vv[4 * q + 1], vv[4 * q + 2], vv[4 * q + 3]);
        }
        }
        }
    }
    GFL_PHASE(3, 6);
    if (!OP && rc.no_pose_grad) {
        // nobody reads the extrinsic partials of this iteration
    } else {
        block_reduce_store<12, BLOCK>(e, partial);
    }
    GFL_PHASE(3, 7);
}

// camera + depth affine: fold the extr partials, chain to the pose, Adam, step += 1
// One block of 1024 lanes also folds the loss partial rows (no separate fold launch): every lane
// is at most a couple of loads deep, the tree has a fixed shape (reproducible).
__global__ void __launch_bounds__(1024) fused_camera_adam_kernel(
    const float* __restrict__ partial, int rows, const float* __restrict__ p_ssim, int n_ssim,
    const float* __restrict__ p_grad, int n_grad, float* __restrict__ pose, float* __restrict__ pose_m,
    float* __restrict__ pose_v, float* __restrict__ depth_ab, float* __restrict__ ab_m, float* __restrict__ ab_v,
    float* __restrict__ sums, AdamCfg ac_cam, AdamCfg ac_ab, int step_camera, int32_t* __restrict__ d_step,
    float* __restrict__ d_extr_out, int32_t* __restrict__ overflow) {
    constexpr int NV = 17;   // 12 extr + {mse, ssim, depth, d/da, d/db}
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    for (int r = threadIdx.x; r < rows; r += 1024) {
        const float4* p4 = reinterpret_cast<const float4*>(partial + (size_t)r * 12);
        const float4 a = p4[0], b = p4[1], c = p4[2];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z;
        acc[7] += b.w; acc[8] += c.x; acc[9] += c.y; acc[10] += c.z; acc[11] += c.w;
    }
    // one workgroup, nothing to overlap a load with but other loads: issue them in batches
    // (one load per trip made this kernel a chain of ~8 dependent L2 round trips)
    for (int r0 = threadIdx.x; r0 < n_ssim; r0 += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (r0 + u * 1024 < n_ssim) ? p_ssim[r0 + u * 1024] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[13] += v[u];
    }
    for (int r0 = threadIdx.x; r0 < n_grad; r0 += 4 * 1024) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            q[u] = (r0 + u * 1024 < n_grad) ? reinterpret_cast<const float4*>(p_grad)[r0 + u * 1024] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[12] += q[u].x; acc[14] += q[u].y; acc[15] += q[u].z; acc[16] += q[u].w; }
    }
    // thread 0 needs these after the reduction: request them now
    float pz[7], pm[7], pv[7], ab[2], abm[2], abv[2];
    int e_step = 0;
    if (threadIdx.x == 0) {
        e_step = *d_step;
#pragma unroll
        for (int k = 0; k < 7; ++k) { pz[k] = pose[k]; pm[k] = pose_m[k]; pv[k] = pose_v[k]; }
#pragma unroll
        for (int k = 0; k < 2; ++k) { ab[k] = depth_ab[k]; abm[k] = ab_m[k]; abv[k] = ab_v[k]; }
    }
    __shared__ float red[16][NV];
    __shared__ float ge[NV];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float s = wave_sum_to_lane63(acc[k]);
        if (lane == 63) red[wid][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w][threadIdx.x];
        ge[threadIdx.x] = t;
        if (threadIdx.x < 12) d_extr_out[threadIdx.x] = t;
        else sums[threadIdx.x - 12] = t;       // sums[0..4] as gfl_loss_fwd_bwd documents
    }
    if (threadIdx.x >= NV && threadIdx.x < NV + 3) sums[threadIdx.x - NV + 5] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int e = e_step;
        if (overflow && (overflow[0] | overflow[2]) != 0) {      // the forward dropped pairs: nothing is stepped, the iteration is counted (LossTail)
            overflow[1] += 1;
            return;
        }
        if (step_camera) {
            // d_extr (rows R|t) -> d_pose; q = raw/|raw| in XYZW order
            const float rx = pz[0], ry = pz[1], rz = pz[2], rw = pz[3];
            const float n = sqrtf(rx * rx + ry * ry + rz * rz + rw * rw);
            const float x = rx / n, y = ry / n, z = rz / n, w = rw / n;
            const float* dR = ge;   // dR[i][j] = ge[4 i + j]
            const float d00 = dR[0], d01 = dR[1], d02 = dR[2], d10 = dR[4], d11 = dR[5], d12 = dR[6], d20 = dR[8],
                        d21 = dR[9], d22 = dR[10];
            float dq[4];   // x y z w
            dq[3] = 2.f * (-z * d01 + y * d02 + z * d10 - x * d12 - y * d20 + x * d21);
            dq[0] = 2.f * (y * d01 + z * d02 + y * d10 - 2.f * x * d11 - w * d12 + z * d20 + w * d21 - 2.f * x * d22);
            dq[1] = 2.f * (-2.f * y * d00 + x * d01 + w * d02 + x * d10 + z * d12 - w * d20 + z * d21 - 2.f * y * d22);
            dq[2] = 2.f * (-2.f * z * d00 - w * d01 + x * d02 + w * d10 - 2.f * z * d11 + y * d12 + x * d20 + y * d21);
            const float qh[4] = {x, y, z, w};
            const float dot = qh[0] * dq[0] + qh[1] * dq[1] + qh[2] * dq[2] + qh[3] * dq[3];
            float gp[7];
#pragma unroll
            for (int k = 0; k < 4; ++k) gp[k] = (dq[k] - qh[k] * dot) / n;
            gp[4] = ge[3]; gp[5] = ge[7]; gp[6] = ge[11];
            float ss, isb;
            adam_scalars(ac_cam, e, ac_cam.lr, ss, isb);
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                pose[k] = adam_update(pz[k], gp[k], pm[k], pv[k], ac_cam, ss, isb);
                pose_m[k] = pm[k]; pose_v[k] = pv[k];
            }
            adam_scalars(ac_ab, e, ac_ab.lr, ss, isb);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                depth_ab[k] = adam_update(ab[k], ge[15 + k], abm[k], abv[k], ac_ab, ss, isb);
                ab_m[k] = abm[k]; ab_v[k] = abv[k];
            }
        }
        *d_step = e + 1;
    }
}

// ---- launchers (gfl_fit.hpp)
// the fit iteration's per-splat launch: `extra` scheduling workgroups behind the `ns.rows` row workgroups, `lds` bytes for them
void launch_splat_bwd_adam(const gfl_fit_state* st, const FitWs& w, int gx, int gy, const RegCfg& rc, const AdamCfg& ac,
                           const NextSched& ns, int extra, size_t lds, hipStream_t s) {
    fused_preprocess_bwd_adam_kernel<false><<<ns.rows + extra, REDUCE_BLOCK, lds, s>>>(
        st->params, st->adam_m, st->adam_v, st->intr, st->pose, st->rec, st->d_rec, w.pair_grad, w.wide_off, w.wide_base, w.stamp,
        gx, gy, st->N, st->W, st->H, st->flow_target, st->flow_w, st->still_target, st->still_w, st->row_flags, rc, ac,
        st->step, w.partial, nullptr, nullptr, nullptr, w.scale_cnt, ns, st->overflow);
}

// the differentiable operator's backward (gfl_render_bwd): gradients wrt the activated attributes into d_params, d_extr folded
void launch_splat_bwd_op(const gfl_fit_state* st, const FitWs& w, int gx, int gy, const NextSched& ns, int extra, size_t lds,
                         const float* d_uv, const float* d_depth, float* d_params, float* d_extr, hipStream_t s) {
    fused_preprocess_bwd_adam_kernel<true><<<ns.rows + extra, REDUCE_BLOCK, lds, s>>>(
        st->params, nullptr, nullptr, st->intr, st->extr, st->rec, st->d_rec, w.pair_grad, w.wide_off, w.wide_base, w.stamp,
        gx, gy, st->N, st->W, st->H, nullptr, nullptr, nullptr, nullptr, nullptr, RegCfg{}, AdamCfg{}, nullptr, w.partial, d_uv,
        d_depth, d_params, nullptr, ns, nullptr);
    fold_partials_kernel<12><<<1, 256, 0, s>>>(w.partial, ns.rows, d_extr);
}

void launch_camera_adam(const gfl_fit_state* st, const FitWs& w, int rows, const float* p_ssim, int n_ssim, const float* p_grad,
                        int n_grad, const AdamCfg& ac_cam, const AdamCfg& ac_ab, int step_camera, hipStream_t s) {
    fused_camera_adam_kernel<<<1, 1024, 0, s>>>(w.partial, rows, p_ssim, n_ssim, p_grad, n_grad, st->pose, st->pose_m, st->pose_v,
                                                st->depth_ab, st->depth_ab_m, st->depth_ab_v, st->sums, ac_cam, ac_ab, step_camera,
                                                st->step, st->d_extr, st->overflow);
}

#ifdef GFL_TRACE
int read_phase_trace_splat(long long* out, int n_values) {   // row 3 (per-splat backward + Adam), at its place in the table
    const int lo = 3 * PHASE_WAVES * 8, hi = 4 * PHASE_WAVES * 8;
    if (n_values <= lo) return 0;
    const int n = (n_values < hi ? n_values : hi) - lo;
    return (int)hipMemcpyFromSymbol(out + lo, HIP_SYMBOL(g_phase_trace), (size_t)n * sizeof(long long), (size_t)lo * sizeof(long long));
}
#endif

}  // namespace gfl
