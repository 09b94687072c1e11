// Fused fit iteration, stage 4: the backward of the compositing -- back to front over the sorted per-tile lists, one
// 48-byte row of raw moments per (splat, tile) pair, no global atomics -- and, in iterations that cannot move the camera,
// the fold of the loss partials + the depth-affine step in one of its workgroups (LossTail, gfl_fit.hpp).
#include "gfl_fit.hpp"

namespace gfl {

// gradient terms of one (pixel, splat) pair; branch-free: a lane that does not see the splat uses
// alpha = 0 (T, S unchanged, every term exactly 0).  v = s0..s4 (moments, below) do df0..3
__device__ __forceinline__ void blend_bwd_terms(const float4& p0, const float4& p1, const float4& p2, float fx, float fy,
                                                bool valid, float alpha, float G, float g0, float g1, float g2, float g3,
                                                float& T, float& S, float (&v)[10]) {
    const float a_eff = valid ? alpha : 0.f;
    const float rom = __builtin_amdgcn_rcpf(1.f - a_eff);
    T = T * rom;
    const float h = fmaf(g0, p1.z, fmaf(g1, p1.w, fmaf(g2, p2.x, g3 * p2.y)));
    const float dalpha = valid ? fmaf(T, h, -(S * rom)) : 0.f;
    const float w = a_eff * T;
    S = fmaf(h, w, S);
    v[6] = w * g0; v[7] = w * g1; v[8] = w * g2; v[9] = w * g3;
    const float dx = p0.x - fx, dy = p0.y - fy;
    v[5] = G * dalpha;
    const float dpow = p1.y * v[5];
    const float mx = -dx * dpow, my = -dy * dpow;
    // raw moments; the conic (A, B, C) is the same for every pixel of the splat, so the per-splat
    // kernel finishes them after the sums: du = A s0 + B s1, dv = C s1 + B s0, dA = s2 / 2,
    // dB = s3, dC = s4 / 2  (six VALU ops fewer per (splat, 8x8 block) unit than forming them here)
    v[0] = mx;
    v[1] = my;
    v[2] = dx * mx;
    v[3] = dx * my;
    v[4] = dy * my;
}

#ifdef GFL_TRACE
// analysis build only (make TRACE=1): per-tile timeline of the backward blend
__device__ long long g_bwd_trace[16384 * 8];
#endif
template <int B = 256>
__device__ void loss_tail(const LossTail& t) {
    constexpr int NV = 5;                   // mse, ssim, depth, d/da, d/db
    __shared__ float red[B / 64][NV];
    __shared__ float ge[NV];
    float ab[2] = {0.f, 0.f}, abm[2] = {0.f, 0.f}, abv[2] = {0.f, 0.f};
    int e_step = 0;
    if (threadIdx.x == 0) {                 // thread 0 needs these after the reduction: request them now
        e_step = *t.d_step;
#pragma unroll
        for (int k = 0; k < 2; ++k) { ab[k] = t.depth_ab[k]; abm[k] = t.ab_m[k]; abv[k] = t.ab_v[k]; }
    }
    float acc[NV] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r0 = threadIdx.x; r0 < t.n_ssim; r0 += 8 * B) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (r0 + u * B < t.n_ssim) ? t.p_ssim[r0 + u * B] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[1] += v[u];
    }
    for (int r0 = threadIdx.x; r0 < t.n_grad; r0 += 4 * B) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            q[u] = (r0 + u * B < t.n_grad) ? reinterpret_cast<const float4*>(t.p_grad)[r0 + u * B]
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[0] += q[u].x; acc[2] += q[u].y; acc[3] += q[u].z; acc[4] += q[u].w; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float s = wave_sum_to_lane63(acc[k]);
        if (lane == 63) red[wid][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < B / 64; ++w) x += red[w][threadIdx.x];
        ge[threadIdx.x] = x;
        t.sums[threadIdx.x] = x;            // sums[0..4] as gfl_loss_fwd_bwd documents
    }
    if (threadIdx.x >= NV && threadIdx.x < NV + 3) t.sums[threadIdx.x] = 0.f;
    if (threadIdx.x >= 16 && threadIdx.x < 28) t.d_extr_out[threadIdx.x - 16] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (t.overflow && (t.overflow[0] | t.overflow[2]) != 0) {
            // The forward of this iteration dropped (splat, tile) pairs: its gradients are not the scene's.  Nothing is
            // stepped -- the per-splat launch skips its rows the same way --, the step counter stays, and the iteration is
            // counted so that the host can run it again once it has grown the lists (FitEngine.settle_overflow).
            t.overflow[1] += 1;
            return;
        }
        if (t.step_affine) {
            float ss, isb;
            adam_scalars(t.ac_ab, e_step, t.ac_ab.lr, ss, isb);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                t.depth_ab[k] = adam_update(ab[k], ge[3 + k], abm[k], abv[k], t.ac_ab, ss, isb);
                t.ab_m[k] = abm[k]; t.ab_v[k] = abv[k];
            }
        }
        *t.d_step = e_step + 1;
    }
}

// SUMS: how many of the ten per-pair sums somebody reads.  10: the first frame.  7: later frames, whose colours are frozen
// (freeze_rgb, trainer.py:537-540) -- the three colour sums are neither formed nor reduced (18 VALU ops in the wave
// reduce-scatter instead of 27 per unit).  6: the camera-only stage (freeze_all_splats) -- every splat gradient is zeroed
// afterwards, and the pose gradient needs only the five moments and the depth feature's gradient (16 ops).
template <int SUMS>
__global__ void __launch_bounds__(256, 8) fused_blend_bwd_kernel(const float* __restrict__ rec,
                                                              const int32_t* __restrict__ ids,
                                                              const int32_t* __restrict__ tile_range, float bg, int W,
                                                              int H, int gx, const float* __restrict__ final_T,
                                                              const int32_t* __restrict__ n_contrib,
                                                              const float* __restrict__ d_out,
                                                              float* __restrict__ pair_grad, TileQueue queue,
                                                              int32_t* __restrict__ tile_work,
                                                              const float* __restrict__ ckpt,
                                                              const float* __restrict__ render, LossTail ltail, int gy,
                                                              const int32_t* __restrict__ wide_off, long long wide_base,
                                                              const int32_t* __restrict__ stamp_ptr) {
    // (see LossTail.  The workgroup that holds the LAST pre-assigned slot of queue 0 does it before its first item: ~3 us
    //  that the seven other workgroups of its queue absorb.  A workgroup of its own behind the others only started when one
    //  of them -- all persistent -- had finished: +1 us at the end of the launch.)
    if (ltail.enabled && blockIdx.x == gridDim.x - queue.nq) loss_tail(ltail);
    // No global atomics: the four waves of the tile combine their per-splat sums in LDS and the tile writes ONE 48-byte row per
    // (splat, tile) pair with plain stores -- round 5: not at the pair's list position (the per-splat launch then needed a table
    // of positions, written by the tile sort from a gather of the records, and gathered rows scattered over 12.7 MB: 41 MB of
    // traffic, docs/history.md section 5) but at row g * SLOT_MAX + (the tile's index in the splat's own tile rectangle), which the
    // per-splat launch computes for itself and reads as ONE contiguous run per splat.  The scattered side of the exchange is
    // now the stores of this kernel, which is bound by instruction issue and does not wait for them.  A row carries the number
    // of the forward it belongs to (FitWs.stamp) in its eleventh float: pairs nobody walks -- behind the tile's deepest
    // contributor -- need no zero row.
    __shared__ RecLDS recs[FBB];
    __shared__ int32_t s_gid[FBB];
    __shared__ float acc[FBB][REC];
    __shared__ unsigned char s_mask[FBB];
    __shared__ int32_t s_max_last;
    __shared__ int32_t s_ticket;
    __shared__ int32_t s_simd[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the component this lane adds into acc[][] (SUMS 6 / 7: the last value is the depth feature's gradient, column 9)
    const int comp_few = SUMS == 6 ? reduce_scatter6_component(lane) : reduce_scatter7_component(lane);
    const int comp = SUMS == 10 ? reduce_scatter10_component(lane) : (comp_few == SUMS - 1 ? 9 : comp_few);
    // Which SIMD is this wave on?  The scheduler plans which SIMD walks which 8x8 block of every item (gfl_sched.hpp,
    // "block plan"); the plan is followed only if the workgroup's four waves sit on four different SIMDs (they do: the
    // dispatcher deals a workgroup's waves round the SIMDs), otherwise wave k walks block k.
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const int simd = (hw_id >> 4) & 3;
    if (lane == 0) s_simd[wave] = simd;
    __syncthreads();
    const bool simd_ok = ((1 << s_simd[0]) | (1 << s_simd[1]) | (1 << s_simd[2]) | (1 << s_simd[3])) == 15;
  for (bool first = true;; first = false) {
    const TileItem item = next_item(queue, &s_ticket, first, HEAVY_PARTS, gridDim.x);
    const int tile = item.tile;
    if (tile < 0) break;
    const unsigned plan = item.plan;
    const bool plan_ok = simd_ok && ((1 << (plan & 3)) | (1 << ((plan >> 2) & 3)) | (1 << ((plan >> 4) & 3)) | (1 << ((plan >> 6) & 3))) == 15;
    // (the segments of a heavy first tile turning the plan by one SIMD each was measured slower: tools/experiments/README.md)
    const int blk = plan_ok ? (int)((plan >> (2 * simd)) & 3u) : wave;          // the 8x8 block of the tile this wave walks
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * GFL_TILE + (blk & 1) * 8 + (lane & 7);
    const int py = ty * GFL_TILE + (blk >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = pixf(px), fy = pixf(py);
    const int start = tile_range[2 * tile], end = tile_range[2 * tile + 1];
    const int total = end - start;
    const int parts = item.part >= 0 ? heavy_parts(total) : 1;
    if (item.part >= parts) continue;                // this tile has fewer segments
    const int seg = heavy_seg(total, parts);
    const bool last_part = item.part < 0 || item.part == parts - 1;      // the farthest segment (or the whole tile)
    int units = 0;
#ifdef GFL_TRACE
    const long long trace_t0 = wall_clock64();
    int trace_lanes = 0;
#endif

    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, T = 1.f, S = 0.f;
    int last = 0;
    if (inside) {
        const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;
        last = n_contrib[pix];
        g0 = d_out[pix]; g1 = d_out[plane + pix]; g2 = d_out[2 * plane + pix]; g3 = d_out[3 * plane + pix];
        if (last_part) {
            T = final_T[pix];
            S = T * bg * (g0 + g1 + g2 + g3);
        } else {
            // state in front of this segment's far boundary, from the forward checkpoint: T as it
            // was there and S = sum_c g_c * (everything blended behind it) = sum_c g_c * (out_c - C_c)
            const float* ck = ckpt + ((size_t)item.queue * (HEAVY_PARTS - 1) + item.part) * 5 * 256 + blk * 64 + lane;
            T = ck[0];
            S = g0 * (render[pix] - ck[256]) + g1 * (render[plane + pix] - ck[512]) +
                g2 * (render[2 * plane + pix] - ck[768]) + g3 * (render[3 * plane + pix] - ck[1024]);
        }
    }
    if (tid == 0) s_max_last = 0;
    __syncthreads();
    const unsigned long long alive0 = __ballot(inside);
    const int px0w = tx * GFL_TILE + (blk & 1) * 8, py0w = ty * GFL_TILE + (blk >> 1) * 8;
    int wave_last = last;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, off));
    if (lane == 0) atomicMax(&s_max_last, wave_last);
    __syncthreads();
    const int depth_n = min(total, (int)s_max_last);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // this item walks list positions hi-1 down to lo
    const int lo = item.part > 0 ? item.part * seg : 0;
    const int hi = last_part ? depth_n : min((item.part + 1) * seg, depth_n);

    const int stamp = *stamp_ptr;
    for (int r0 = 0; r0 < hi - lo; r0 += FBB) {
        const int pos_t = tid < FBB ? hi - 1 - r0 - tid : -1;          // slot tid <-> list position pos_t
        __syncthreads();
        if (pos_t >= lo) {
            const int g = ids[start + pos_t];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * REC);
            const float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
            recs[tid].p0 = p0; recs[tid].p1 = p1; recs[tid].p2 = p2;
            s_gid[tid] = g;
            s_mask[tid] = (unsigned char)block_mask(p0, p1, p2.z, tx * GFL_TILE, ty * GFL_TILE);
        }
        if (tid < FBB) {
            float4* az = reinterpret_cast<float4*>(&acc[tid][0]);
            az[0] = zero4; az[1] = zero4; az[2] = zero4;
        }
        __syncthreads();
        const int cnt = min(FBB, hi - lo - r0);
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            const int slot = c0 + lane;
            const int spos = hi - 1 - r0 - slot;
            bool hit = slot < cnt && spos < wave_last && ((s_mask[slot] >> blk) & 1);
            {
                // as in the forward pass: only the pixels whose last contributor lies at or behind this group of 64
                // positions can receive anything from it; splats that do not reach their bounding box are skipped
                // before their alpha is evaluated (the test of block_mask on a smaller box)
                const unsigned long long alive = __ballot(last > hi - 1 - r0 - c0 - 63);
                if (alive != alive0 && alive != 0ull) {
                    unsigned long long a = alive | (alive >> 32);
                    a |= a >> 16;
                    a |= a >> 8;
                    const unsigned cols = (unsigned)a & 0xffu;
                    const int xl = __builtin_ctz(cols), xh = 31 - __builtin_clz(cols);
                    const int yl = (int)__builtin_ctzll(alive) >> 3, yh = (63 - (int)__builtin_clzll(alive)) >> 3;
                    if (hit) {
                        const BlockTest t = block_test(recs[slot].p0, recs[slot].p1, recs[slot].p2.z);
                        hit = box_hit(t, (float)(px0w + xl), (float)(px0w + xh), (float)(py0w + yl), (float)(py0w + yh));
                    }
                }
            }
            unsigned long long bits = __ballot(hit);
            while (bits) {
                const int j = c0 + (int)__builtin_ctzll(bits);
                bits &= bits - 1;
                const int pos = hi - 1 - r0 - j;
                const float4 p0 = recs[j].p0, p1 = recs[j].p1, p2 = recs[j].p2;
                float alpha, G;
                const bool valid = splat_alpha2(p0, p1, fx, fy, alpha, G) && (pos < last);
                if (__ballot(valid) == 0ull) continue;
#ifdef GFL_TRACE
                trace_lanes += __popcll(__ballot(valid));
#endif
                ++units;                     // wave-uniform: work feedback for the tile scheduler
                // (an interleaved two-splat version of this body was measured slower, twice)
                float v[10];
                blend_bwd_terms(p0, p1, p2, fx, fy, valid, alpha, G, g0, g1, g2, g3, T, S, v);
                float mine;
                if (SUMS == 6) {
                    const float v6[6] = {v[0], v[1], v[2], v[3], v[4], v[9]};
                    mine = wave_reduce_scatter6(v6, lane);
                } else if (SUMS == 7) {
                    const float v7[7] = {v[0], v[1], v[2], v[3], v[4], v[5], v[9]};
                    mine = wave_reduce_scatter7(v7, lane);
                } else {
                    mine = wave_reduce_scatter10(v, lane);
                }
                if (comp >= 0) atomicAdd(&acc[j][comp], mine);
            }
        }
        __syncthreads();
        if (pos_t >= lo) {
            // the pair's row: the tile's index inside the splat's tile rectangle (tile_rect of the staged u, v, radius: what the
            // binning walked and the per-splat launch will walk)
            const int g = s_gid[tid];
            const float4 p0 = recs[tid].p0;
            int x0, x1, y0, y1;
            tile_rect(p0.x, p0.y, __float_as_int(recs[tid].p2.w), gx, gy, x0, x1, y0, y1);
            const int nx = x1 - x0, j = (ty - y0) * nx + (tx - x0);
            long long row = (long long)g * SLOT_MAX + j;
            if (nx * (y1 - y0) > SLOT_MAX) {
                const int off = wide_off[g];
                row = off >= 0 ? wide_base + off + j : -1;
            }
            if (row >= 0) {
                const float4* a4 = reinterpret_cast<const float4*>(&acc[tid][0]);
                float4* o = reinterpret_cast<float4*>(pair_grad + (size_t)row * PG);
                o[0] = a4[0]; o[1] = a4[1];
                o[2] = make_float4(a4[2].x, a4[2].y, __int_as_float(stamp), 0.f);
            }
        }
    }
    // work feedback for the next iteration's schedule, per 8x8 block (gfl_sched.hpp)
    if (lane == 0) atomicAdd(&tile_work[4 * tile + blk], units + 1);
#ifdef GFL_TRACE
    if (lane == 0 && tile < 2048) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        long long* tr = g_bwd_trace + (size_t)(tile + (last_part ? 0 : 2048 * (1 + item.part))) * 8;
        if (wave == 0) {
            tr[0] = trace_t0; tr[1] = wall_clock64();
            tr[2] = ((long long)total << 32) | (unsigned)depth_n;
            tr[3] = ((long long)(xcc & 15) << 32) | hw;
        }
        tr[4 + wave] = ((long long)(trace_lanes | ((hw >> 4) & 3) << 28) << 32) | (unsigned)units;   // bits 60-61: this wave's SIMD
    }
#endif
  }
}

// ---- launcher (gfl_fit.hpp).  sums: 10 / 7 / 6 (see the kernel)
void launch_blend_bwd(const gfl_fit_state* st, float bg, int gx, int gy, int grid, int sums, const float* d_out, const TileQueue& q,
                      const FitWs& w, const LossTail& lt, hipStream_t s) {
    auto kern = sums == 6 ? fused_blend_bwd_kernel<6> : (sums == 7 ? fused_blend_bwd_kernel<7> : fused_blend_bwd_kernel<10>);
    kern<<<grid, 256, 0, s>>>(st->rec, st->ids, st->tile_range, bg, st->W, st->H, gx, st->final_T, st->n_contrib, d_out, w.pair_grad,
                              q, w.sched.work, w.ckpt, st->render, lt, gy, w.wide_off, w.wide_base, w.stamp);
}

#ifdef GFL_TRACE
int read_bwd_trace(long long* out, int n_tiles) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_trace), (size_t)n_tiles * 8 * sizeof(long long));
}
#endif

}  // namespace gfl
