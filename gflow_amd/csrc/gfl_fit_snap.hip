// Fused fit iteration, the snapshot's own kernels (trainer.py:573-582 keeps three images every 10th iteration): the "center"
// composite, the range of the splats' depths for the turbo map, the conversion of three float images to uint8 and the copy of what
// a forward left behind into a second engine.  (depth_map_color is the forward blend itself, modes 1 / 2 of gfl_fit_fwd.hip.)
#include "gfl_fit.hpp"

namespace gfl {

// ------------------------------------------------- the "center" image of a snapshot (render.py:98-106)
// alpha_blending over the SAME sorted lists with conic (1, 0, 1) and opacity 1: a unit blob at every splat's centre, which reaches
// pixels within sqrt(2 ln 255) = 3.33 of it and nobody else.  Until round 5 this was the blend kernel in a mode of its own -- 57 us
// on the side stream every tenth iteration, as long as the fit's own forward, although a pixel sees a handful of blobs: that kernel
// is made for long walks (queues, block plans, checkpoints, the four-CU walk of a pile), and what it costs here is its per-item
// latency.  This one is the footprint kernel's shape: a workgroup per tile, all tiles resident at once, lanes = pixels, a wave per
// 8x8 block; the arithmetic of a pixel is the blend kernel's, term for term (splat_alpha2 on the substituted record, one splat at
// a time in list order).
__global__ void __launch_bounds__(256) center_blend_kernel(const float* __restrict__ rec, const int32_t* __restrict__ ids,
                                                           const int32_t* __restrict__ tile_range, float bg, int W, int H, int gx,
                                                           float* __restrict__ out, uint8_t* __restrict__ out_u8) {
    __shared__ RecLDS recs[FB + 1];          // recs[FB]: an all-zero record (opacity 0: never blends)
    __shared__ unsigned char s_mask[FB];
    if (threadIdx.x == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        recs[FB].p0 = z; recs[FB].p1 = z; recs[FB].p2 = z;
    }
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * GFL_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GFL_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = pixf(px), fy = pixf(py);
    const int start = tile_range[2 * tile], end = tile_range[2 * tile + 1];
    const float blob_cutoff = alpha_cutoff(1.f, 1.f);
    float T = 1.f, Tw = inside ? 1.f : 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int base = start; base < end; base += FB) {
        if (__syncthreads_and(Tw == 0.f)) break;
        const int idx = base + tid;
        unsigned char m = 0;
        if (idx < end) {
            const int g = ids[idx];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * REC);
            float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
            p0.z = 1.f; p0.w = 0.f; p1.x = 1.f; p1.y = 1.f;
            p2.z = p2.z < 0.f ? p2.z : blob_cutoff;
            recs[tid].p0 = p0; recs[tid].p1 = p1; recs[tid].p2 = p2;
            m = (unsigned char)block_mask(p0, p1, p2.z, tx * GFL_TILE, ty * GFL_TILE);
        }
        s_mask[tid] = m;
        __syncthreads();
        const int cnt = min(FB, end - base);
        for (int c0 = 0; c0 < cnt && !__all(Tw == 0.f); c0 += 64) {
            const int slot = c0 + lane;
            unsigned long long bits = __ballot(slot < cnt && ((s_mask[slot] >> wave) & 1));
            // four hit blobs per trip, as the blend kernel's whole-tile walk: records fetched and alphas evaluated together, only
            // the T recurrence is serial (a pile's thousand blobs are walked by ONE wave for the ring of pixels around it)
            while (bits) {
                int j[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    j[u] = bits ? c0 + (int)__builtin_ctzll(bits) : FB;      // (missing blobs of the last trip: the null record)
                    bits &= bits - 1;
                }
                float4 q0[4], q1[4];
                float cb[4], al[4];
                bool val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { q0[u] = recs[j[u]].p0; q1[u] = recs[j[u]].p1; cb[u] = recs[j[u]].p2.x; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float G;
                    val[u] = splat_alpha2(q0[u], q1[u], fx, fy, al[u], G);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a = val[u] ? al[u] : 0.f;
                    const float test_T = Tw * (1.f - a);
                    const bool stop = test_T < GFL_T_MIN;
                    const float w = stop ? 0.f : a * Tw;
                    a0 = fmaf(q1[u].z, w, a0); a1 = fmaf(q1[u].w, w, a1); a2 = fmaf(cb[u], w, a2);
                    T = stop ? T : test_T;
                    Tw = stop ? 0.f : test_T;
                }
                if (__all(Tw == 0.f)) break;
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;
        const float o0 = fmaf(T, bg, a0), o1 = fmaf(T, bg, a1), o2 = fmaf(T, bg, a2);
        if (out_u8) {                                    // (a snapshot iteration: straight into the uint8 image)
            uint8_t* u = out_u8 + pix * 3;
            u[0] = img_u8(o0); u[1] = img_u8(o1); u[2] = img_u8(o2);
        } else {
            out[pix] = o0; out[plane + pix] = o1; out[2 * plane + pix] = o2;
        }
    }
}

// min over the non-zero / max over all depths of the records, as ordered-uint keys (the range of
// apply_float_colormap(non_zero=True), color.py:28-31; the encoding of cmap_range_kernel of gfl_loss.hip, the minimum
// COMPLEMENTED so that both words are initialised by the one memset that also clears the snapshot's pull counters)
__global__ void __launch_bounds__(256) rec_depth_range_kernel(const float* __restrict__ rec, int N, unsigned* __restrict__ mm) {
    unsigned lo = 0xffffffffu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const float x = rec[(size_t)i * REC + 9];
        const unsigned b = __float_as_uint(x);
        const unsigned k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        if (x != 0.f) lo = min(lo, k);
        hi = max(hi, k);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, off));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&mm[0], ~lo);
        atomicMax(&mm[1], hi);
    }
}

// three float images [3][H][W] -> uint8 [3 images][H][W][3]: clamp to [0,1], x 255, truncate (render.py:158-166)
__global__ void __launch_bounds__(256) snapshot_u8_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ c, int P, uint8_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float* src[3] = {a, b, c};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            out[((size_t)k * P + i) * 3 + ch] = img_u8(src[k][(size_t)ch * P + i]);
        }
    }
}

// Everything gfl_fit_snapshot reads of a forward -- records, sorted ids, tile ranges, the rgb planes of the render, the
// forward's tile queues -- copied from one engine to another in ONE launch (gfl_fit_snapshot_stage).  The number of ids
// is a device value (tile_offsets[T]).
__global__ void __launch_bounds__(256) snapshot_stage_kernel(StageCopy c) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
#pragma unroll
    for (int sgi = 0; sgi < 7; ++sgi) {
        const StageSeg sg = c.seg[sgi];
        unsigned n = sg.n;
        if (sgi == 0) n = min((unsigned)max(*c.k_ptr, 0), c.ids_cap);      // segment 0: the ids
        const unsigned n4 = n >> 2;
        const uint4* s4 = reinterpret_cast<const uint4*>(sg.src);
        uint4* d4 = reinterpret_cast<uint4*>(sg.dst);
        for (unsigned i = tid; i < n4; i += stride) d4[i] = s4[i];
        for (unsigned i = (n4 << 2) + tid; i < n; i += stride) sg.dst[i] = sg.src[i];
    }
}

void launch_center_blend(const gfl_fit_state* st, float bg, int gx, int T, float* out, uint8_t* out_u8, hipStream_t s) {
    center_blend_kernel<<<T, 256, 0, s>>>(st->rec, st->ids, st->tile_range, bg, st->W, st->H, gx, out, out_u8);
}

void launch_rec_depth_range(const float* rec, int N, unsigned* mm, hipStream_t s) {
    // (few blocks: a thousand waves hitting the two result words with atomics took 23 us)
    rec_depth_range_kernel<<<min((N + 255) / 256, 32), 256, 0, s>>>(rec, N, mm);
}

void launch_snapshot_u8(const float* a, const float* b, const float* c, int P, uint8_t* out, hipStream_t s) {
    snapshot_u8_kernel<<<(P + 255) / 256, 256, 0, s>>>(a, b, c, P, out);
}

void launch_snapshot_stage(const StageCopy& c, hipStream_t s) { snapshot_stage_kernel<<<1024, 256, 0, s>>>(c); }

}  // namespace gfl
