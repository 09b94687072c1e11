// Fused fit iteration, stage 2: front-to-back compositing (C = 4: rgb + depth) over the sorted per-tile lists, the
// moving-splat footprint of the camera-only stage, and the snapshot's kernels (its depth_map_color image is this file's blend
// kernel in mode 1, its center image a kernel of its own).
#include "gfl_fit.hpp"

#ifndef GFL_FWD_PARTS
#define GFL_FWD_PARTS 4      // items the first tile of a forward queue counts as (next_item)
#endif

namespace gfl {

// (RecLDS, splat_alpha2: gfl_fit.hpp; block_test / box_hit / block_mask: gfl_math.hpp)

#ifdef GFL_TRACE
__device__ long long g_fwd_trace[16384 * 8];     // analysis build: per-tile timeline of the forward blend
__device__ long long g_fwd_trace2[4096 * 16 * 4]; // ... of a long first tile's sixteen quarter waves: staging ticks, walk ticks, steps, units
#endif

// apply_float_colormap(depth, "turbo", non_zero=True) for one value (color.py:24-44): mm = ordered-uint encodings of
// min over the non-zero values and max over all (cmap_range_kernel, gfl_loss.hip)
__device__ __forceinline__ float3 cmap_nonzero_lookup(float v, const unsigned* __restrict__ mm, const float* __restrict__ lut) {
    const unsigned k0 = ~mm[0], k1 = mm[1];          // (the minimum is kept complemented: both words start from zero)
    const float lo = (k0 == 0xffffffffu) ? 0.f : __uint_as_float((k0 & 0x80000000u) ? (k0 & 0x7fffffffu) : ~k0);
    const float hi = __uint_as_float((k1 & 0x80000000u) ? (k1 & 0x7fffffffu) : ~k1) - lo;
    float x = (v - lo) / (hi + 1e-5f);
    x = fminf(fmaxf(x, 0.f), 1.f);
    if (x != x) x = 0.f;
    const int idx = (int)(x * 255.f);
    return make_float3(lut[3 * idx], lut[3 * idx + 1], lut[3 * idx + 2]);
}

// sum over the four 16-lane rows of the wave, in every lane
__device__ __forceinline__ float rows_sum(float x) {
    float lo = x, hi = x;
    permlane32_swap(lo, hi);                 // rows {0, 1, 0, 1} / {2, 3, 2, 3}
    float y0 = lo + hi, y1 = y0;
    permlane16_swap(y0, y1);                 // rows {0+2} x 4 / {1+3} x 4
    return y0 + y1;
}

#define GFL_ROW_SHR(old, val, n) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (val)), 0x110 + (n), 0xF, 0xF, false))

// inclusive scans along the sixteen lanes of a row.  One DPP instruction per step: v_mul_f32_dpp x, x(row_shr:n), x -- a lane
// whose source lies outside the row is DISABLED by the instruction (bound_ctrl off) and keeps its x, which is what an
// inclusive scan wants.  (Written with update_dpp + multiply the compiler emitted three instructions per step: the
// identity, the DPP move, the product.)  The s_nop covers the VALU-write -> DPP-read hazard hipcc does not pad inside asm.
__device__ __forceinline__ float row_scan_mul(float x) {
    asm("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
        : "+v"(x));
    return x;
}
__device__ __forceinline__ float row_scan_add(float x) {
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
        : "+v"(x));
    return x;
}
__device__ __forceinline__ float row_sum16(float x) {          // sum over the sixteen lanes of the row, in every lane
#define GFL_ROW_ROR(val, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (val)), 0x120 + (n), 0xF, 0xF, true))
    x += GFL_ROW_ROR(x, 1);
    x += GFL_ROW_ROR(x, 2);
    x += GFL_ROW_ROR(x, 4);
    x += GFL_ROW_ROR(x, 8);
    return x;
}
__device__ __forceinline__ int row_max16(int x) {
#define GFL_ROW_ROR_I(val, n) __builtin_amdgcn_update_dpp(0, (val), 0x120 + (n), 0xF, 0xF, true)
    x = max(x, GFL_ROW_ROR_I(x, 1));
    x = max(x, GFL_ROW_ROR_I(x, 2));
    x = max(x, GFL_ROW_ROR_I(x, 4));
    x = max(x, GFL_ROW_ROR_I(x, 8));
    return x;
}
// lane 15 of the own row, in every lane of the row (ds_swizzle, bit-mask mode inside 32 lanes: lane' = (lane & 0x10) | 0x0f)
__device__ __forceinline__ float row_last(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x10 | (0x0F << 5)));
}

// ------------------------------------------------- moving-splat footprint (camera-only stage)
// GFlow renders the tentative moving splats on their own and masks every pixel whose grey value
// is > 0 (trainer.py:426-451).  With a black background that is exactly the set of pixels some
// moving splat reaches with alpha >= 1/255 in a tile it was binned into: the first such splat in
// depth order always blends (T = 1), and every colour is a sigmoid, hence > 0.  So no second sort
// and composite: the flagged splats of every tile list mark their pixels, in any order.
// One workgroup per tile walks the tile's list (already binned and sorted by the forward that just
// ran) and evaluates only the flagged splats, lanes = pixels as in the blend; a wave stops as soon
// as all its pixels are marked.  (A first version gave every flagged splat one wave that walked the
// splat's bounding box: later frames of a clip grow moving splats hundreds of pixels wide, and
// that launch then took 325 us.)
// Round 5: in the fit's own iterations these workgroups ride BEHIND the blend kernel's in the same launch (mode 3 of
// fused_blend_fwd_kernel: blocks past the blend grid each take a tile here) -- they are dispatched as the CUs whose queues have run
// dry free their registers, i.e. into the launch's idle tail, instead of a launch of their own behind it (9.8 us, a third of a
// clip's iterations).
__device__ __forceinline__ void footprint_tile(const float* __restrict__ rec, const int32_t* __restrict__ ids,
                                               const int32_t* __restrict__ tile_range, const uint8_t* __restrict__ foot_flags,
                                               int W, int H, int gx, uint8_t* __restrict__ keep, int tile, RecLDS* recs,
                                               unsigned char* s_mask) {
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * GFL_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GFL_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = pixf(px), fy = pixf(py);
    const int start = tile_range[2 * tile], end = tile_range[2 * tile + 1];
    bool marked = !inside;                               // nothing left to find for this lane
    for (int base = start; base < end; base += FB) {
        if (__syncthreads_and(marked)) break;
        const int idx = base + tid;
        unsigned char m = 0;
        if (idx < end) {
            const int g = ids[idx];
            if (foot_flags[g]) {
                const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * REC);
                const float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
                recs[tid].p0 = p0; recs[tid].p1 = p1;
                m = (unsigned char)block_mask(p0, p1, p2.z, tx * GFL_TILE, ty * GFL_TILE);
            }
        }
        s_mask[tid] = m;
        __syncthreads();
        const int cnt = min(FB, end - base);
        for (int c0 = 0; c0 < cnt && !__all(marked); c0 += 64) {
            const int slot = c0 + lane;
            unsigned long long bits = __ballot(slot < cnt && ((s_mask[slot] >> wave) & 1));
            while (bits) {
                const int j = c0 + (int)__builtin_ctzll(bits);
                bits &= bits - 1;
                float alpha, G;
                if (splat_alpha2(recs[j].p0, recs[j].p1, fx, fy, alpha, G)) marked = true;
            }
        }
    }
    if (inside && marked) keep[(size_t)py * W + px] = 0;
}
__global__ void __launch_bounds__(256) footprint_kernel(const float* __restrict__ rec, const int32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ tile_range,
                                                        const uint8_t* __restrict__ foot_flags, int W, int H, int gx,
                                                        uint8_t* __restrict__ keep) {
    __shared__ RecLDS recs[FB];
    __shared__ unsigned char s_mask[FB];
    footprint_tile(rec, ids, tile_range, foot_flags, W, H, gx, keep, blockIdx.x, recs, s_mask);
}
struct FootArgs {            // mode 3 of the blend kernel
    const uint8_t* flags;
    uint8_t* keep;
    int n_blend;             // workgroups of the blend itself; block n_blend + t takes tile t's footprint
};

// (mode: a template parameter -- the fit's forward, mode 0, carries none of the snapshot composites' code or constants)
// mode 0: the fit's forward.  mode 1: the depth_map_color composite alone (gfl_fit_snapshot: a snapshot of a forward that has
// already run).  mode 2: the forward of a SNAPSHOT iteration (gfl_fit_iteration_snapshot) -- the fit's forward, and in the same
// walk the depth_map_color composite: the same alphas, transmittances and stop rule, three more sums per pixel with the turbo
// colour of the splat's depth as the colour (render.py:76-91 blends the same lists a second time); both images leave as uint8
// (snap_u8: [image][H][W][3], images 0 and 1), the render's float planes as always.  Round 5, until then: a second blend launch
// on a side stream, 53 us every tenth iteration for the 4 us the three sums cost here.
template <int mode>
__global__ void __launch_bounds__(256, mode == 2 ? FWD_WG_PER_CU - 1 : FWD_WG_PER_CU) fused_blend_fwd_kernel(const float* __restrict__ rec,
                                                              const int32_t* __restrict__ ids,
                                                              const int32_t* __restrict__ tile_range, float bg, int W,
                                                              int H, int gx, unsigned inv_gx, float* __restrict__ out,
                                                              float* __restrict__ final_T,
                                                              int32_t* __restrict__ n_contrib, TileQueue queue,
                                                              float* __restrict__ ckpt,
                                                              const unsigned* __restrict__ cmap_mm,
                                                              const float* __restrict__ cmap_lut, int split_min,
                                                              int32_t* __restrict__ tile_work,
                                                              const int32_t* __restrict__ first_slot,
                                                              int32_t* __restrict__ stamp, uint8_t* __restrict__ snap_u8,
                                                              FootArgs foot) {
    constexpr bool FIT = mode != 1;          // the fit's own forward: stamp, work feedback
    constexpr bool DC = mode == 2;           // ... which also composes depth_map_color
    constexpr bool FOOT = mode == 3;         // ... with the footprint workgroups of the camera-only stage behind its own
    // the number of this forward: the backward blend stamps its pair rows with it, the per-splat launch takes only rows that
    // carry it (FitWs.stamp; nobody reads it before this launch has ended)
    if (FIT && stamp && blockIdx.x == 0 && threadIdx.x == 0) *stamp += 1;
    // mode 0: the records as they are.  mode 1: the depth_map_color image of render.py:76-91, a composite of the SAME lists
    // with colour := turbo map of the splat's depth (apply_float_colormap(non_zero=True), range in cmap_mm), substituted while
    // a record is staged.  (The third snapshot image, "center", has a kernel of its own below.)
    __shared__ RecLDS recs[FBL + 1];         // recs[FBL]: an all-zero record (opacity 0: never blends)
    __shared__ float4 s_dc[DC ? FBL + 1 : 1];       // mode 2: the turbo colour of every staged splat's depth ([FBL]: zero)
    __shared__ unsigned char s_mask[FBL];
    __shared__ unsigned short s_hits[4][FBL];        // long first tiles: a wave's (= a 4x4 quarter's) hit list of the staged batch
    __shared__ int32_t s_gs[4][FBL / 64 + 1];        // ... and the number of hits in front of every 64-slot group
    __shared__ int32_t s_ticket;
    __shared__ int32_t s_simd[4];
    __shared__ int32_t s_vote[2];
    if constexpr (FOOT) {
        if ((int)blockIdx.x >= foot.n_blend) {
            footprint_tile(rec, ids, tile_range, foot.flags, W, H, gx, foot.keep, (int)blockIdx.x - foot.n_blend, recs, s_mask);
            return;
        }
    }
    if (threadIdx.x == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        recs[FBL].p0 = z; recs[FBL].p1 = z; recs[FBL].p2 = z;
        if (DC) s_dc[FBL] = z;
        s_vote[0] = 0; s_vote[1] = 0;
    }
    WgVote vote = wg_vote_init(s_vote);
    // block plan (gfl_sched.hpp): which 8x8 block of a whole tile this wave walks follows from the SIMD it sits on
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const int simd = (hw_id >> 4) & 3;
    if ((threadIdx.x & 63) == 0) s_simd[threadIdx.x >> 6] = simd;
    __syncthreads();
    const bool simd_ok = ((1 << s_simd[0]) | (1 << s_simd[1]) | (1 << s_simd[2]) | (1 << s_simd[3])) == 15;
  for (bool first = true;; first = false) {
    // (lane-derived values are formed again for every item, from an id the compiler cannot see through: hoisted out of this
    //  loop they lived across both walks, and the kernel -- at its 96 registers -- kept four of them in scratch)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const TileItem item = next_item(queue, &s_ticket, first, GFL_FWD_PARTS, FOOT ? (unsigned)foot.n_blend : gridDim.x);
    if (item.tile < 0) {
        if (item.part < 0) break;                    // the queue is empty
        if (item.part == 0) continue;                // ... but its items 1..3 may have to help other queues
    }
    // The first (heaviest) tile of every queue comes as several items (the backward pass walks it in up to eight
    // segments; the forward pass uses the first four).  When its list is
    // long, the forward pass walks it as four 8x8 BLOCKS: item 0 on the tile's own CU, items 1..3 of THIS queue
    // as helpers for the first tiles of three OTHER queues (side by side on one CU the four would share its SIMDs,
    // and a long tile's waves are bound by their own issue rate: ~8 cycles per instruction alone, ~14 with three
    // others).  A block's workgroup gives its waves the block's four 4x4 quarters, sixteen lanes each: a quarter is
    // reached by less than half of the splats that reach the block, and the launch lasts as long as the longest chain of
    // one wave (real fits pile ~1 000 splats into single tiles: their waves finished at 50-73 us, the mean CU at 29 us).
    int tile = item.tile, owner = item.queue;
    if (item.part >= 4) continue;                    // (only with -DGFL_FWD_PARTS=8, the mapping of rounds 2-4)
    if (item.part > 0) {
        owner = (item.queue + item.part * (queue.nq / 4)) % queue.nq;
        tile = queue.count[owner] > 0 ? (queue.list[(size_t)owner * queue.cap_q] & 0xffff) : -1;
        if (tile < 0) continue;
    }
    int tx, ty;
    tile_xy(tile, gx, inv_gx, tx, ty);
    const int start = tile_range[2 * tile], end = tile_range[2 * tile + 1];
    const bool first_tile = item.part >= 0;
    const int blk = (first_tile && end - start > split_min) ? item.part : -1;
    if (first_tile && blk < 0 && item.part > 0) continue;                     // not long enough: its own CU walks it whole
    // leave the per-pixel state at the split positions for the backward pass, which walks the first tile of each of
    // ITS queues in segments (first_slot: that queue, -1 for every other tile), at the place of the thread that owns the
    // pixel in the whole-tile layout
    const int slot = first_slot[tile];
    const bool heavy = slot >= 0;
    const int parts = heavy ? heavy_parts(end - start) : 1;
    const int seg = heavy_seg(end - start, parts);
    int units = 0;                                   // work feedback for the forward schedule (wave-uniform)
    if (blk >= 0) {
        // ---- A long first tile: this workgroup walks its 8x8 block `blk`, wave k the block's 4x4 quarter k, SIXTEEN hit
        // splats of the quarter per step (round 4; round 2 took four, one per 16-lane row, every row running the same chain).
        // lane = (s, r): s = lane & 15 = the splat of the step, in list order along the sixteen lanes of a row; r = lane >> 4 =
        // the pixel column of the quarter; four passes g over the quarter's pixel rows.  The transmittance along the sixteen
        // splats is a ROW SCAN (four DPP multiplies) instead of a chain -- r_k = T_in * prod_{j<=k} (1 - a_j) --, a lane adds
        // only its own splat's colour, and the sixteen partial sums of a pixel are folded at the checkpoints and at the end.
        // ~35 VALU instructions per pass, ~150 per step of sixteen (splat, quarter) units, against ~100 per step of four:
        // the chains of the piles that densification leaves in single tiles set the duration of this launch on real fits
        // (tools/bwd_trace.py --fwd --fit: ~120 steps of 0.41 us in the slowest quarter of a 1 260-entry list).
        // The products are formed in tree order: T differs from the one-splat-at-a-time walk in the last bit; the stop rule
        // (the FIRST splat behind which T would fall below 1e-4) and the last contributor are found on the scanned values.
        const int ls = lane & 15, lr = lane >> 4;
        const int qx0 = tx * GFL_TILE + (blk & 1) * 8 + (wave & 1) * 4, qy0 = ty * GFL_TILE + (blk >> 1) * 8 + (wave >> 1) * 4;
        const float fxq = pixf(qx0 + lr);
        float Tq[4], c0q[4], c1q[4], c2q[4], c3q[4];
        float d0q[4], d1q[4], d2q[4];                // (mode 2: depth_map_color)
        int lastq[4];
        unsigned alive = 0;                          // bit g: pixel (column lr, row g) is in the image and has not stopped
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            Tq[g] = 1.f; c0q[g] = 0.f; c1q[g] = 0.f; c2q[g] = 0.f; c3q[g] = 0.f; lastq[g] = 0;
            d0q[g] = 0.f; d1q[g] = 0.f; d2q[g] = 0.f;
            if (qx0 + lr < W && qy0 + g < H) alive |= 1u << g;
        }
        // (checkpoint layout of the backward pass: [boundary][T C0 C1 C2 C3][256 pixels of the tile, block-major])
        float* ckq = ckpt + (size_t)max(slot, 0) * (HEAVY_PARTS - 1) * 5 * 256 + blk * 64 + ((((wave >> 1) * 4) << 3) | ((wave & 1) * 4 + lr));
        auto write_ckpt = [&](int k) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float s0 = row_sum16(c0q[g]), s1 = row_sum16(c1q[g]), s2 = row_sum16(c2q[g]), s3 = row_sum16(c3q[g]);
                if (ls == 0) {
                    float* c5 = ckq + (size_t)(k - 1) * 5 * 256 + (g << 3);
                    c5[0] = Tq[g]; c5[256] = s0; c5[512] = s1; c5[768] = s2; c5[1024] = s3;
                }
            }
        };
        int ck_nextq = 1;
#ifdef GFL_TRACE
        long long tq_stage = 0, tq_walk = 0, tq_mark = wall_clock64();
        int tq_steps = 0;
#endif
        for (int base = start; base < end; base += FBL) {
#ifdef GFL_TRACE
            { const long long now = wall_clock64(); tq_walk += now - tq_mark; tq_mark = now; }
#endif
            if (wg_all(vote, alive == 0, 4)) break;
            {
                // FBL / 256 entries per lane, their ids and then their records requested together
                constexpr int PER = FBL / 256;
                int gidx[PER];
#pragma unroll
                for (int e = 0; e < PER; ++e) gidx[e] = base + tid + 256 * e < end ? ids[base + tid + 256 * e] : -1;
                // (requesting the NEXT batch's ids here, a whole walk ahead, was measured again in round 4: forward inside a
                //  clip fit 53.7 against 50.7 us without it, same box.  Round 5: ids a batch ahead AND the records of the next
                //  batch straight into a second set of LDS planes with global_load_lds while this one is walked -- no
                //  registers, no wait left inside the walk (ISA checked) -- neutral: clip fits 0.480 / 0.449 s against
                //  0.456 / 0.453 s, bench forward 41.0-41.1 us both ways; tools/experiments/blend_fwd_lds_dma_prefetch.hip.txt.
                //  The trace says why: the slowest quarter wave of a pile spends 35-45 us in its 30-42 steps, 1.1 us each
                //  with four other walks on its SIMD, and ~10 us at the batches' barriers waiting for ITS OWN kind.)
#pragma unroll 1
                for (int e = 0; e < PER; ++e) {          // (one record at a time: two in flight spilled the walk's registers)
                    if (gidx[e] < 0) continue;
                    const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)gidx[e] * REC);
                    float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
                    if (mode == 1) {
                        const float3 col = cmap_nonzero_lookup(p2.y, cmap_mm, cmap_lut);
                        p1.z = col.x; p1.w = col.y; p2.x = col.z;
                    }
                    const int sl = tid + 256 * e;
                    if (DC) {
                        const float3 col = cmap_nonzero_lookup(p2.y, cmap_mm, cmap_lut);
                        s_dc[sl] = make_float4(col.x, col.y, col.z, 0.f);
                    }
                    recs[sl].p0 = p0; recs[sl].p1 = p1; recs[sl].p2 = p2;
                    s_mask[sl] = (unsigned char)block_mask(p0, p1, p2.z, tx * GFL_TILE + (blk & 1) * 8, ty * GFL_TILE + (blk >> 1) * 8, 4);
                }
            }
            __syncthreads();
#ifdef GFL_TRACE
            { const long long now = wall_clock64(); tq_stage += now - tq_mark; tq_mark = now; }
#endif
            const int cnt = min(FBL, end - base);
            if (__all(alive == 0)) continue;         // this wave is finished; keep meeting the barriers
            // ---- the quarter's hit list of the batch (slot order = list order); gs[k]: hits in front of 64-slot group k.
            // Once pixels have stopped, only splats that reach the box of the pixels still ALIVE matter (in a tile where
            // densification piled up a thousand small splats the pile's own pixels stop early and the rest of the pile
            // reaches no one else): the test of block_mask on that smaller box -- drops only what contributes nothing.
            // (the box test costs ~45 instructions per 64 slots: it is applied once half of the quarter's pixels have stopped)
            const bool all_alive = __popcll(__ballot((alive & 1u) != 0)) + __popcll(__ballot((alive & 2u) != 0)) +
                                   __popcll(__ballot((alive & 4u) != 0)) + __popcll(__ballot((alive & 8u) != 0)) > 8 * 16;
            float bx_lo = 0.f, bx_hi = 0.f, by_lo = 0.f, by_hi = 0.f;
            if (!all_alive) {
                unsigned arows = alive;
                arows |= (unsigned)__shfl_xor((int)arows, 16);
                arows |= (unsigned)__shfl_xor((int)arows, 32);           // pixel rows with an alive pixel (wave-uniform)
                const unsigned long long acol = __ballot(alive != 0);   // lanes of the columns with an alive pixel
                const unsigned cols = (unsigned)((acol & 1ull) | ((acol >> 15) & 2ull) | ((acol >> 30) & 4ull) | ((acol >> 45) & 8ull));
                bx_lo = (float)(qx0 + __builtin_ctz(cols | 16u)); bx_hi = (float)(qx0 + 31 - __builtin_clz(cols | 1u));
                by_lo = (float)(qy0 + __builtin_ctz(arows | 16u)); by_hi = (float)(qy0 + 31 - __builtin_clz(arows | 1u));
            }
            int n_hit = 0;
#pragma unroll
            for (int k = 0; k < FBL / 64; ++k) {
                if (lane == 0) s_gs[wave][k] = n_hit;
                const int sl = 64 * k + lane;
                bool hit = sl < cnt && ((s_mask[sl] >> wave) & 1);
                if (!all_alive && hit) {
                    const BlockTest t = block_test(recs[sl].p0, recs[sl].p1, recs[sl].p2.z);
                    hit = box_hit(t, bx_lo, bx_hi, by_lo, by_hi);
                }
                const unsigned long long bal = __ballot(hit);
                if (hit) s_hits[wave][n_hit + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0))] = (unsigned short)sl;
                n_hit += (int)__popcll(bal);
            }
            units += n_hit;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the list is read back by other lanes of this wave)
            // the steps, cut where the backward pass wants a checkpoint (a list position that is a multiple of 64)
            int h = 0, k = 0;
            for (;;) {
                int h_hi = n_hit;
                bool due = false;
                for (; k < FBL / 64; ++k)
                    if (64 * k < cnt && ck_nextq < parts && base - start + 64 * k == ck_nextq * seg) {
                        h_hi = s_gs[wave][k];
                        due = true;
                        break;
                    }
                for (; h < h_hi && !__all(alive == 0); h += 16) {
#ifdef GFL_TRACE
                    ++tq_steps;
#endif
                    const bool have = h + ls < h_hi;
                    const int j = have ? (int)s_hits[wave][h + ls] : FBL;
                    const int pos1 = base - start + j + 1;
                    const float4 q0 = recs[j].p0, q1 = recs[j].p1, q2 = recs[j].p2;
                    float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (DC) dq = s_dc[j];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float fyq = pixf(qy0 + g);
                        float al, G;
                        const bool val = splat_alpha2(q0, q1, fxq, fyq, al, G);
                        const float a = val ? al : 0.f;
                        const bool live = (alive >> g) & 1u;
                        const float P = row_scan_mul(1.f - a);                    // prod_{j<=k} (1 - a_j) along the row
                        const float Pex = GFL_ROW_SHR(1.0f, P, 1);                // prod_{j<k}
                        const float Tin = live ? Tq[g] : 0.f;
                        const float r = Tin * P, q = Tin * Pex;                   // T behind / in front of this lane's splat
                        bool stopped = false;
                        const unsigned long long sb = __ballot(live && r < GFL_T_MIN);
                        if (sb != 0ull) {
                            // (rare: a pixel stops at most once.)  Per row: the first lane whose splat would take T below the
                            // threshold; it and everything behind it blend with weight 0, T keeps the value in front of it
                            float t_new = Tq[g];
                            int fs_mine = 16;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const unsigned bits = (unsigned)(sb >> (16 * rr)) & 0xffffu;
                                if (bits == 0u) continue;
                                const int fs = __builtin_ctz(bits);
                                const float t_before = fs == 0 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, Tq[g]), 16 * rr))
                                                               : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), 16 * rr + fs - 1));
                                if (lr == rr) { fs_mine = fs; t_new = t_before; }
                            }
                            stopped = ls >= fs_mine;
                            if (fs_mine < 16) { Tq[g] = t_new; alive &= ~(1u << g); }
                        }
                        const float w = (stopped || !live) ? 0.f : a * q;
                        c0q[g] = fmaf(q1.z, w, c0q[g]); c1q[g] = fmaf(q1.w, w, c1q[g]);
                        c2q[g] = fmaf(q2.x, w, c2q[g]); c3q[g] = fmaf(q2.y, w, c3q[g]);
                        if (DC) { d0q[g] = fmaf(dq.x, w, d0q[g]); d1q[g] = fmaf(dq.y, w, d1q[g]); d2q[g] = fmaf(dq.z, w, d2q[g]); }
                        lastq[g] = max(lastq[g], (val && live && !stopped) ? pos1 : 0);
                        const float r15 = row_last(r);
                        if ((alive >> g) & 1u) Tq[g] = r15;                        // (rows that stopped keep t_new)
                    }
                    // (Measured and dropped: the four passes as one branch-free block -- alphas and scans of all four first,
                    //  ONE test for a stop, then the four accumulations --, so that a lone wave has four independent streams
                    //  to issue from: the forward's average inside a clip fit went from 53.9 to 59.2 us; the sixteen values
                    //  kept live across the phases cost more than the interleaving gave.)
                }
                h = h_hi;
                if (!due) break;
                write_ckpt(ck_nextq);
                ++ck_nextq;
                ++k;
            }
        }
        // the wave stopped before a split position: every pixel's state is frozen, final = checkpoint
        for (; ck_nextq < parts; ++ck_nextq) write_ckpt(ck_nextq);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float s0 = row_sum16(c0q[g]), s1 = row_sum16(c1q[g]), s2 = row_sum16(c2q[g]), s3 = row_sum16(c3q[g]);
            const int lastp = row_max16(lastq[g]);
            float e0 = 0.f, e1 = 0.f, e2 = 0.f;
            if (DC) { e0 = row_sum16(d0q[g]); e1 = row_sum16(d1q[g]); e2 = row_sum16(d2q[g]); }
            if (ls == 0 && qx0 + lr < W && qy0 + g < H) {
                const size_t pix = (size_t)(qy0 + g) * W + (qx0 + lr), plane = (size_t)H * W;
                const float Tf = Tq[g];
                const float o0 = fmaf(Tf, bg, s0), o1 = fmaf(Tf, bg, s1), o2 = fmaf(Tf, bg, s2);
                out[pix] = o0;
                out[plane + pix] = o1;
                out[2 * plane + pix] = o2;
                out[3 * plane + pix] = fmaf(Tf, bg, s3);
                final_T[pix] = Tf;
                n_contrib[pix] = lastp;
                if (DC) {
                    uint8_t* u0 = snap_u8 + pix * 3;
                    u0[0] = img_u8(o0); u0[1] = img_u8(o1); u0[2] = img_u8(o2);
                    uint8_t* u1 = snap_u8 + (plane + pix) * 3;
                    u1[0] = img_u8(fmaf(Tf, bg, e0)); u1[1] = img_u8(fmaf(Tf, bg, e1)); u1[2] = img_u8(fmaf(Tf, bg, e2));
                }
            }
        }
        if (FIT && lane == 0) atomicAdd(&tile_work[4 * tile + blk], units + 1);
#ifdef GFL_TRACE
        if (lane == 0 && tile < 4096 && FIT) {
            long long* t2 = g_fwd_trace2 + ((size_t)tile * 16 + blk * 4 + wave) * 4;
            t2[0] = tq_stage; t2[1] = tq_walk + (wall_clock64() - tq_mark); t2[2] = tq_steps; t2[3] = units;
        }
#endif
        continue;
    }
    // ---- every other tile: a wave walks an 8x8 block of the tile, the one the item's plan names for this wave's SIMD
    const unsigned plan = item.plan;
    const bool plan_ok = item.part <= 0 && simd_ok &&
                         ((1 << (plan & 3)) | (1 << ((plan >> 2) & 3)) | (1 << ((plan >> 4) & 3)) | (1 << ((plan >> 6) & 3))) == 15;
    const int wb = plan_ok ? (int)((plan >> (2 * simd)) & 3u) : wave;
    const int org_x = tx * GFL_TILE, org_y = ty * GFL_TILE;
    const int px0w = org_x + (wb & 1) * 8, py0w = org_y + (wb >> 1) * 8;
    const int px = px0w + (lane & 7), py = py0w + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = pixf(px), fy = pixf(py);
    float* ck = ckpt + (size_t)max(slot, 0) * (HEAVY_PARTS - 1) * 5 * 256 + wb * 64 + lane;
    int ck_next = 1;                                 // next boundary to checkpoint: position ck_next * seg
#ifdef GFL_TRACE
    const long long trace_t0 = wall_clock64();
    int trace_units = 0;
#endif

    // Tw: working transmittance, set to 0 when the pixel stops (T would fall below 1e-4) so that
    // everything behind blends with weight 0 without a per-splat "done" flag; T keeps the value
    // the pixel stopped at.  (Lanes outside the image start stopped.)
    float T = 1.f, Tw = inside ? 1.f : 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;              // (mode 2: depth_map_color)
    int last = 0;
    const unsigned long long alive0 = __ballot(inside);            // every pixel of the box that is in the image

    for (int base = start; base < end; base += FB) {
        if (wg_all(vote, Tw == 0.f, 4)) break;
        const int idx = base + tid;
        if (idx < end) {
            const int g = ids[idx];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * REC);
            float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
            if (mode == 1) {
                const float3 col = cmap_nonzero_lookup(p2.y, cmap_mm, cmap_lut);
                p1.z = col.x; p1.w = col.y; p2.x = col.z;
            }
            if (DC) {
                const float3 col = cmap_nonzero_lookup(p2.y, cmap_mm, cmap_lut);
                s_dc[tid] = make_float4(col.x, col.y, col.z, 0.f);
            }
            recs[tid].p0 = p0; recs[tid].p1 = p1; recs[tid].p2 = p2;
            s_mask[tid] = (unsigned char)block_mask(p0, p1, p2.z, org_x, org_y);
        }
        __syncthreads();
        const int cnt = min(FB, end - base);
        if (__all(Tw == 0.f)) continue;      // this wave is finished; keep meeting the barriers
        for (int c0 = 0; c0 < cnt; c0 += 64) {
            if (__all(Tw == 0.f)) break;
            if (ck_next < parts && base - start + c0 == ck_next * seg) {
                float* c5 = ck + (size_t)(ck_next - 1) * 5 * 256;
                c5[0] = T; c5[256] = a0; c5[512] = a1; c5[768] = a2; c5[1024] = a3;
                ++ck_next;
            }
            const int slot = c0 + lane;
            bool hit = slot < cnt && ((s_mask[slot] >> wb) & 1);
            {
                // Once pixels have stopped, only splats that reach a pixel that is still ALIVE matter.  In a tile
                // where densification piled up a thousand small splats the pile's own pixels stop early and the rest
                // of the pile reaches no one else -- yet the wave used to walk the whole list for the few pixels
                // beside it (the launch lasted as long as that one chain).  Test each slot's alpha >= 1/255 disc
                // against the bounding box of the alive pixels (the test of block_mask, on a smaller box):
                // drops only work that contributes exactly nothing.
                const unsigned long long alive = __ballot(Tw != 0.f);   // lane = (y << 3) | x of the 8x8 block
                if (alive != alive0) {
                    unsigned long long a = alive | (alive >> 32);
                    a |= a >> 16;
                    a |= a >> 8;
                    const unsigned cols = (unsigned)a & 0xffu;               // columns with an alive pixel
                    const int xl = __builtin_ctz(cols), xh = 31 - __builtin_clz(cols);
                    const int yl = (int)__builtin_ctzll(alive) >> 3, yh = (63 - (int)__builtin_clzll(alive)) >> 3;
                    if (hit) {
                        const BlockTest t = block_test(recs[slot].p0, recs[slot].p1, recs[slot].p2.z);
                        hit = box_hit(t, (float)(px0w + xl), (float)(px0w + xh), (float)(py0w + yl), (float)(py0w + yh));
                    }
                }
            }
            unsigned long long bits = __ballot(hit);
            // FWD_UNITS (4) hit splats per trip: their records are fetched and their alphas
            // evaluated together; only the T recurrence is serial.  The body is branch-free: a lane
            // that skips a splat contributes w = 0.  A wave issues one instruction at a time, so the
            // wave of a long list is bound by its own instruction count (a heavy workgroup left
            // ALONE on its CU still took ~380 cycles per splat): four per trip amortise the scalar
            // loop control (two: 52 us; four at 72 VGPRs, six workgroups per CU: 48 us; four at the
            // 64-VGPR budget of eight workgroups spill and gain nothing; requesting the next
            // records a trip ahead was slower: LDS returns in order, the wait covers them too).
            while (bits) {
                units += min((int)__popcll(bits), FWD_UNITS);
                int j[FWD_UNITS];            // missing splats of the last trip: the null record
#pragma unroll
                for (int u = 0; u < FWD_UNITS; ++u) {
                    j[u] = bits ? c0 + (int)__builtin_ctzll(bits) : FBL;
                    bits &= bits - 1;        // no-op when bits is already 0
                }
#ifdef GFL_TRACE
#pragma unroll
                for (int u = 0; u < FWD_UNITS; ++u) trace_units += j[u] != FBL ? 1 : 0;
#endif
                float4 q0[FWD_UNITS], q1[FWD_UNITS], q2[FWD_UNITS];
#pragma unroll
                for (int u = 0; u < FWD_UNITS; ++u) { q0[u] = recs[j[u]].p0; q1[u] = recs[j[u]].p1; q2[u] = recs[j[u]].p2; }
                float4 dq[DC ? FWD_UNITS : 1];
                if constexpr (DC) {
#pragma unroll
                    for (int u = 0; u < FWD_UNITS; ++u) dq[u] = s_dc[j[u]];
                }
                float al[FWD_UNITS];
                bool val[FWD_UNITS];
#pragma unroll
                for (int u = 0; u < FWD_UNITS; ++u) {
                    float G;
                    val[u] = splat_alpha2(q0[u], q1[u], fx, fy, al[u], G);
                }
#pragma unroll
                for (int u = 0; u < FWD_UNITS; ++u) {
                    const float a = val[u] ? al[u] : 0.f;
                    const float test_T = Tw * (1.f - a);
                    const bool stop = test_T < GFL_T_MIN;            // now, or earlier (Tw = 0)
                    const float w = stop ? 0.f : a * Tw;
                    a0 = fmaf(q1[u].z, w, a0); a1 = fmaf(q1[u].w, w, a1); a2 = fmaf(q2[u].x, w, a2); a3 = fmaf(q2[u].y, w, a3);
                    if constexpr (DC) { b0 = fmaf(dq[u].x, w, b0); b1 = fmaf(dq[u].y, w, b1); b2 = fmaf(dq[u].z, w, b2); }
                    T = stop ? T : test_T;
                    Tw = stop ? 0.f : test_T;
                    last = (val[u] && !stop) ? base - start + j[u] + 1 : last;
                }
                if (__all(Tw == 0.f)) break;
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;
        const float o0 = fmaf(T, bg, a0), o1 = fmaf(T, bg, a1), o2 = fmaf(T, bg, a2);
        out[pix] = o0;
        out[plane + pix] = o1;
        out[2 * plane + pix] = o2;
        out[3 * plane + pix] = fmaf(T, bg, a3);
        final_T[pix] = T;
        n_contrib[pix] = last;
        if (DC) {
            uint8_t* u0 = snap_u8 + pix * 3;
            u0[0] = img_u8(o0); u0[1] = img_u8(o1); u0[2] = img_u8(o2);
            uint8_t* u1 = snap_u8 + (plane + pix) * 3;
            u1[0] = img_u8(fmaf(T, bg, b0)); u1[1] = img_u8(fmaf(T, bg, b1)); u1[2] = img_u8(fmaf(T, bg, b2));
        }
    }
    if (FIT && lane == 0) atomicAdd(&tile_work[4 * tile + wb], units + 1);     // per 8x8 block (gfl_sched.hpp: block plan)
    // the wave stopped before the split position: every pixel's state is frozen, final = checkpoint
    for (; ck_next < parts; ++ck_next) {
        float* c5 = ck + (size_t)(ck_next - 1) * 5 * 256;
        c5[0] = T; c5[256] = a0; c5[512] = a1; c5[768] = a2; c5[1024] = a3;
    }
#ifdef GFL_TRACE
    const int trace_done = __popcll(__ballot(Tw == 0.f && inside));
    if (lane == 0 && tile < 16384) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        long long* tr = g_fwd_trace + (size_t)tile * 8;
        if (wave == 0) {
            tr[0] = trace_t0; tr[1] = wall_clock64();
            tr[2] = ((long long)(end - start) << 32) | (unsigned)(end - start);
            tr[3] = ((long long)(xcc & 15) << 32) | hw;
        }
        tr[4 + wave] = ((long long)trace_done << 32) | (unsigned)trace_units;
    }
#endif
  }
}
// ---- launchers (gfl_fit.hpp)
void launch_blend_fwd(const gfl_fit_state* st, float bg, int gx, int grid, float* out, float* final_T, int32_t* n_contrib,
                      const TileQueue& q, const FitWs& w, int mode, const unsigned* cmap_mm, const float* cmap_lut, int split_min,
                      hipStream_t s, uint8_t* snap_u8) {
    auto kern = mode == 0 ? fused_blend_fwd_kernel<0> : (mode == 1 ? fused_blend_fwd_kernel<1> :
                (mode == 2 ? fused_blend_fwd_kernel<2> : fused_blend_fwd_kernel<3>));
    // mode 3: one more workgroup per tile behind the blend's own (footprint_tile)
    const FootArgs foot = {st->foot_flags, st->keep, grid};
    const int T = gx * ((st->H + GFL_TILE - 1) / GFL_TILE);
    kern<<<mode == 3 ? grid + T : grid, 256, 0, s>>>(st->rec, st->ids, st->tile_range, bg, st->W, st->H, gx, inv_of(gx), out, final_T,
                              n_contrib, q, w.ckpt, cmap_mm, cmap_lut, split_min, w.sched_fwd.work, w.sched.first_slot,
                              mode != 1 ? w.stamp : nullptr, snap_u8, foot);
}

void launch_footprint(const gfl_fit_state* st, int gx, int T, hipStream_t s) {
    footprint_kernel<<<T, 256, 0, s>>>(st->rec, st->ids, st->tile_range, st->foot_flags, st->W, st->H, gx, st->keep);
}

#ifdef GFL_TRACE
int read_fwd_trace(long long* out, int n_tiles) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_trace), (size_t)n_tiles * 8 * sizeof(long long));
}
int read_fwd_trace2(long long* out, int n_values) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_trace2), (size_t)n_values * sizeof(long long));
}
#endif

}  // namespace gfl
