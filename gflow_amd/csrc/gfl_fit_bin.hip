// Fused fit iteration, stage 1: from the parameter rows to unsorted per-tile key lists.
//
// Binning without global atomics (measured: 258k L2 atomics cost 70-100 us) -- the EXACT path, the first iteration on a set
// of splats:
//   preprocess  : each 512-splat block counts its splat-tile pairs in an LDS histogram and writes the row hist[b][*];
//   colscan     : the columns become exclusive per-block bases, the tile totals tile_counts;
//   scatter     : each block re-walks its splats, ranks pairs with LDS atomics and writes keys[offset[t] + base[b][t] + rank];
//   tile sort   : per-tile bitonic sort of the unique 64-bit keys (gfl_tile_sort.hpp) -> the order is independent of the
//                 LDS-atomic arrival order.
// Every iteration that follows a full iteration takes "reserved tile regions" instead -- preprocess, column scan and scatter in
// ONE launch, with one returning global atomic per (block, tile): fused_preprocess_bin_kernel below.  (The 258k atomics above
// were one per PAIR on counters nobody had arranged; 118 blocks x 26 wave-level atomics on dense counters cost 2 us,
// tools/atomic_probe.hip.)
#include "gfl_fit_order.hpp"

namespace gfl {

// BINNED: the block's histogram stays in LDS and what the scatter needs of the splat comes back in `po` (PreOut, gfl_fit.hpp)
template <bool EWA_MFMA, bool PHASES, bool BINNED = false, int BLOCK = BIN_BLOCK>
__device__ __forceinline__ void preprocess_block(const PreArgs& a, const float4 (&row_v)[4], unsigned own_flags, int i,
                                                 int32_t* __restrict__ hist, PreOut* po = nullptr) {
    // op_mode (gfl_render_fwd): activated attributes in the rows, camera = the extrinsic in extr_out
    // scale_rows_mode != 0 (lambda_scale): count the rows the scale term averages over, per block
    const int N = a.N, W = a.W, H = a.H, gx = a.gx, gy = a.gy;
    const int T = gx * gy;
    const Cam c = a.op_mode ? load_cam(a.intr, a.extr_out) : cam_from_pose(a.intr, a.pose);
    if (!a.op_mode && blockIdx.x == 0 && threadIdx.x == 0) {
        float* extr_out = a.extr_out;
        extr_out[0] = c.r00; extr_out[1] = c.r01; extr_out[2] = c.r02; extr_out[3] = c.t0;
        extr_out[4] = c.r10; extr_out[5] = c.r11; extr_out[6] = c.r12; extr_out[7] = c.t1;
        extr_out[8] = c.r20; extr_out[9] = c.r21; extr_out[10] = c.r22; extr_out[11] = c.t2;
    }
    float u = 0.f, v = 0.f, cutoff = 0.f;
    int wx0 = 0, wy0 = 0, wnx = 0, wnt = 0;      // rectangle of a "wide" splat (walked by the wave below)
    bool in_scale_rows = false;
    Splat s = {};
    Proj p = {};
    float cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < N) {
        s = splat_from_row(row_v[0], row_v[1], row_v[2], row_v[3], a.op_mode != 0);
#ifdef GFL_TRACE
        if (PHASES) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GFL_PHASE(0, 2);
        }
#endif
        p = project_fwd(c, s.x, s.y, s.z, W, H, a.nearest, a.extent);
        if (p.vis) cov3d_fwd(s.s, s.q, cov);
    }
    if (PHASES) GFL_PHASE(0, 3);
    Ewa e = {};
    if (EWA_MFMA) e = ewa_fwd_mfma(c, p.vis, p.px, p.py, p.pz, cov, W, H);       // (the whole wave: no divergence here)
    if (i < N) {
        float depth = 0.f, A = 0.f, B = 0.f, C = 0.f;
        int rad = 0;
        if (p.vis) {
            u = p.u; v = p.v; depth = p.pz;
            if (!EWA_MFMA) e = ewa_fwd(c, p.px, p.py, p.pz, cov, W, H);
            if (e.ok) {
                const int r = ewa_radius(e);
                int x0, x1, y0, y1;
                tile_rect(u, v, r, gx, gy, x0, x1, y0, y1);
                const int nt = (x1 - x0) * (y1 - y0);
                if (nt > 0) {
                    rad = r;
                    A = e.c / e.det; B = -e.b / e.det; C = e.a / e.det;
                    cutoff = alpha_cutoff(s.o, e.lam);
                    if (nt > WIDE_TILES) {
                        wx0 = x0; wy0 = y0; wnx = x1 - x0; wnt = nt;
                    } else {
                        for (int ty = y0; ty < y1; ++ty)
                            for (int tx = x0; tx < x1; ++tx)
                                if (tile_hit2(u, v, cutoff, tx, ty)) atomicAdd(&hist[ty * gx + tx], 1);
                    }
                }
            }
        }
        float4* r4 = reinterpret_cast<float4*>(a.rec + (size_t)i * REC);
        r4[0] = make_float4(u, v, A, B);
        r4[1] = make_float4(C, s.o, s.c[0], s.c[1]);
        r4[2] = make_float4(s.c[2], depth, cutoff, __int_as_float(rad));
        if (BINNED) { po->u = u; po->v = v; po->cutoff = cutoff; po->depth = depth; po->rad = rad; }
        in_scale_rows = a.scale_rows_mode && scale_row(u, v, W, H, own_flags, a.scale_rows_mode);
        if (wnt > SLOT_MAX) {
            // too many tiles for the splat's own SLOT_MAX pair rows: a run of wnt rows behind them (FitWs.wide_off)
            const int off = atomicAdd(a.pool_counter, wnt);
            a.wide_off[i] = off + wnt <= a.pool_cap ? off : -1;      // (-1: no rows -- and the lists' overflow flag: they must grow)
            if (off + wnt > a.pool_cap) *a.overflow = 1;
        }
    }
    if (PHASES) GFL_PHASE(0, 4);
    {
        // splats covering many tiles: the whole wave counts their tiles, 64 at a time
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(wnt > 0);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const float su = __shfl(u, src), sv = __shfl(v, src), sc = __shfl(cutoff, src);
            const int sx0 = __shfl(wx0, src), sy0 = __shfl(wy0, src), snx = __shfl(wnx, src), snt = __shfl(wnt, src);
            for (int q = lane; q < snt; q += 64) {
                const int tx = sx0 + q % snx, ty = sy0 + q / snx;
                if (tile_hit2(su, sv, sc, tx, ty)) atomicAdd(&hist[ty * gx + tx], 1);
            }
        }
    }
    if (a.scale_rows_mode) {
        // (no atomics on global memory: one partial per block, folded by every block of the backward kernel)
        const int wcnt = __popcll(__ballot(in_scale_rows));
        // (scale_cnt: one partial per 256 splats -- a 512-splat block leaves its count in the first of its two)
        __shared__ int32_t s_cnt[BLOCK / 64];
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = wcnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < BLOCK / 64; ++w) tot += s_cnt[w];
            if (BLOCK == 512) { a.scale_cnt[2 * blockIdx.x] = tot; a.scale_cnt[2 * blockIdx.x + 1] = 0; }
            else a.scale_cnt[blockIdx.x] = tot;
        }
    }
    if (PHASES) GFL_PHASE(0, 5);
    __syncthreads();
    if (PHASES) GFL_PHASE(0, 6);
    if (BINNED) return;
    int32_t* row = a.hist_g + (size_t)blockIdx.x * T;
    for (int t = threadIdx.x; t < T; t += BIN_BLOCK) row[t] = hist[t];
    if (PHASES) GFL_PHASE(0, 7);
}

template <bool EWA_MFMA>
__global__ void __launch_bounds__(BIN_BLOCK) fused_preprocess_fwd_kernel(const float* __restrict__ params, PreArgs a,
                                                                         const uint8_t* __restrict__ row_flags) {
    extern __shared__ int32_t hist[];
    const int T = a.gx * a.gy;
    GFL_PHASE(0, 0);
    // the splat's row first: its latency (1.1 us of the launch's 8, tools/phase_trace.py) then overlaps the clearing of
    // the histogram, the barrier and the camera's scalar loads instead of following them
    const int i = blockIdx.x * BIN_BLOCK + threadIdx.x;
    float4 row_v[4] = {};
    unsigned own_flags = 0;
    if (i < a.N) {
        const float4* prow = reinterpret_cast<const float4*>(params + (size_t)i * ROW);
#pragma unroll
        for (int q = 0; q < 4; ++q) row_v[q] = prow[q];
        if (a.scale_rows_mode && row_flags) own_flags = row_flags[i];
    }
    for (int t = threadIdx.x; t < T; t += BIN_BLOCK) hist[t] = 0;
    __syncthreads();
    GFL_PHASE(0, 1);
    preprocess_block<EWA_MFMA, true>(a, row_v, own_flags, i, hist);
}
// ------------------------------------------------------------------ reserved tile regions (round 4)
// Preprocess + binning in ONE launch, for an iteration that follows another full iteration: the histogram rows, the column
// scan and the scatter launch exist because a key's position in its tile's list needs every block's count of every tile --
// a dependency across the whole launch.  But the lists of iteration i + 1 are the lists of iteration i but for one Adam step:
// at the END of iteration i one workgroup of the per-splat launch (build_sort_order<.., true>) gives every tile a REGION of
// the key array sized by what the tile holds now plus a margin (region_cap), and the next iteration's blocks reserve their
// part of it with one returning atomicAdd per (block, tile with keys): fill[position] += the block's count (dense 4-byte counters:
// 118 blocks x 1 620 tiles cost 2.1 us on top of the launch, tools/atomic_probe.hip; counters a cache line apart cost 7).
// The order inside a region is whatever order the blocks arrived in -- the tile sort, which follows anyway, makes the lists
// what the exact path's are (keys are unique: depth bits | splat id), so ids / tile ranges / everything downstream is
// bit-identical but for the gaps between the lists.  A tile that outgrows its region voids the iteration: its surplus keys are
// not written, nothing is stepped and the iteration is counted in overflow[1] (like K_cap overflow, but not sticky: the
// regions reserved at the end of the void iteration are sized by what the tiles WANTED, so the next iteration fits), and the
// host runs one more iteration for each (FitEngine.settle_overflow).  Launches per iteration: 8 -> 6.
template <bool EWA_MFMA>
__global__ void __launch_bounds__(RBIN_BLOCK) fused_preprocess_bin_kernel(const float* __restrict__ params, PreArgs a,
                                                                          const uint8_t* __restrict__ row_flags, BinArgs b) {
    constexpr int BLOCK = RBIN_BLOCK;
    extern __shared__ int32_t hist[];               // [T] counts, then cursors; [T] limits behind them; 64 idle cursors
    const int T = a.gx * a.gy;
    int32_t* lim = hist + T;
    const int tid = threadIdx.x;
    GFL_PHASE(1, 0);                                 // (the column scan's row of the phase trace: it does not run here)
    const int i = blockIdx.x * BLOCK + tid;
    float4 row_v[4] = {};
    unsigned own_flags = 0;
    if (i < a.N) {
        const float4* prow = reinterpret_cast<const float4*>(params + (size_t)i * ROW);
#pragma unroll
        for (int q = 0; q < 4; ++q) row_v[q] = prow[q];
        if (a.scale_rows_mode && row_flags) own_flags = row_flags[i];
    }
    // this lane's tiles' regions: requested here, used after the preprocess
    constexpr int PER_MAX = 4096 / BLOCK;            // (T <= 4096: fit_reserved_ok)
    int4 reg[PER_MAX];
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) reg[k] = tid + k * BLOCK < T ? b.region[tid + k * BLOCK] : make_int4(0, 0, 0, 0);
    if (blockIdx.x == 0 && tid == 0) {
        if (*b.regions_valid == 0) *a.overflow = 2;      // the host asked for regions nobody has reserved
        *b.regions_valid = 0;
        *b.extent = *b.extent_next;
    }
    if (blockIdx.x == 0)
        for (int c = tid; c < b.n_pull; c += BLOCK) b.pull_counters[c] = 0;
    for (int t = tid; t < T; t += BLOCK) hist[t] = 0;
    __syncthreads();
    GFL_PHASE(1, 1);
    PreOut o = {0.f, 0.f, 0.f, 0.f, 0};
    preprocess_block<EWA_MFMA, false, true, BLOCK>(a, row_v, own_flags, i, hist, &o);      // (ends behind a barrier)
    GFL_PHASE(1, 2);
    // ---- this block's part of every tile's region
    int got[PER_MAX], cnt[PER_MAX];
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int t = tid + k * BLOCK;
        cnt[k] = t < T ? hist[t] : 0;
        got[k] = cnt[k] > 0 ? atomicAdd(&b.fill[reg[k].z], cnt[k]) : 0;
    }
    bool over = false, over_cap = false;
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int t = tid + k * BLOCK;
        if (t < T) {
            hist[t] = reg[k].x + got[k];
            lim[t] = min(reg[k].x + reg[k].y, b.K_cap);
            over |= got[k] + cnt[k] > reg[k].y;
            over_cap |= reg[k].x + min(got[k] + cnt[k], reg[k].y) > b.K_cap;
        }
    }
    // (a region cut short by K_cap is the lists' overflow: sticky, the lists have to grow; a tile that outgrew its region voids
    //  THIS iteration only -- overflow[3], which the tile sort's launch moves to overflow[2] where the update launches look)
    if (over_cap) *a.overflow = 1;
    else if (over) a.overflow[3] = 1;
    GFL_PHASE(1, 3);
    __syncthreads();
    GFL_PHASE(1, 4);
    // ---- keys (the scatter launch's walk, with the cursors above)
    const float u = o.u, v = o.v, cutoff = o.cutoff, depth = o.depth;
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    if (o.rad > 0) tile_rect(u, v, o.rad, a.gx, a.gy, x0, x1, y0, y1);
    const int gx = a.gx, nx = x1 - x0, nt = nx * (y1 - y0);
    const bool wide = nt > WIDE_TILES;
    if (nt > 0 && !wide) {
        // Four tiles per trip: the cursor's LDS add returns the key's position, and a lane that waits for one add per trip
        // spends the walk waiting (4.3 us of the launch's 17, tools/phase_trace.py; two waves per SIMD hide nothing).  Lanes
        // without a hit add to a cursor of their own behind the limits, so that the four adds are straight-line code.
        const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned long long)(unsigned)i;
        const int dummy = 2 * T + (tid & 63);
        int cx = 0, cy = 0;
        for (int q0 = 0; q0 < nt; q0 += 4) {
            int t4[4];
            bool h4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tx = x0 + cx, ty = y0 + cy;
                h4[e] = q0 + e < nt && tile_hit2(u, v, cutoff, tx, ty);
                t4[e] = ty * gx + tx;
                if (++cx == nx) { cx = 0; ++cy; }
            }
            int pos4[4], lim4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pos4[e] = atomicAdd(&hist[h4[e] ? t4[e] : dummy], 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) lim4[e] = lim[h4[e] ? t4[e] : 0];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (h4[e] && pos4[e] < lim4[e]) __hip_atomic_store(&b.keys[pos4[e]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    GFL_PHASE(1, 5);
    const int lane = tid & 63;
    unsigned long long todo = __ballot(wide);
    while (todo) {
        const int src = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        const float su = __shfl(u, src), sv = __shfl(v, src), sc = __shfl(cutoff, src);
        const int sx0 = __shfl(x0, src), sy0 = __shfl(y0, src), snx = __shfl(nx, src), snt = __shfl(nt, src);
        const unsigned long long key =
            ((unsigned long long)__float_as_uint(__shfl(depth, src)) << 32) | (unsigned long long)(unsigned)__shfl(i, src);
        for (int q = lane; q < snt; q += 64) {
            const int tx = sx0 + q % snx, ty = sy0 + q / snx;
            if (!tile_hit2(su, sv, sc, tx, ty)) continue;
            const int t = ty * gx + tx;
            const int pos = atomicAdd(&hist[t], 1);
            if (pos < lim[t]) __hip_atomic_store(&b.keys[pos], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    GFL_PHASE(1, 6);
}

// Columns of hist -> exclusive per-block bases (in place) and per-tile totals.
// 32 tiles per workgroup, eight row groups per tile; loads are issued up to 24 at a time before any
// store so that they overlap (an in-place load/store chain serialises on the L2 latency:
// measured 64 us for 118 rows in the first version of this kernel).  With four row groups and chunks of 16 a
// 67 000-splat frame (131 rows: 33 per group) needed three dependent chunks per pass and took 11 us instead of 5:
// eight groups x 24 rows cover 192 rows (98 000 splats) with ONE round trip per pass.
constexpr int CS_CHUNK = 24;
constexpr int CS_TILES = 32;
constexpr int CS_GROUPS = 8;
__global__ void __launch_bounds__(256) bin_colscan_kernel(int32_t* __restrict__ hist_g, int nblk, int T,
                                                         int32_t* __restrict__ tile_counts,
                                                         int32_t* __restrict__ pool_counter,
                                                         int32_t* __restrict__ pull_counters, int n_pull,
                                                         int32_t* __restrict__ overflow) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *pool_counter = 0;   // preprocess is done with it
        overflow[2] = 0; overflow[3] = 0;      // (no reserved regions in this iteration: nothing can outgrow one)
    }
    // the pull counters of this iteration's two blend launches (the tile queues themselves may be older: they are
    // rebuilt at the END of an iteration, beside the per-splat launch)
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < n_pull; c += 256) pull_counters[c] = 0;
    __shared__ int32_t gsum[CS_GROUPS][CS_TILES];
    GFL_PHASE(1, 0);
    const int tl = threadIdx.x % CS_TILES, rg = threadIdx.x / CS_TILES;
    const int t = blockIdx.x * CS_TILES + tl;
    const int R = (nblk + CS_GROUPS - 1) / CS_GROUPS;
    const int b0 = rg * R, b1 = min(nblk, b0 + R);
    int total = 0;
    if (t < T) {
        for (int b = b0; b < b1; b += CS_CHUNK) {
            int v[CS_CHUNK];
#pragma unroll
            for (int k = 0; k < CS_CHUNK; ++k) v[k] = (b + k < b1) ? hist_g[(size_t)(b + k) * T + t] : 0;
#pragma unroll
            for (int k = 0; k < CS_CHUNK; ++k) total += v[k];
        }
    }
    GFL_PHASE(1, 1);
    gsum[rg][tl] = total;
    __syncthreads();
    GFL_PHASE(1, 2);
    int run = 0, all = 0;
#pragma unroll
    for (int g = 0; g < CS_GROUPS; ++g) {
        const int x = gsum[g][tl];
        run += g < rg ? x : 0;
        all += x;
    }
    if (t < T) {
        if (rg == 0) tile_counts[t] = all;
        for (int b = b0; b < b1; b += CS_CHUNK) {
            int v[CS_CHUNK];
#pragma unroll
            for (int k = 0; k < CS_CHUNK; ++k) v[k] = (b + k < b1) ? hist_g[(size_t)(b + k) * T + t] : 0;
#pragma unroll
            for (int k = 0; k < CS_CHUNK; ++k) {
                if (b + k < b1) hist_g[(size_t)(b + k) * T + t] = run;
                run += v[k];
            }
        }
    }
    GFL_PHASE(1, 3);
}
__global__ void __launch_bounds__(BIN_BLOCK) fused_scatter_kernel(const float* __restrict__ rec, int N, int gx, int gy,
                                                                  const int32_t* __restrict__ hist_g,
                                                                  const int32_t* __restrict__ tile_counts,
                                                                  int32_t* __restrict__ tile_offsets, int K_cap,
                                                                  unsigned long long* __restrict__ keys,
                                                                  int32_t* __restrict__ overflow,
                                                                  Sched sched_bwd, Sched sched_fwd,
                                                                  const int32_t* __restrict__ sched_valid,
                                                                  int4* __restrict__ sort_order,
                                                                  int32_t* __restrict__ extent) {
    extern __shared__ int32_t cursor[];
    __shared__ int32_t wsum[BIN_BLOCK / 64];
    const int T = gx * gy;
    if (sort_order && blockIdx.x == gridDim.x - 3) {     // (a workgroup of its own: in workgroup 0 it lengthened the launch)
        build_sort_order<BIN_BLOCK, false>(tile_counts, T, sort_order, wsum);
        return;
    }
    if (blockIdx.x >= gridDim.x - 2) {
        // Two extra workgroups build the blend kernels' tile queues -- but only while there is no schedule yet: from
        // the first backward on, the queues of iteration i + 1 are built at the END of iteration i, by two extra
        // workgroups of the per-splat launch (they need nothing but the work the blend kernels of iteration i counted).
        // In here they set the duration of the whole launch: 16-18 us against the scatter's own 12.
        if (*sched_valid) return;
        __shared__ SchedLds sched_lds;
        const Sched sc = blockIdx.x == gridDim.x - 1 ? sched_bwd : sched_fwd;
        uint32_t* frac4 = T <= SCHED_PLAN_TILES ? reinterpret_cast<uint32_t*>(cursor + T) : nullptr;      // (sched_dyn_lds)
        if (sched_xcd_usable(sc, T, SCHED_BLOCK)) schedule_tiles_xcd<SCHED_BLOCK>(tile_counts, T, sc, cursor, wsum, sched_lds, frac4);
        else schedule_tiles(tile_counts, T, sc, cursor, wsum, sched_lds, frac4);
        return;
    }
    const int32_t* base_row = hist_g + (size_t)blockIdx.x * T;
    GFL_PHASE(2, 0);
    // The launch has about one wave per SIMD: a load that is issued where its value is needed costs a full round trip
    // with nothing to hide it (tools/phase_trace.py: tile totals 0.9 us, then this block's bases 1.3 us, then the splat
    // record 0.5 us, one after the other).  So everything a lane will need is requested here, together.
    const int i = blockIdx.x * BIN_BLOCK + threadIdx.x;
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p2 = p0;
    if (i < N) {
        const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)i * REC);
        p0 = r4[0]; p2 = r4[2];
    }
    {
        // every block scans the T tile totals itself (a few elements per thread); block 0
        // publishes the exclusive offsets for the tile sort / blend kernels
        const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
        const int per = (T + BIN_BLOCK - 1) / BIN_BLOCK;
        const int t0 = tid * per;
        constexpr int PER_MAX = 8;                   // tiles per lane held in registers (T <= 4096)
        int cnt[PER_MAX], bas[PER_MAX];
        int local = 0;
        if (per <= PER_MAX) {
#pragma unroll
            for (int k = 0; k < PER_MAX; ++k) {
                const bool ok = k < per && t0 + k < T;
                cnt[k] = ok ? tile_counts[t0 + k] : 0;
                bas[k] = ok ? base_row[t0 + k] : 0;
            }
#pragma unroll
            for (int k = 0; k < PER_MAX; ++k) local += cnt[k];
        } else {
            for (int k = 0; k < per; ++k)
                if (t0 + k < T) local += tile_counts[t0 + k];
        }
        int sc = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(sc, off);
            if (lane >= off) sc += n;
        }
        if (lane == 63) wsum[wid] = sc;
        GFL_PHASE(2, 1);
        __syncthreads();
        int wprefix = 0;
        for (int w = 0; w < wid; ++w) wprefix += wsum[w];
        int run = wprefix + sc - local;
        if (per <= PER_MAX) {
#pragma unroll
            for (int k = 0; k < PER_MAX; ++k) {
                const int t = t0 + k;
                if (k < per && t < T) {
                    cursor[t] = run + bas[k];
                    if (blockIdx.x == 0) tile_offsets[t] = run;
                    run += cnt[k];
                }
            }
        } else {
            for (int k = 0; k < per; ++k) {
                const int t = t0 + k;
                if (t < T) {
                    cursor[t] = run + base_row[t];
                    if (blockIdx.x == 0) tile_offsets[t] = run;
                    run += tile_counts[t];
                }
            }
        }
        if (blockIdx.x == 0 && tid == BIN_BLOCK - 1) {
            tile_offsets[T] = run;
            if (extent) *extent = run;           // (the lists are gap-free here: extent = pairs)
        }
    }
    GFL_PHASE(2, 2);
    __syncthreads();
    GFL_PHASE(2, 3);
    float u = 0.f, v = 0.f, cutoff = 0.f, depth = 0.f;
    int rad = 0;
    if (i < N) {
        rad = __float_as_int(p2.w);
        u = p0.x; v = p0.y; cutoff = p2.z; depth = p2.y;
    }
#ifdef GFL_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GFL_PHASE(2, 4);
#endif
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    if (rad > 0) tile_rect(u, v, rad, gx, gy, x0, x1, y0, y1);
    const int nx = x1 - x0, nt = nx * (y1 - y0);
    // a splat covering many tiles would keep its lane (and so its wave) busy for ~100 trips:
    // such splats are walked by the whole wave instead, 64 tiles at a time
    const bool wide = nt > WIDE_TILES;
    if (nt > 0 && !wide) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned long long)(unsigned)i;
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx) {
                if (!tile_hit2(u, v, cutoff, tx, ty)) continue;
                const int pos = atomicAdd(&cursor[ty * gx + tx], 1);
                // scattered 8-byte stores: write-through (sc1)
                if (pos < K_cap) __hip_atomic_store(&keys[pos], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *overflow = 1;
            }
    }
    GFL_PHASE(2, 5);
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(wide);
    while (todo) {
        const int src = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        const float su = __shfl(u, src), sv = __shfl(v, src), sc = __shfl(cutoff, src);
        const int sx0 = __shfl(x0, src), sy0 = __shfl(y0, src), snx = __shfl(nx, src), snt = __shfl(nt, src);
        const unsigned long long key =
            ((unsigned long long)__float_as_uint(__shfl(depth, src)) << 32) | (unsigned long long)(unsigned)__shfl(i, src);
        for (int q = lane; q < snt; q += 64) {
            const int tx = sx0 + q % snx, ty = sy0 + q / snx;
            if (!tile_hit2(su, sv, sc, tx, ty)) continue;
            const int pos = atomicAdd(&cursor[ty * gx + tx], 1);
            if (pos < K_cap) __hip_atomic_store(&keys[pos], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *overflow = 1;
        }
    }
    GFL_PHASE(2, 6);
}

// ---- launchers (gfl_fit.hpp)
void launch_preprocess_fwd(const float* params, const PreArgs& a, const uint8_t* row_flags, int nblk, bool mfma, hipStream_t s) {
    const size_t lds = (size_t)a.gx * a.gy * sizeof(int32_t);
    auto kern = mfma ? fused_preprocess_fwd_kernel<true> : fused_preprocess_fwd_kernel<false>;
    kern<<<nblk, BIN_BLOCK, lds, s>>>(params, a, row_flags);
}

void launch_preprocess_bin(const float* params, const PreArgs& a, const uint8_t* row_flags, const BinArgs& b, int nblk, bool mfma,
                           hipStream_t s) {
    const size_t lds = (size_t)a.gx * a.gy * sizeof(int32_t);
    auto kern = mfma ? fused_preprocess_bin_kernel<true> : fused_preprocess_bin_kernel<false>;
    (void)nblk;      // (the exact path's blocks; this launch has its own block size)
    kern<<<(a.N + RBIN_BLOCK - 1) / RBIN_BLOCK, RBIN_BLOCK, 2 * lds + 64 * sizeof(int32_t), s>>>(params, a, row_flags, b);
}

void launch_colscan(const FitWs& w, int nblk, int T, int32_t* overflow, hipStream_t s) {
    bin_colscan_kernel<<<(T + CS_TILES - 1) / CS_TILES, 256, 0, s>>>(w.hist, nblk, T, w.tile_counts, w.pool_counter,
                                                                     w.sched.counters, 2 * w.sched.nq, overflow);
}

void launch_scatter(const gfl_fit_state* st, const FitWs& w, int nblk, int gx, int gy, bool ordered, hipStream_t s) {
    fused_scatter_kernel<<<nblk + 2 + (ordered ? 1 : 0), BIN_BLOCK, sched_dyn_lds(gx * gy), s>>>(
        st->rec, st->N, gx, gy, w.hist, w.tile_counts, st->tile_offsets, st->K_cap, w.keys, st->overflow, w.sched, w.sched_fwd,
        w.sched_valid, ordered ? w.sort_order : nullptr, w.extent);
}

#ifdef GFL_TRACE
int read_phase_trace_bin(long long* out, int n_values) {     // rows 0-2 (preprocess, column scan / binning, scatter)
    const int n = n_values < 3 * PHASE_WAVES * 8 ? n_values : 3 * PHASE_WAVES * 8;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_trace), (size_t)n * sizeof(long long));
}
#endif

}  // namespace gfl
