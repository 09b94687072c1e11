// Fused fit iteration: the whole of gflow/trainer.py:387-558 (activations, render,
// losses, backward, gradient masking, Adam) as ~10 kernel launches issued by ONE
// library call, with no host read-back and every launch sized by N, T or the image
// (never by the data-dependent pair count K) -> graph-capturable.
//
// Data layout in HBM (288 GB: capacity-based, nothing is reallocated when N grows)
//   params / adam_m / adam_v : [cap][16] f32, one 64-byte row per splat
//        x y z | sx sy sz | qw qx qy qz | opacity | r g b | pad pad      (raw values)
//   rec   : [cap][12] f32, what a pixel needs from a splat (3 x 16-byte loads)
//        u v A B | C opacity r g | b depth cutoff radius(int bits)
//   d_rec : [cap][12] f32, gradient of the loss wrt the first 10 entries of rec
//   keys  : [K_cap] u64 (depth bits << 32 | id), ids : [K_cap] i32, tile_range [T][2]
//   hist  : [n_bin_blocks][T] i32 per-block tile histogram -> per-block base offsets
//
// Binning without global atomics (measured: 258k L2 atomics cost 70-100 us):
//   preprocess  : each 512-splat block counts its splat-tile pairs in an LDS histogram
//                 and writes the row hist[b][*];
//   colscan     : one workgroup turns the columns into exclusive per-block bases and the
//                 tile totals into tile_offsets (exclusive scan);
//   scatter     : each block re-walks its splats, ranks pairs with LDS atomics and writes
//                 keys[tile_offsets[t] + base[b][t] + rank];
//   tile sort   : per-tile bitonic sort of the unique 64-bit keys (gfl_bin.hip) ->
//                 order is independent of the LDS-atomic arrival order.
// That is the EXACT path: the first iteration on a set of splats.  Every iteration that follows a full iteration (round 4)
// takes "reserved tile regions" instead -- preprocess, column scan and scatter in ONE launch, with one returning global
// atomic per (block, tile): see fused_preprocess_bin_kernel below.  (The 258k atomics above were one per PAIR on counters
// nobody had arranged; 118 blocks x 26 wave-level atomics on dense counters cost 2 us, tools/atomic_probe.hip.)
#include "gfl_math.hpp"
#include "gfl_profile.hpp"
#include "gfl_sched.hpp"

#include <stdlib.h>

namespace gfl {

constexpr int BIN_BLOCK = 512;
#ifndef GFL_WIDE_TILES
#define GFL_WIDE_TILES 16
#endif
constexpr int WIDE_TILES = GFL_WIDE_TILES;   // splats covering more tiles are binned by a whole wave
constexpr int ROW = 16;   // floats per params row
constexpr int REC = 12;   // floats per rec row
#ifndef GFL_PG_STRIDE
#define GFL_PG_STRIDE 12
#endif
constexpr int PG = GFL_PG_STRIDE;   // floats per pair_grad row (12 live; 16 = one 64-byte sector per row)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// analysis build only (make TRACE=1, tools/phase_trace.py): time stamps of the phases of the latency-bound launches,
// one row of eight per wave.  kernel 0 = preprocess, 1 = column scan, 2 = scatter, 3 = per-splat backward + Adam
#ifdef GFL_TRACE
constexpr int PHASE_WAVES = 4096;
__device__ long long g_phase_trace[4 * PHASE_WAVES * 8];
#define GFL_PHASE(kernel, slot)                                                                                      \
    do {                                                                                                             \
        const int w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                                          \
        if ((threadIdx.x & 63) == 0 && w_ < PHASE_WAVES) g_phase_trace[((kernel) * PHASE_WAVES + w_) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define GFL_PHASE(kernel, slot) do {} while (0)
#endif

// pose [qx,qy,qz,qw,tx,ty,tz] -> camera (trainer.py:115-121)
__device__ __forceinline__ Cam cam_from_pose(const float* __restrict__ intr, const float* __restrict__ pose) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel this is inlined into, and the oracle's arithmetic)
    Cam c;
    c.fx = intr[0]; c.fy = intr[1]; c.cx = intr[2]; c.cy = intr[3];
    float x = pose[0], y = pose[1], z = pose[2], w = pose[3];
    const float inv = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
    x *= inv; y *= inv; z *= inv; w *= inv;
    c.r00 = 1.f - 2.f * (y * y + z * z); c.r01 = 2.f * (x * y - w * z); c.r02 = 2.f * (x * z + w * y);
    c.r10 = 2.f * (x * y + w * z); c.r11 = 1.f - 2.f * (x * x + z * z); c.r12 = 2.f * (y * z - w * x);
    c.r20 = 2.f * (x * z - w * y); c.r21 = 2.f * (y * z + w * x); c.r22 = 1.f - 2.f * (x * x + y * y);
    c.t0 = pose[4]; c.t1 = pose[5]; c.t2 = pose[6];
    return c;
}

struct Splat {           // activated parameters of one splat
    float x, y, z;
    float s[3], raw_s[3];
    float q[4], raw_q[4], qn;
    float o, c[3];
};

// activated: the row already holds what the rasteriser consumes (scale, unit quaternion, opacity, colour) --
// the differentiable operator gfl_render_*, whose caller applies GFlow's activations in PyTorch (render.py:6-20)
__device__ __forceinline__ Splat splat_from_row(const float4& a, const float4& b, const float4& c, const float4& d,
                                                bool activated = false) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel this is inlined into, and the oracle's arithmetic)
    Splat s;
    s.x = a.x; s.y = a.y; s.z = a.z;
    s.raw_s[0] = a.w; s.raw_s[1] = b.x; s.raw_s[2] = b.y;
    s.raw_q[0] = b.z; s.raw_q[1] = b.w; s.raw_q[2] = c.x; s.raw_q[3] = c.y;
    if (activated) {
#pragma unroll
        for (int k = 0; k < 3; ++k) s.s[k] = s.raw_s[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) s.q[k] = s.raw_q[k];
        s.qn = 1.f;
        s.o = c.z;
        s.c[0] = c.w; s.c[1] = d.x; s.c[2] = d.y;
        return s;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) s.s[k] = fabsf(s.raw_s[k]);                       // trainer.py:65
    s.qn = fmaxf(sqrtf(s.raw_q[0] * s.raw_q[0] + s.raw_q[1] * s.raw_q[1] + s.raw_q[2] * s.raw_q[2] +
                       s.raw_q[3] * s.raw_q[3]), 1e-12f);                         // F.normalize, trainer.py:66
#pragma unroll
    for (int k = 0; k < 4; ++k) s.q[k] = s.raw_q[k] / s.qn;
    s.o = sigmoidf_(10.0f * c.z);                                                 // trainer.py:58-59,67
    s.c[0] = sigmoidf_(c.w); s.c[1] = sigmoidf_(d.x); s.c[2] = sigmoidf_(d.y);    // trainer.py:68
    return s;
}

// squared radius of the disc outside which alpha < 1/255 for every pixel, with a
// safety margin so that a culled (splat, tile) pair is skipped by the blend as well
__device__ __forceinline__ float alpha_cutoff(float o, float lam) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel this is inlined into, and the oracle's arithmetic)
    if (o < GFL_ALPHA_MIN) return -1.0f;                 // never visible
    const float r = 255.0f * o;
    if (r < 1.05f) return 3.0e38f;                       // too close to the threshold: no culling
    return 2.0f * __logf(r) * lam * 1.002f + 0.01f;
}

__device__ __forceinline__ bool tile_hit2(float u, float v, float cutoff, int tx, int ty) {
#pragma clang fp contract(off)     // (no fused multiply-adds: the same bits in every kernel this is inlined into, and the oracle's arithmetic)
    const float x_lo = (float)(tx * GFL_TILE), x_hi = x_lo + (float)(GFL_TILE - 1);
    const float y_lo = (float)(ty * GFL_TILE), y_hi = y_lo + (float)(GFL_TILE - 1);
    const float ddx = fmaxf(fmaxf(x_lo - u, u - x_hi), 0.f);
    const float ddy = fmaxf(fmaxf(y_lo - v, v - y_hi), 0.f);
    return ddx * ddx + ddy * ddy <= cutoff;
}

// rows of the scale term (trainer.py:495-502): the reference's `within_index` ALIASES valid_uv_index, which is
// narrowed in place to the still (camera-only stage) / moving (joint stage) rows at trainer.py:467-471, so
// scale and 1/depth are both taken over: inside the image AND (unlabelled OR still / moving by stage).
// flags: bit0 = still, bit1 = the row has a still/moving label;  mode: 1 = joint stage, 2 = camera-only stage
__device__ __forceinline__ bool scale_row(float u, float v, int W, int H, unsigned flags, int mode) {
    const bool within = u > 0.f && u < (float)(W - 1) && v > 0.f && v < (float)(H - 1);
    const bool labelled = flags & 2u, still = flags & 1u;
    return within && (!labelled || (mode == 2 ? still : !still));
}

// ------------------------------------------------------------------ preprocess fwd
// What one block of splats does for the forward pass -- activations, projection, covariance, EWA, the record a pixel
// needs, the block's row of the tile histogram -- once the splats' rows are in registers and the LDS histogram is cleared
// (and a barrier has passed).  Two callers: the stand-alone launch (fused_preprocess_fwd_kernel) and the TAIL of the
// per-splat backward + Adam launch of the iteration before (round 4, "next preprocess"): there the row is still in
// registers when Adam has stepped it, and while the camera cannot move the next iteration's preprocess needs nothing else.
// PARAMS of a block: splats [blockIdx.x * BLOCK, +BLOCK); BLOCK must be BIN_BLOCK (the scatter re-walks the same blocks).
// EWA_MFMA: the J Sigma J^T contraction on the matrix cores (cov2d_mfma, gfl_math.hpp) instead of 30 FMAs in the lane
// -- the variant north_star names; selected with GFL_EWA_MFMA=1, measured in DESIGN.md section 4, off by default.
struct PreArgs {
    const float* intr; const float* pose;
    int N, W, H;
    float nearest, extent;
    int gx, gy;
    float* rec; int32_t* slot_inv; int32_t* hist_g; float* extr_out; int32_t* overflow;
    int32_t* slot_pool; int32_t* pool_counter; int pool_cap; int op_mode;
    int scale_rows_mode; int32_t* scale_cnt;
    int32_t* pre_valid;              // set by a tail that ran the preprocess, checked and cleared by the column scan
};

// BINNED (reserved tile regions, fused_preprocess_bin_kernel): the block's histogram stays in LDS -- the caller reserves the
// block's part of every tile's region with it and scatters the keys itself -- and what the scatter needs of the splat comes
// back in `po`.
struct PreOut { float u, v, cutoff, depth; int rad; };

template <bool EWA_MFMA, bool PHASES, bool BINNED = false>
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
    int woff = -1;                               // its offset in the slot pool (more than SLOT_MAX tiles)
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
        int rad = 0, nt_slots = 0;
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
                    nt_slots = min(nt, SLOT_MAX);
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
        int4* iv = reinterpret_cast<int4*>(a.slot_inv + (size_t)i * SLOT_MAX);
        const int4 none = make_int4(-1, -1, -1, -1);
        // only the slots of the splat's own tile rectangle are ever read (gather of the backward)
#pragma unroll
        for (int q = 0; q < SLOT_MAX / 4; ++q)
            if (4 * q < nt_slots) iv[q] = none;
        if (wnt > SLOT_MAX) {
            // too many tiles for the slot row: reserve wnt entries of the pool; the row's first
            // entry carries the pool offset as -2 - offset
            const int off = atomicAdd(a.pool_counter, wnt);
            if (off + wnt <= a.pool_cap) {
                woff = off;
                a.slot_inv[(size_t)i * SLOT_MAX] = -2 - off;
            } else {
                *a.overflow = 1;
            }
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
            const int soff = __shfl(woff, src);
            for (int q = lane; q < snt; q += 64) {
                const int tx = sx0 + q % snx, ty = sy0 + q / snx;
                if (tile_hit2(su, sv, sc, tx, ty)) atomicAdd(&hist[ty * gx + tx], 1);
                if (soff >= 0) a.slot_pool[soff + q] = -1;
            }
        }
    }
    if (a.scale_rows_mode) {
        // (no atomics on global memory: one partial per block, folded by every block of the backward kernel)
        const int wcnt = __popcll(__ballot(in_scale_rows));
        __shared__ int32_t s_cnt[BIN_BLOCK / 64];
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = wcnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int w = 0; w < BIN_BLOCK / 64; ++w) tot += s_cnt[w];
            a.scale_cnt[blockIdx.x] = tot;
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
struct BinArgs {
    const int4* region;              // [T] {start, capacity, position in the sort's order}, written at the end of the iteration before
    int32_t* fill;                   // [T] keys counted so far, BY POSITION: the sort's workgroup reads its count beside its order
                                     // entry instead of behind it (zeroed with the regions)
    unsigned long long* keys;
    int K_cap;
    int32_t* regions_valid;          // set with the regions, checked and cleared here
    const int32_t* extent_next;      // one past the last region ...
    int32_t* extent;                 // ... published as the extent of THIS iteration's lists (gfl_fit_snapshot_stage)
    int32_t* pull_counters; int n_pull;      // the blend launches' pull counters (a forward-only call may have used them since
                                             //  the regions were reserved)
};

__host__ __device__ __forceinline__ int region_cap(int c) { return c + (c >> 2) + 32; }

template <bool EWA_MFMA>
__global__ void __launch_bounds__(BIN_BLOCK) fused_preprocess_bin_kernel(const float* __restrict__ params, PreArgs a,
                                                                         const uint8_t* __restrict__ row_flags, BinArgs b) {
    extern __shared__ int32_t hist[];               // [T] counts, then cursors; [T] limits behind them; 64 idle cursors
    const int T = a.gx * a.gy;
    int32_t* lim = hist + T;
    const int tid = threadIdx.x;
    GFL_PHASE(1, 0);                                 // (the column scan's row of the phase trace: it does not run here)
    const int i = blockIdx.x * BIN_BLOCK + tid;
    float4 row_v[4] = {};
    unsigned own_flags = 0;
    if (i < a.N) {
        const float4* prow = reinterpret_cast<const float4*>(params + (size_t)i * ROW);
#pragma unroll
        for (int q = 0; q < 4; ++q) row_v[q] = prow[q];
        if (a.scale_rows_mode && row_flags) own_flags = row_flags[i];
    }
    // this lane's tiles' regions: requested here, used after the preprocess
    constexpr int PER_MAX = 8;                       // (T <= 4096: fit_reserved_ok)
    int4 reg[PER_MAX];
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) reg[k] = tid + k * BIN_BLOCK < T ? b.region[tid + k * BIN_BLOCK] : make_int4(0, 0, 0, 0);
    if (blockIdx.x == 0 && tid == 0) {
        if (*b.regions_valid == 0) *a.overflow = 2;      // the host asked for regions nobody has reserved
        *b.regions_valid = 0;
        *b.extent = *b.extent_next;
    }
    if (blockIdx.x == 0)
        for (int c = tid; c < b.n_pull; c += BIN_BLOCK) b.pull_counters[c] = 0;
    for (int t = tid; t < T; t += BIN_BLOCK) hist[t] = 0;
    __syncthreads();
    GFL_PHASE(1, 1);
    PreOut o = {0.f, 0.f, 0.f, 0.f, 0};
    preprocess_block<EWA_MFMA, false, true>(a, row_v, own_flags, i, hist, &o);      // (ends behind a barrier)
    GFL_PHASE(1, 2);
    // ---- this block's part of every tile's region
    int got[PER_MAX], cnt[PER_MAX];
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int t = tid + k * BIN_BLOCK;
        cnt[k] = t < T ? hist[t] : 0;
        got[k] = cnt[k] > 0 ? atomicAdd(&b.fill[reg[k].z], cnt[k]) : 0;
    }
    bool over = false, over_cap = false;
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int t = tid + k * BIN_BLOCK;
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
                                                         int32_t* __restrict__ pre_valid, int expect_pre,
                                                         int32_t* __restrict__ overflow) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *pool_counter = 0;   // preprocess is done with it
        // this forward has no preprocess launch of its own (expect_pre): the previous iteration's tail must have run it
        if (expect_pre && *pre_valid == 0) *overflow = 2;
        *pre_valid = 0;
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

static_assert(SCHED_BLOCK == BIN_BLOCK, "the tile scheduler runs as one extra block of the scatter launch");

__device__ __forceinline__ bool sched_xcd_usable(const Sched& sc, int T, int block) {
    return sc.xcd && T <= SCHED_PLAN_TILES && sc.nq % 8 == 0 && sc.nq / 8 <= 64 && sc.nq <= block;
}

// The order the tile sort's workgroups take the tiles in (one workgroup of the scatter launch, beside the scatter's own):
// inside every XCD's run of tiles (the sort keeps workgroup b's tile on XCD b % 8, gfl_tile_sort.hpp) the tiles with
// more than twice the mean list length first, both classes in their old order -- a stable partition from two scans
// (list lengths -> offsets, heavy flags -> ranks).  {tile, start, end} per position: the sort reads ONE 16-byte item.
// RESERVE (reserved tile regions, fused_preprocess_bin_kernel): the same walk at the END of an iteration, for the NEXT one --
// every tile gets region_cap(count) positions instead of count, {tile, start, capacity, split} per position and
// region[tile] = {start, capacity, position} for the binning launch; the fill counters are zeroed; *extent_next = one past the last
// region (beyond K_cap: overflow = 1, the lists must grow), *total = the pairs of the iteration that ends here.
struct ReserveOut { int4* region; int32_t* fill; int32_t* extent_next; int32_t* total; int32_t* overflow; int K_cap; };

template <int BLOCK, bool RESERVE>
__device__ void build_sort_order(const int32_t* __restrict__ tile_counts, int T, int4* __restrict__ sort_order,
                                 int32_t* __restrict__ wsum /* [BLOCK / 64] */, ReserveOut ro = ReserveOut{}) {
    __shared__ int32_t hsum[BLOCK / 64], hstart[9], s_nsplit;
    __shared__ int32_t csum[BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (T + BLOCK - 1) / BLOCK;
    const int t0 = tid * per;
    constexpr int PER_MAX = 4096 / BLOCK;
    // the trailer behind order[T]: the positions of the lists the sort cuts in two (gfl_tile_sort.hpp), their number first
    int32_t* trailer = reinterpret_cast<int32_t*>(sort_order + T);
    if (tid == 0) s_nsplit = 0;
    if (per > PER_MAX) {                                 // more than 4096 tiles: the plain order (one lane; never hot)
        if (tid == 0 && !RESERVE) {                      // (no regions for such grids: fit_reserved_ok)
            int run = 0;
            for (int t = 0; t < T; ++t) {
                const int c = tile_counts[t];
                sort_order[t] = make_int4(t, run, run + c, 0);
                run += c;
            }
            trailer[0] = 0;
        }
        return;
    }
    int cnt[PER_MAX];
    int local = 0, local_c = 0;
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const bool ok = k < per && t0 + k < T;
        cnt[k] = ok ? tile_counts[t0 + k] : 0;
        local += ok ? (RESERVE ? region_cap(cnt[k]) : cnt[k]) : 0;         // positions the tile gets
        local_c += cnt[k];
    }
    if (RESERVE) {
#pragma unroll
        for (int k = 0; k < PER_MAX; ++k)
            if (k < per && t0 + k < T) ro.fill[t0 + k] = 0;
        // (mean list length for the heavy-tile threshold: from the counts, not from the regions)
        int cs = local_c;
#pragma unroll
        for (int off = 32; off; off >>= 1) cs += __shfl_xor(cs, off);
        if (lane == 0) csum[wid] = cs;
    }
    int sc = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(sc, off);
        if (lane >= off) sc += n;
    }
    if (lane == 63) wsum[wid] = sc;
    __syncthreads();
    int run = sc - local, total = 0;
    for (int w = 0; w < BLOCK / 64; ++w) {
        run += w < wid ? wsum[w] : 0;
        total += wsum[w];
    }
    int pairs = total;
    if (RESERVE) {
        pairs = 0;
        for (int w = 0; w < BLOCK / 64; ++w) pairs += csum[w];
        if (tid == 0) {
            *ro.extent_next = min(total, ro.K_cap);
            *ro.total = pairs;
            if (total > ro.K_cap) *ro.overflow = 1;
        }
    }
    const int thr = max(2 * (pairs / max(T, 1)), 64);
    int lh = 0;
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) lh += cnt[k] > thr ? 1 : 0;
    int hs = lh;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(hs, off);
        if (lane >= off) hs += n;
    }
    if (lane == 63) hsum[wid] = hs;
    __syncthreads();
    int H = hs - lh;                                     // heavy tiles before this lane's first tile
    for (int w = 0; w < wid; ++w) H += hsum[w];
    // XCD x owns the tiles [start(x), start(x + 1)): start(x) = x q + min(x, r) (xcd_logical_block)
    const int q = T >> 3, r = T & 7, big = r * (q + 1);
    auto start_of = [&](int x) { return x < r ? x * (q + 1) : big + (x - r) * q; };
    if (tid < 9) {
        // heavy tiles before each XCD's run: found by the lane that owns the run's first tile, below; runs that are empty
        // (fewer than eight tiles) keep the total
        int all = 0;
        for (int w = 0; w < BLOCK / 64; ++w) all += hsum[w];
        hstart[tid] = all;
    }
    __syncthreads();
    int x = t0 < big ? t0 / (q + 1) : r + (q ? (t0 - big) / q : 0);       // (one division per lane; a lane's tiles cross
    int next = x < 7 ? start_of(x + 1) : T;                                //  at most one boundary)
    {
        int Hk = H, xk = x, nk = next;
#pragma unroll
        for (int k = 0; k < PER_MAX; ++k) {
            const int t = t0 + k;
            if (k < per && t < T) {
                if (t == nk) { ++xk; nk = xk < 7 ? start_of(xk + 1) : T; }
                if (t == start_of(xk)) hstart[xk] = Hk;
                Hk += cnt[k] > thr ? 1 : 0;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER_MAX; ++k) {
        const int t = t0 + k;
        if (k < per && t < T) {
            if (t == next) { ++x; next = x < 7 ? start_of(x + 1) : T; }
            const int sx = start_of(x);
            const int hx = hstart[x], nh = (x < 7 ? hstart[x + 1] : hstart[8]) - hx;      // heavy tiles of this run
            const bool heavy = cnt[k] > thr;
            const int hr = H - hx;
            const int pos = sx + (heavy ? hr : nh + (t - sx) - hr);
            int w = 0;
            if (cnt[k] > SORT_SPLIT_MIN && cnt[k] <= 4 * BIN_BLOCK) {
                const int j = atomicAdd(&s_nsplit, 1);       // (which extra workgroup takes which tile does not matter)
                if (j < SORT_MAX_SPLIT) { w = 1 + j; trailer[1 + j] = pos; }
            }
            const int size = RESERVE ? region_cap(cnt[k]) : cnt[k];
            sort_order[pos] = make_int4(t, run, RESERVE ? size : run + cnt[k], w);
            if (RESERVE) ro.region[t] = make_int4(run, size, pos, 0);
            H += heavy ? 1 : 0;
            run += size;
        }
    }
    __syncthreads();
    if (tid == 0) trailer[0] = min((int)s_nsplit, SORT_MAX_SPLIT);
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

// ------------------------------------------------------------------- blend (C = 4)
constexpr int BLEND_WG_PER_CU = 8;
constexpr int FWD_WG_PER_CU = 5;       // the forward blend trades workgroups per CU for registers: 72 VGPRs for four splats per trip (six per CU,
                                       // rounds 1-3); 96 since the long-tile walk keeps sixteen colour sums per lane (round 4: at six per CU the
                                       // kernel spilled 23 registers -- 28 MB of scratch traffic per launch; five cost nothing measurable)
constexpr int FWD_UNITS = 4;
constexpr int FWD_SPLIT_MIN = 448;     // forward: a queue's first tile is walked as four blocks on four CUs when its list is longer
                                       // (round 4 sweep, sixteen-splat steps: bench-scene forward 47.0 / 40.8 / 39.7 / 39.7 us and 4-frame clip
                                       //  fit 0.483 / 0.485 / 0.490 / 0.523 s at 256 / 448 / 640 / never)
constexpr int FB = 256;   // staged splats per batch (forward)
#ifndef GFL_FWD_LONG_BATCH
#define GFL_FWD_LONG_BATCH 256
#endif
constexpr int FBL = GFL_FWD_LONG_BATCH;   // ... of the long-tile walk (512: forward inside a clip fit 56.6 against 52.7 us, round 4)
constexpr int FBB = 192;  // backward: 18.6 KB of LDS per workgroup -> 8 workgroups per CU (the tile queues of
                          // gfl_sched.hpp assume that all workgroups of a blend launch are resident)

struct RecLDS {
    float4 p0, p1, p2;    // p2 = (b, depth, cutoff, radius bits)
};

__device__ __forceinline__ bool splat_alpha2(const float4& p0, const float4& p1, float fx, float fy, float& alpha,
                                             float& G) {
#pragma clang fp contract(off)
    // branch-free; the same instruction sequence in the forward and the backward kernel so that
    // both take the same skip/keep decision for every (pixel, splat)
    const float dx = p0.x - fx, dy = p0.y - fy;
    const float q = __builtin_fmaf(p0.z * dx, dx, (p1.x * dy) * dy);
    const float power = __builtin_fmaf(-0.5f, q, -((p0.w * dx) * dy));
    G = __expf(fminf(power, 0.f));
    alpha = fminf(GFL_ALPHA_MAX, p1.y * G);
    return (power <= 0.f) & (alpha >= GFL_ALPHA_MIN);
}

// (block_test / box_hit / block_mask: gfl_math.hpp)

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

__global__ void __launch_bounds__(256, FWD_WG_PER_CU) fused_blend_fwd_kernel(const float* __restrict__ rec,
                                                              const int32_t* __restrict__ ids,
                                                              const int32_t* __restrict__ tile_range, float bg, int W,
                                                              int H, int gx, float* __restrict__ out,
                                                              float* __restrict__ final_T,
                                                              int32_t* __restrict__ n_contrib, TileQueue queue,
                                                              float* __restrict__ ckpt, int mode,
                                                              const unsigned* __restrict__ cmap_mm,
                                                              const float* __restrict__ cmap_lut, int split_min,
                                                              int32_t* __restrict__ tile_work,
                                                              const int32_t* __restrict__ first_slot) {
    // mode 0: the records as they are.  The two snapshot-only images of render.py:76-106 are composites of the SAME
    // lists with other per-splat values, made while a record is staged: mode 1 = colour := turbo map of the splat's
    // depth (apply_float_colormap(non_zero=True), range in cmap_mm), mode 2 = unit blob at the centre (conic 1 0 1,
    // opacity 1).
    __shared__ RecLDS recs[FBL + 1];         // recs[FBL]: an all-zero record (opacity 0: never blends)
    __shared__ unsigned char s_mask[FBL];
    __shared__ unsigned short s_hits[4][FBL];        // long first tiles: a wave's (= a 4x4 quarter's) hit list of the staged batch
    __shared__ int32_t s_gs[4][FBL / 64 + 1];        // ... and the number of hits in front of every 64-slot group
    __shared__ int32_t s_ticket;
    __shared__ int32_t s_simd[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        recs[FBL].p0 = z; recs[FBL].p1 = z; recs[FBL].p2 = z;
    }
    // block plan (gfl_sched.hpp): which 8x8 block of a whole tile this wave walks follows from the SIMD it sits on
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const int simd = (hw_id >> 4) & 3;
    if (lane == 0) s_simd[wave] = simd;
    __syncthreads();
    const bool simd_ok = ((1 << s_simd[0]) | (1 << s_simd[1]) | (1 << s_simd[2]) | (1 << s_simd[3])) == 15;
  for (bool first = true;; first = false) {
    const TileItem item = next_item(queue, &s_ticket, first, true);
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
    if (item.part >= 4) continue;                    // (the backward pass has more segments than the forward pass blocks)
    if (item.part > 0) {
        owner = (item.queue + item.part * (queue.nq / 4)) % queue.nq;
        tile = queue.count[owner] > 0 ? (queue.list[(size_t)owner * queue.cap_q] & 0xffff) : -1;
        if (tile < 0) continue;
    }
    const int tx = tile % gx, ty = tile / gx;
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
        const float fxq = (float)(qx0 + lr);
        float Tq[4], c0q[4], c1q[4], c2q[4], c3q[4];
        int lastq[4];
        unsigned alive = 0;                          // bit g: pixel (column lr, row g) is in the image and has not stopped
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            Tq[g] = 1.f; c0q[g] = 0.f; c1q[g] = 0.f; c2q[g] = 0.f; c3q[g] = 0.f; lastq[g] = 0;
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
            if (__syncthreads_and(alive == 0)) break;
            {
                // FBL / 256 entries per lane, their ids and then their records requested together
                constexpr int PER = FBL / 256;
                int gidx[PER];
#pragma unroll
                for (int e = 0; e < PER; ++e) gidx[e] = base + tid + 256 * e < end ? ids[base + tid + 256 * e] : -1;
                // (requesting the NEXT batch's ids here, a whole walk ahead, was measured again in round 4: forward inside a
                //  clip fit 53.7 against 50.7 us without it, same box)
#pragma unroll 1
                for (int e = 0; e < PER; ++e) {          // (one record at a time: two in flight spilled the walk's registers)
                    if (gidx[e] < 0) continue;
                    const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)gidx[e] * REC);
                    float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
                    if (mode == 1) {
                        const float3 col = cmap_nonzero_lookup(p2.y, cmap_mm, cmap_lut);
                        p1.z = col.x; p1.w = col.y; p2.x = col.z;
                    } else if (mode == 2) {
                        p0.z = 1.f; p0.w = 0.f; p1.x = 1.f; p1.y = 1.f;
                        p2.z = p2.z < 0.f ? p2.z : alpha_cutoff(1.f, 1.f);
                    }
                    const int sl = tid + 256 * e;
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
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float fyq = (float)(qy0 + g);
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
            if (ls == 0 && qx0 + lr < W && qy0 + g < H) {
                const size_t pix = (size_t)(qy0 + g) * W + (qx0 + lr), plane = (size_t)H * W;
                const float Tf = Tq[g];
                out[pix] = fmaf(Tf, bg, s0);
                out[plane + pix] = fmaf(Tf, bg, s1);
                out[2 * plane + pix] = fmaf(Tf, bg, s2);
                out[3 * plane + pix] = fmaf(Tf, bg, s3);
                final_T[pix] = Tf;
                n_contrib[pix] = lastp;
            }
        }
        if (mode == 0 && lane == 0) atomicAdd(&tile_work[4 * tile + blk], units + 1);
#ifdef GFL_TRACE
        if (lane == 0 && tile < 4096 && mode == 0) {
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
    const float fx = (float)px, fy = (float)py;
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
    int last = 0;
    const unsigned long long alive0 = __ballot(inside);            // every pixel of the box that is in the image

    for (int base = start; base < end; base += FB) {
        if (__syncthreads_and(Tw == 0.f)) break;
        const int idx = base + tid;
        if (idx < end) {
            const int g = ids[idx];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * REC);
            float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
            if (mode == 1) {
                const float3 col = cmap_nonzero_lookup(p2.y, cmap_mm, cmap_lut);
                p1.z = col.x; p1.w = col.y; p2.x = col.z;
            } else if (mode == 2) {
                p0.z = 1.f; p0.w = 0.f; p1.x = 1.f; p1.y = 1.f;
                p2.z = p2.z < 0.f ? p2.z : alpha_cutoff(1.f, 1.f);
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
        out[pix] = fmaf(T, bg, a0);
        out[plane + pix] = fmaf(T, bg, a1);
        out[2 * plane + pix] = fmaf(T, bg, a2);
        out[3 * plane + pix] = fmaf(T, bg, a3);
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
    if (mode == 0 && lane == 0) atomicAdd(&tile_work[4 * tile + wb], units + 1);     // per 8x8 block (gfl_sched.hpp: block plan)
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

struct AdamCfg {
    float lr, b1, b2, eps, lr_end_factor;
    int total_iters;
};

__device__ __forceinline__ void adam_scalars(const AdamCfg& a, int e, float lr, float& step_size, float& inv_sqrt_bc2) {
    const float t = (float)(e + 1);
    if (a.total_iters > 0) lr *= 1.f + (a.lr_end_factor - 1.f) * (float)min(e, a.total_iters) / (float)a.total_iters;
    step_size = lr / (1.f - powf(a.b1, t));
    inv_sqrt_bc2 = 1.f / sqrtf(1.f - powf(a.b2, t));
}

__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, const AdamCfg& a, float step_size,
                                             float inv_sqrt_bc2) {
    m = fmaf(a.b1, m, (1.f - a.b1) * g);
    v = fmaf(a.b2, v, (1.f - a.b2) * g * g);
    const float denom = sqrtf(v) * inv_sqrt_bc2 + a.eps;
    return p - step_size * (m / denom);
}

// Iterations that do not move the camera (lr_camera = 0: the first frame and every joint stage, three quarters of a clip's
// iterations; or nothing is stepped any more after a densification) need no pose gradient, and what is left of the
// camera / depth-affine launch -- fold the loss partials into sums[], step the depth affine, advance the step counter --
// depends on the loss launch only.  One workgroup of the BACKWARD BLEND launch does it before its first tile (the per-splat
// launch that follows reads the counter one too high and is told so): a launch less per iteration, and no reduction of
// the twelve extrinsic partials in the per-splat launch.
struct LossTail {
    int enabled;
    const float* p_ssim; int n_ssim;        // SSIM partials of the loss launch
    const float* p_grad; int n_grad;        // [n_grad][4] = {sum mse_px, sum depth term, d/d depth_a, d/d depth_b}
    float* depth_ab; float* ab_m; float* ab_v;
    float* sums;                            // [8]
    AdamCfg ac_ab;
    int step_affine;                        // hp->step_camera (after a densification nothing is stepped)
    int32_t* d_step;
    float* d_extr_out;                      // [12]: zeros (not computed in such an iteration)
    int32_t* overflow;                      // [4]: the forward dropped pairs ([0] | [2]) -> nothing is stepped, [1] counts the iteration
};

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
                                                              const float* __restrict__ render, LossTail ltail) {
    // (see LossTail.  The workgroup that holds the LAST pre-assigned slot of queue 0 does it before its first item: ~3 us
    //  that the seven other workgroups of its queue absorb.  A workgroup of its own behind the others only started when one
    //  of them -- all persistent -- had finished: +1 us at the end of the launch.)
    if (ltail.enabled && blockIdx.x == gridDim.x - queue.nq) loss_tail(ltail);
    // No global atomics: the four waves of the tile combine their per-splat sums in LDS and the
    // tile writes ONE 48-byte row per (splat, tile) pair at the pair's list position with plain,
    // coalesced stores.  The per-splat kernel gathers its rows afterwards (deterministic).
    __shared__ RecLDS recs[FBB];
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
    const TileItem item = next_item(queue, &s_ticket, first, true);
    const int tile = item.tile;
    if (tile < 0) break;
    const unsigned plan = item.plan;
    const bool plan_ok = simd_ok && ((1 << (plan & 3)) | (1 << ((plan >> 2) & 3)) | (1 << ((plan >> 4) & 3)) | (1 << ((plan >> 6) & 3))) == 15;
    // segments of one tile run side by side on one CU and a pile sits in ONE of its blocks: from four segments on, segment
    // p turns the plan by p SIMDs (the scheduler counts such a tile's blocks a quarter each on every SIMD: Sched.rotate)
    const int n_parts_ = item.part >= 0 ? heavy_parts(tile_range[2 * tile + 1] - tile_range[2 * tile]) : 1;
    const int psimd = (queue.rotate && n_parts_ >= 4) ? ((simd + item.part) & 3) : simd;
    const int blk = plan_ok ? (int)((plan >> (2 * psimd)) & 3u) : wave;         // the 8x8 block of the tile this wave walks
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * GFL_TILE + (blk & 1) * 8 + (lane & 7);
    const int py = ty * GFL_TILE + (blk >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
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

    // pairs behind the deepest contributor of the tile get a zero row
    if (last_part)
        for (int p = depth_n + tid; p < total; p += 256) {
            float4* o = reinterpret_cast<float4*>(pair_grad + (size_t)(start + p) * PG);
            o[0] = zero4; o[1] = zero4; o[2] = zero4;
        }

    for (int r0 = 0; r0 < hi - lo; r0 += FBB) {
        const int pos_t = tid < FBB ? hi - 1 - r0 - tid : -1;          // slot tid <-> list position pos_t
        __syncthreads();
        if (pos_t >= lo) {
            const int g = ids[start + pos_t];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)g * REC);
            const float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
            recs[tid].p0 = p0; recs[tid].p1 = p1; recs[tid].p2 = p2;
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
            const float4* a4 = reinterpret_cast<const float4*>(&acc[tid][0]);
            float4* o = reinterpret_cast<float4*>(pair_grad + (size_t)(start + pos_t) * PG);
            o[0] = a4[0]; o[1] = a4[1]; o[2] = a4[2];
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

// ------------------------------------------------------------------ backward blend, "rows" formulation (round 4)
// The kernel above gives a wave the 64 pixels of an 8x8 block and ONE splat per step: every step ends in a reduction of
// ten sums over the 64 lanes (27 of its ~65 VALU instructions), 41 % of the lanes see the splat at all, and the
// transmittance recurrence is one dependent chain per pixel.  Here a wave owns a 4x4 QUARTER of the tile (sixteen waves,
// one 1024-lane workgroup per tile item) and a step takes SIXTEEN splats of the quarter's hit list at once:
//     lane = (s, r): s = lane & 15 = the splat of the step (list order along the lanes of a 16-lane row),
//                    r = lane >> 4 = the pixel column of the quarter; four passes g = 0..3 over the quarter's pixel rows.
// * The recurrences along the list become ROW SCANS: T in front of splat k = T_in * prod_{j<=k} 1 / (1 - a_j)  (four DPP
//   row_shr multiplies), S behind splat k = S_in + sum_{j<k} h_j w_j (four DPP row_shr adds); the carries for the next
//   step are lane 15's values (ds_swizzle broadcast inside the row).
// * The sums over the PIXELS of a splat need no cross-lane work inside a pass: a lane adds its four pixels (one per pass)
//   in registers; once per STEP the four rows are folded with two permlane swap stages (the first two stages of the
//   reduce-scatter above: ten values -> three registers) and every lane adds its components into the splat's LDS row.
// ~290 VALU instructions per step of sixteen (splat, quarter) units = ~18 per unit, against ~65 per (splat, 8x8 block)
// unit = ~30 per quarter-sized share of it; 4x4 units that are evaluated at all have 63 % of their lanes inside the splat
// (tools/lane_efficiency.py: 1.08 M such units against 0.42 M 8x8 units on the bench scene).
// Products and sums along the list are formed in tree order by the scans, not one after the other: gradients differ from
// the kernel above in the last bits (they already differ from run to run there: unordered LDS adds).  The skip / keep
// decision of every (pixel, splat) pair is the forward's: the same splat_alpha2, the same n_contrib.
#ifndef GFL_ROWS_BATCH
#define GFL_ROWS_BATCH 128
#endif
constexpr int RB_BATCH = GFL_ROWS_BATCH;   // staged splats per batch (LDS: 96 B per slot + 8 KB of pixel state -> 8 workgroups per CU at 128)

// the 4x4 quarters of two 8x8 blocks (b0, b0 + 1) of a tile a staged splat reaches with alpha >= 1/255:
// bit 4 * (b - b0) + c, c = the quarter inside the block (x fastest)
__device__ __forceinline__ unsigned quarter_mask_pair(const float4& p0, const float4& p1, float cutoff, int tx0, int ty0, int b0) {
    const BlockTest t = block_test(p0, p1, cutoff);
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int b = b0 + k;
        const float bx = (float)(tx0 + (b & 1) * 8), by = (float)(ty0 + (b >> 1) * 8);
        if (!box_hit(t, bx, bx + 7.f, by, by + 7.f)) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x_lo = bx + (float)((c & 1) * 4), y_lo = by + (float)((c >> 1) * 4);
            if (box_hit(t, x_lo, x_lo + 3.f, y_lo, y_lo + 3.f)) m |= 1u << (4 * k + c);
        }
    }
    return m;
}

struct PixState {            // per pixel of the workgroup's tile, in LDS: 32 bytes
    float T, S;
    int last, pad;
    float g0, g1, g2, g3;
};

template <int SUMS>
__global__ void __launch_bounds__(256, 5) fused_blend_bwd_rows_kernel(
    const float* __restrict__ rec, const int32_t* __restrict__ ids, const int32_t* __restrict__ tile_range, float bg, int W,
    int H, int gx, const float* __restrict__ final_T, const int32_t* __restrict__ n_contrib,
    const float* __restrict__ d_out, float* __restrict__ pair_grad, TileQueue queue, int32_t* __restrict__ tile_work,
    const float* __restrict__ ckpt, const float* __restrict__ render, LossTail ltail) {
    if (ltail.enabled && blockIdx.x == gridDim.x - queue.nq) loss_tail(ltail);
    constexpr int FB_ = RB_BATCH;
    static_assert(2 * FB_ == 256, "two staging lanes per slot");
    __shared__ RecLDS recs[FB_ + 1];                 // recs[FB_]: the null record (opacity 0)
    __shared__ float acc[FB_ + 1][REC];              // acc[FB_]: where the null record's (zero) sums go
    __shared__ unsigned char s_mask[2][FB_];         // quarters of blocks {0, 1} / {2, 3} a slot's splat reaches
    __shared__ unsigned char s_hits[4][FB_];         // a wave's hit list of the quarter it is walking
    __shared__ __attribute__((aligned(16))) PixState s_px[4][64];
    __shared__ int32_t s_max_last;
    __shared__ int32_t s_ticket;
    __shared__ int32_t s_simd[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ls = lane & 15, lr = lane >> 4;        // the splat of a step, the pixel column of the quarter
    if (tid == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        recs[FB_].p0 = z; recs[FB_].p1 = z; recs[FB_].p2 = z;
    }
    // which components of a splat's row this lane adds after the two swap stages (gfl_common.hpp): by row of the wave
    const int hi = lane >> 5, odd = (lane >> 4) & 1;
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    const int simd = (hw_id >> 4) & 3;
    if (lane == 0) s_simd[wave] = simd;
    __syncthreads();
    const bool simd_ok = ((1 << s_simd[0]) | (1 << s_simd[1]) | (1 << s_simd[2]) | (1 << s_simd[3])) == 15;
  for (bool first = true;; first = false) {
    const TileItem item = next_item(queue, &s_ticket, first, true);
    const int tile = item.tile;
    if (tile < 0) break;
    const unsigned plan = item.plan;
    const bool plan_ok = simd_ok && ((1 << (plan & 3)) | (1 << ((plan >> 2) & 3)) | (1 << ((plan >> 4) & 3)) | (1 << ((plan >> 6) & 3))) == 15;
    const int blk = plan_ok ? (int)((plan >> (2 * simd)) & 3u) : wave;           // the 8x8 block of the tile this wave walks
    const int tx = tile % gx, ty = tile / gx;
    const int bx0 = tx * GFL_TILE + (blk & 1) * 8, by0 = ty * GFL_TILE + (blk >> 1) * 8;
    const int start = tile_range[2 * tile], end = tile_range[2 * tile + 1];
    const int total = end - start;
    const int parts = item.part >= 0 ? heavy_parts(total) : 1;
    if (item.part >= parts) continue;                // this tile has fewer segments
    const int seg = heavy_seg(total, parts);
    const bool last_part = item.part < 0 || item.part == parts - 1;      // the farthest segment (or the whole tile)
    int units = 0;

    // ---- per-pixel state into LDS: lane = pixel of the 8x8 block ((y << 3) | x), stored by (quarter, row, column)
    int q_last;                                      // deepest contributor of this lane's quarter (lanes 16 c .. 16 c + 15)
    {
        const int lx = lane & 7, ly = lane >> 3;
        const int px = bx0 + lx, py = by0 + ly;
        PixState ps;
        ps.T = 1.f; ps.S = 0.f; ps.last = 0; ps.pad = 0; ps.g0 = 0.f; ps.g1 = 0.f; ps.g2 = 0.f; ps.g3 = 0.f;
        if (px < W && py < H) {
            const size_t pix = (size_t)py * W + px, plane = (size_t)H * W;
            ps.last = n_contrib[pix];
            ps.g0 = d_out[pix]; ps.g1 = d_out[plane + pix]; ps.g2 = d_out[2 * plane + pix]; ps.g3 = d_out[3 * plane + pix];
            if (last_part) {
                ps.T = final_T[pix];
                ps.S = ps.T * bg * (ps.g0 + ps.g1 + ps.g2 + ps.g3);
            } else {
                const float* ck = ckpt + ((size_t)item.queue * (HEAVY_PARTS - 1) + item.part) * 5 * 256 + blk * 64 + lane;
                ps.T = ck[0];
                ps.S = ps.g0 * (render[pix] - ck[256]) + ps.g1 * (render[plane + pix] - ck[512]) +
                       ps.g2 * (render[2 * plane + pix] - ck[768]) + ps.g3 * (render[3 * plane + pix] - ck[1024]);
            }
        }
        const int c = ((ly >> 2) << 1) | (lx >> 2);
        const int slot = c * 16 + (ly & 3) * 4 + (lx & 3);
        float4* dst = reinterpret_cast<float4*>(&s_px[wave][slot]);
        dst[0] = make_float4(ps.T, ps.S, __int_as_float(ps.last), 0.f);
        dst[1] = make_float4(ps.g0, ps.g1, ps.g2, ps.g3);
        // the block's deepest contributor (for the tile's depth_n) and each quarter's (for its hit list)
        int wl = ps.last;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wl = max(wl, __shfl_xor(wl, off));
        if (tid == 0) s_max_last = 0;
        __syncthreads();
        if (lane == 0) atomicMax(&s_max_last, wl);
        __syncthreads();
        // (after the barrier: the pixel state is visible to the whole wave)
        int ql = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) ql = max(ql, s_px[wave][lr * 16 + g * 4 + (ls & 3)].last);
        // lanes of row lr now hold the maximum over the quarter lr's column (ls & 3); fold the four columns
        ql = max(ql, __shfl_xor(ql, 1));
        ql = max(ql, __shfl_xor(ql, 2));
        q_last = ql;                                 // row lr: quarter lr
    }
    const int depth_n = min(total, (int)s_max_last);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // this item walks list positions hi_pos-1 down to lo
    const int lo = item.part > 0 ? item.part * seg : 0;
    const int hi_pos = last_part ? depth_n : min((item.part + 1) * seg, depth_n);
    // pairs behind the deepest contributor of the tile get a zero row
    if (last_part)
        for (int p = depth_n + tid; p < total; p += 256) {
            float4* o = reinterpret_cast<float4*>(pair_grad + (size_t)(start + p) * PG);
            o[0] = zero4; o[1] = zero4; o[2] = zero4;
        }

    for (int r0 = 0; r0 < hi_pos - lo; r0 += FB_) {
        // two staging lanes per slot: both fetch the record, each tests the quarters of two of the tile's four blocks
        const int sslot = tid & (FB_ - 1), shalf = tid >> 7;
        const int pos_t = hi_pos - 1 - r0 - sslot;                        // slot <-> list position
        __syncthreads();
        if (pos_t >= lo) {
            const int gid = ids[start + pos_t];
            const float4* r4 = reinterpret_cast<const float4*>(rec + (size_t)gid * REC);
            const float4 p0 = r4[0], p1 = r4[1], p2 = r4[2];
            if (shalf == 0) { recs[sslot].p0 = p0; recs[sslot].p1 = p1; recs[sslot].p2 = p2; }
            s_mask[shalf][sslot] = (unsigned char)quarter_mask_pair(p0, p1, p2.z, tx * GFL_TILE, ty * GFL_TILE, 2 * shalf);
        }
        if (tid <= FB_) {
            float4* az = reinterpret_cast<float4*>(&acc[tid][0]);
            az[0] = zero4; az[1] = zero4; az[2] = zero4;
        }
        __syncthreads();
        const int cnt = min(FB_, hi_pos - lo - r0);
        for (int c = 0; c < 4; ++c) {                // the four quarters of this wave's block
            const int c_last = __shfl(q_last, 16 * c);
            const int mbit = (blk & 1) * 4 + c;
            // ---- the quarter's hit list of the batch, in slot order = back to front
            int n_hit = 0;
            for (int c0 = 0; c0 < cnt; c0 += 64) {
                const int slot = c0 + lane;
                const bool hit = slot < cnt && (hi_pos - 1 - r0 - slot) < c_last && ((s_mask[blk >> 1][slot] >> mbit) & 1);
                const unsigned long long bal = __ballot(hit);
                if (hit) s_hits[wave][n_hit + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0))] = (unsigned char)slot;
                n_hit += (int)__popcll(bal);
            }
            if (n_hit == 0) continue;
            units += n_hit;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the list is read back by other lanes of this wave)
            const float fx = (float)(bx0 + (c & 1) * 4 + lr);
            const int fy0 = by0 + (c >> 1) * 4;
            for (int st = 0; st < n_hit; st += 16) {
                const bool have = st + ls < n_hit;
                const int j = have ? (int)s_hits[wave][st + ls] : FB_;
                const int pos = have ? hi_pos - 1 - r0 - j : 0x7fffffff;
                const float4 p0 = recs[j].p0, p1 = recs[j].p1, p2 = recs[j].p2;
                const float dx = p0.x - fx;
                float v[10];
#pragma unroll
                for (int k = 0; k < 10; ++k) v[k] = 0.f;
#pragma unroll 1
                for (int g = 0; g < 4; ++g) {
                    PixState* sp = &s_px[wave][c * 16 + g * 4 + lr];
                    const float4 st0 = *reinterpret_cast<const float4*>(sp);
                    const float4 gq = *(reinterpret_cast<const float4*>(sp) + 1);
                    const float fy = (float)(fy0 + g);
                    float alpha, G;
                    const bool valid = splat_alpha2(p0, p1, fx, fy, alpha, G) && (pos < __float_as_int(st0.z));
                    const float a_eff = valid ? alpha : 0.f;
                    const float rom = __builtin_amdgcn_rcpf(1.f - a_eff);
                    const float Tk = st0.x * row_scan_mul(rom);                     // T in front of this splat
                    const float h = fmaf(gq.x, p1.z, fmaf(gq.y, p1.w, fmaf(gq.z, p2.x, gq.w * p2.y)));
                    const float w = a_eff * Tk;
                    const float hw = h * w;
                    const float incl = row_scan_add(hw);
                    const float Sk = st0.y + (incl - hw);                            // S behind this splat
                    const float dalpha = valid ? fmaf(Tk, h, -(Sk * rom)) : 0.f;
                    if (ls == 15) *reinterpret_cast<float2*>(sp) = make_float2(Tk, Sk + hw);      // the carries of the next step
                    const float dy = p0.y - fy;
                    const float t5 = G * dalpha;
                    const float dpow = p1.y * t5;
                    const float mx = -dx * dpow, my = -dy * dpow;
                    v[0] += mx;
                    v[1] += my;
                    v[2] = fmaf(dx, mx, v[2]);
                    v[3] = fmaf(dx, my, v[3]);
                    v[4] = fmaf(dy, my, v[4]);
                    if (SUMS >= 7) v[5] += t5;
                    if (SUMS == 10) { v[6] = fmaf(w, gq.x, v[6]); v[7] = fmaf(w, gq.y, v[7]); v[8] = fmaf(w, gq.z, v[8]); }
                    v[9] = fmaf(w, gq.w, v[9]);
                }
                // ---- fold the four rows (the four pixel columns) and add into the splat's LDS row
                float* arow = &acc[j][0];
                if (SUMS == 10) {
                    float a[5] = {v[0], v[2], v[4], v[6], v[8]}, b[5] = {v[1], v[3], v[5], v[7], v[9]};
                    permlane32_swap_x5(a, b);
                    float w5[5];
#pragma unroll
                    for (int m = 0; m < 5; ++m) w5[m] = a[m] + b[m];               // lanes 0-31: component 2m, 32-63: 2m + 1
                    float cc[3] = {w5[0], w5[2], w5[4]}, dd[3] = {w5[1], w5[3], 0.f};
                    permlane16_swap_x3(cc, dd);
                    atomicAdd(&arow[2 * odd + hi], cc[0] + dd[0]);
                    atomicAdd(&arow[4 + 2 * odd + hi], cc[1] + dd[1]);
                    if (!odd) atomicAdd(&arow[8 + hi], cc[2] + dd[2]);
                } else {
                    // SUMS 7: v0..v5, v9; SUMS 6: v0..v4, v9 (the colour sums are neither formed nor folded)
                    float a[3] = {v[0], v[2], v[4]}, b[3] = {v[1], v[3], SUMS == 7 ? v[5] : v[9]};
                    float e = v[9], f = 0.f;
                    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %3\n\tv_permlane32_swap_b32 %1, %4\n\tv_permlane32_swap_b32 %2, %5"
                        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
                    const float w3[3] = {a[0] + b[0], a[1] + b[1], a[2] + b[2]};  // lower half: v0 v2 v4, upper: v1 v3 v5|v9
                    if (SUMS == 7) { permlane32_swap(e, f); }
                    const float w9 = e + f;                                       // SUMS 7, lower half: v9 over lanes i, i + 32
                    float cc[2] = {w3[0], w3[2]}, dd[2] = {w3[1], SUMS == 7 ? w9 : 0.f};
                    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3"
                        : "+v"(cc[0]), "+v"(cc[1]), "+v"(dd[0]), "+v"(dd[1]));
                    // cc0 + dd0 by row (hi, odd): component 2 odd + hi of {v0 v1 v2 v3};
                    // cc1 + dd1: even rows: component 4 (hi 0) / 5 or 9 (hi 1); odd rows: v9 (SUMS 7, hi 0 only)
                    atomicAdd(&arow[2 * odd + hi], cc[0] + dd[0]);
                    const float x1 = cc[1] + dd[1];
                    if (!odd) atomicAdd(&arow[hi == 0 ? 4 : (SUMS == 7 ? 5 : 9)], x1);
                    else if (SUMS == 7 && hi == 0) atomicAdd(&arow[9], x1);
                }
            }
        }
        __syncthreads();
        if (tid < FB_ && pos_t >= lo) {
            const float4* a4 = reinterpret_cast<const float4*>(&acc[tid][0]);
            float4* o = reinterpret_cast<float4*>(pair_grad + (size_t)(start + pos_t) * PG);
            o[0] = a4[0]; o[1] = a4[1]; o[2] = a4[2];
        }
    }
    // work feedback for the next iteration's schedule, per 8x8 block (gfl_sched.hpp): (splat, quarter) units
    if (lane == 0) atomicAdd(&tile_work[4 * tile + blk], units + 1);
  }
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
__global__ void __launch_bounds__(256) footprint_kernel(const float* __restrict__ rec, const int32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ tile_range,
                                                        const uint8_t* __restrict__ foot_flags, int W, int H, int gx,
                                                        uint8_t* __restrict__ keep) {
    __shared__ RecLDS recs[FB];
    __shared__ unsigned char s_mask[FB];
    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int px = tx * GFL_TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GFL_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float fx = (float)px, fy = (float)py;
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

// ------------------------------------------------- preprocess backward + Adam (A13)
// (AdamCfg, adam_scalars, adam_update: above, beside loss_tail)

struct RegCfg {                 // per-splat regularisers (trainer.py:490-530)
    float lambda_scale;         // lambda_scale (the row count divides it in the kernel)
    int scale_blocks;           // partial counts to fold (blocks of the preprocess launch)
    float lambda_var;           // lambda_var / N
    float lambda_flow;          // lambda_flow (per-row weight in flow_w carries 1/(2 count))
    float lambda_still;         // lambda_still (per-row weight in still_w carries 1/count)
    int freeze_rgb;             // trainer.py:537-540
    int freeze_all;             // camera_only, trainer.py:548-551
    int no_pose_grad;           // the camera does not move in this iteration (LossTail): the step counter has been advanced
                                // already, the twelve extrinsic partials are not reduced
};

// Camera + depth-affine step as the TAIL of the per-splat kernel (round 3; it used to be a launch of its own: one
// workgroup, 7.5 us of dependent L2 round trips = 3 % of every iteration).  Every workgroup of
// fused_preprocess_bwd_adam takes a ticket after it has stored its row of extr partials; the workgroup that draws the
// last one folds the rows and the loss partials (fixed shape: the result does not depend on WHICH workgroup is last),
// chains d_extr to the pose, steps Adam for pose + depth affine and advances the step counter.  Nothing else reads
// pose / depth_ab / step after that point of the launch: every workgroup reads them before it takes its ticket.
struct CamTail {
    const float* p_ssim; int n_ssim;        // SSIM partials of the loss launch
    const float* p_grad; int n_grad;        // [n_grad][4] = {sum mse_px, sum depth term, d/d depth_a, d/d depth_b}
    float* pose; float* pose_m; float* pose_v;
    float* depth_ab; float* ab_m; float* ab_v;
    float* sums;                            // [8]: sums[0..4] as gfl_loss_fwd_bwd documents
    AdamCfg ac_cam, ac_ab;
    int step_camera;
    int32_t* d_step;
    float* d_extr_out;                      // [12]
    int32_t* ticket;                        // zero between launches
    int32_t* overflow;                      // [2], as LossTail
};

__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // written by other workgroups of THIS launch
}

__device__ __forceinline__ void camera_tail(const CamTail& t, const float* partial, int rows, int e_step) {
    constexpr int NV = 17;   // 12 extr + {mse, ssim, depth, d/da, d/db}
    __shared__ int32_t s_last;
    __shared__ float red[REDUCE_BLOCK / 64][NV];
    __shared__ float ge[NV];
    // The partial row was stored with agent-scope (write-through) atomic stores; wait until they have left this
    // wave -- NOT __threadfence(): an agent-scope release on gfx950 writes back the XCD's whole L2, i.e. the 30 MB of
    // parameter / moment rows this launch has just written (measured: the launch went from 22 to 59 us) -- and only
    // then draw the ticket.  The last workgroup reads the rows with agent-scope loads.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(t.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == rows - 1;
    __syncthreads();
    if (!s_last) return;
    // thread 0 needs these after the reduction: request them now
    float pz[7], pm[7], pv[7], ab[2], abm[2], abv[2];
    if (threadIdx.x == 0) {
        __hip_atomic_store(t.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 7; ++k) { pz[k] = t.pose[k]; pm[k] = t.pose_m[k]; pv[k] = t.pose_v[k]; }
#pragma unroll
        for (int k = 0; k < 2; ++k) { ab[k] = t.depth_ab[k]; abm[k] = t.ab_m[k]; abv[k] = t.ab_v[k]; }
    }
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
    // one workgroup, nothing to overlap a load with but other loads: issue them in batches
    for (int r = threadIdx.x; r < rows; r += REDUCE_BLOCK) {
        float v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = ld_agent(partial + (size_t)r * 12 + k);
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] += v[k];
    }
    for (int r0 = threadIdx.x; r0 < t.n_ssim; r0 += 8 * REDUCE_BLOCK) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (r0 + u * REDUCE_BLOCK < t.n_ssim) ? t.p_ssim[r0 + u * REDUCE_BLOCK] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[13] += v[u];
    }
    for (int r0 = threadIdx.x; r0 < t.n_grad; r0 += 4 * REDUCE_BLOCK) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            q[u] = (r0 + u * REDUCE_BLOCK < t.n_grad) ? reinterpret_cast<const float4*>(t.p_grad)[r0 + u * REDUCE_BLOCK]
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[12] += q[u].x; acc[14] += q[u].y; acc[15] += q[u].z; acc[16] += q[u].w; }
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
        for (int w = 0; w < REDUCE_BLOCK / 64; ++w) x += red[w][threadIdx.x];
        ge[threadIdx.x] = x;
        if (threadIdx.x < 12) t.d_extr_out[threadIdx.x] = x;
        else t.sums[threadIdx.x - 12] = x;
    }
    if (threadIdx.x >= NV && threadIdx.x < NV + 3) t.sums[threadIdx.x - NV + 5] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int e = e_step;
        if (t.overflow && (t.overflow[0] | t.overflow[2]) != 0) {
            t.overflow[1] += 1;
            return;
        }
        if (t.step_camera) {
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
            adam_scalars(t.ac_cam, e, t.ac_cam.lr, ss, isb);
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                t.pose[k] = adam_update(pz[k], gp[k], pm[k], pv[k], t.ac_cam, ss, isb);
                t.pose_m[k] = pm[k]; t.pose_v[k] = pv[k];
            }
            adam_scalars(t.ac_ab, e, t.ac_ab.lr, ss, isb);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                t.depth_ab[k] = adam_update(ab[k], ge[15 + k], abm[k], abv[k], t.ac_ab, ss, isb);
                t.ab_m[k] = abm[k]; t.ab_v[k] = abv[k];
            }
        }
        *t.d_step = e + 1;
    }
}

// the next iteration's schedule, built by two extra workgroups of the per-splat launch (rows = the per-splat workgroups)
struct NextSched {
    int rows;
    int T;
    const int32_t* tile_counts;
    Sched bwd, fwd;
    int32_t* valid;
    // a third workgroup reserves the next iteration's tile regions (fused_preprocess_bin_kernel)
    int reserve;
    int4* order_next;
    ReserveOut ro;
    int32_t* regions_valid;
    int32_t* pool_counter;
    int32_t* pull_counters;
    int n_pull;
};

// OP = true is the differentiable operator's backward (gfl_render_bwd): the rows hold ACTIVATED attributes, the
// camera is the extrinsic `pose` points at (12 floats), the caller's dL/d uv and dL/d depth join the gradient, and
// the 14 gradients are WRITTEN to d_params rows instead of stepping Adam (no regularisers, no masks).
// NEXT (round 4): the NEXT iteration's preprocess in the tail of this launch (preprocess_block; BLOCK = BIN_BLOCK then, so
// that a workgroup's splats are one block of the binning): Adam has just stepped the row in registers, and while the
// camera cannot move nothing else is needed -- the next iteration starts at the column scan, a launch and 56 N bytes
// less.  Only plain joint-stage iterations of a multi-iteration call (gfl_fit_iterations) take it.
template <bool OP, int BLOCK, bool NEXT>
__global__ void __launch_bounds__(BLOCK) fused_preprocess_bwd_adam_kernel(
    float* __restrict__ params, float* __restrict__ adam_m, float* __restrict__ adam_v, const float* __restrict__ intr,
    const float* pose, const float* __restrict__ rec, float* __restrict__ d_rec,
    const float* __restrict__ pair_grad, const int32_t* __restrict__ slot_pool,
    const int32_t* __restrict__ tile_range, const int32_t* __restrict__ slot_inv, int gx, int gy, int N, int W, int H,
    const float* __restrict__ flow_target, const float* __restrict__ flow_w, const float* __restrict__ still_target,
    const float* __restrict__ still_w, const uint8_t* __restrict__ row_flags, RegCfg rc, AdamCfg ac,
    const int32_t* d_step, float* partial, const float* __restrict__ d_uv_in,
    const float* __restrict__ d_depth_in, float* __restrict__ d_params, const int32_t* __restrict__ scale_cnt, CamTail tail,
    NextSched ns, PreArgs next, const int32_t* next_overflow) {
    extern __shared__ int32_t sched_scratch[];           // T ints: the scheduler's scratch, or (NEXT) the tile histogram
    if ((int)blockIdx.x >= ns.rows) {
        __shared__ int32_t sched_wsum[BLOCK / 64];
        if ((int)blockIdx.x == ns.rows + 2) {
            // ... and a third the next iteration's tile regions, its sort order, and what the column scan of the exact path
            // resets (the slot pool's counter, the blend launches' pull counters: both done with for this iteration)
            build_sort_order<BLOCK, true>(ns.tile_counts, ns.T, ns.order_next, sched_wsum, ns.ro);
            for (int c = threadIdx.x; c < ns.n_pull; c += BLOCK) ns.pull_counters[c] = 0;
            if (threadIdx.x == 0) {
                *ns.pool_counter = 0;
                *ns.regions_valid = 1;
            }
            return;
        }
        // the two workgroups behind the per-splat ones build the NEXT iteration's tile queues (see fused_scatter_kernel)
        __shared__ SchedLds sched_lds;
        const Sched sc = (int)blockIdx.x == ns.rows ? ns.bwd : ns.fwd;
        schedule_tiles_xcd<BLOCK>(ns.tile_counts, ns.T, sc, sched_scratch, sched_wsum, sched_lds,
                                  reinterpret_cast<uint32_t*>(sched_scratch + ns.T));      // (T <= SCHED_PLAN_TILES: next_sched_ok)
        if (threadIdx.x == 0) *ns.valid = 1;
        return;
    }
    GFL_PHASE(3, 0);
    if (NEXT) {
        // (the barrier that publishes the cleared histogram is the one in front of the tail, below)
        for (int t = threadIdx.x; t < next.gx * next.gy; t += BLOCK) sched_scratch[t] = 0;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // the forward dropped pairs (the lists overflowed, [0], or a tile outgrew its reserved region, [2]): no row is stepped
    const bool dropped = !OP && next_overflow != nullptr && (next_overflow[0] | next_overflow[2]) != 0;
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
                    // The launch has about one wave per SIMD, so nothing hides a dependent load:
                    // the whole slot row is fetched first (up to eight 16-byte loads in flight),
                    // then the gradient rows four at a time (twelve loads in flight), summed in
                    // tile order.  One slot and one row per trip took 2 x nt round trips to L2.
                    const int4* sl4 = reinterpret_cast<const int4*>(slot_inv + (size_t)i * SLOT_MAX);
                    int4 s4[SLOT_MAX / 4];
#pragma unroll
                    for (int k = 0; k < SLOT_MAX / 4; ++k) s4[k] = (4 * k < nt) ? sl4[k] : make_int4(-1, -1, -1, -1);
#pragma unroll
                    for (int k = 0; k < SLOT_MAX / 4; ++k) {
                        if (4 * k >= nt) break;
                        const int pos[4] = {s4[k].x, s4[k].y, s4[k].z, s4[k].w};
                        float4 r0[4], r1[4], r2[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool ok = 4 * k + j < nt && pos[j] >= 0;
                            const float4* g4 = reinterpret_cast<const float4*>(pair_grad + (size_t)(ok ? pos[j] : 0) * PG);
                            r0[j] = g4[0]; r1[j] = g4[1]; r2[j] = g4[2];
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (4 * k + j < nt && pos[j] >= 0) {
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
    // ---- wave-cooperative gather for the few splats with more than SLOT_MAX tiles: their list
    // positions sit in the slot pool (offset encoded in the first slot); rows are wave-summed
    {
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(big);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int si = __shfl(i, src);
            const int nt = __shfl(big_nt, src);
            const int code = slot_inv[(size_t)si * SLOT_MAX];
            float a[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) a[k] = 0.f;
            for (int q = lane; q < nt && code <= -2; q += 64) {
                const int lo = slot_pool[-2 - code + q];
                if (lo >= 0) {
                    const float4* g4 = reinterpret_cast<const float4*>(pair_grad + (size_t)lo * PG);
                    const float4 q0 = g4[0], q1 = g4[1], q2 = g4[2];
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
        {
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
            prow_v[q] = make_float4(pv[4 * q], pv[4 * q + 1], pv[4 * q + 2], pv[4 * q + 3]);      // the stepped row (NEXT)
            prow[q] = prow_v[q];
            mrow[q] = make_float4(mv[4 * q], mv[4 * q + 1], mv[4 * q + 2], mv[4 * q + 3]);
            vrow[q] = make_float4(vv[4 * q], vv[4 * q + 1], vv[4 * q + 2], vv[4 * q + 3]);
        }
        }
        }
    }
    GFL_PHASE(3, 6);
    if (!OP && rc.no_pose_grad) {
        // nobody reads the extrinsic partials of this iteration
    } else if (!OP && tail.ticket && BLOCK == REDUCE_BLOCK) {
        block_reduce_store<12, BLOCK, true>(e, partial);
        camera_tail(tail, partial, ns.rows, e_step);
    } else {
        block_reduce_store<12, BLOCK>(e, partial);
    }
    GFL_PHASE(3, 7);
    if (NEXT) {
        // the next iteration's preprocess on the row Adam has just written (prow_v), this workgroup = one binning block
        static_assert(!NEXT || BLOCK == BIN_BLOCK, "the tail bins one block of BIN_BLOCK splats");
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x == 0) *next.pre_valid = 1;
        preprocess_block<false, false>(next, prow_v, own_flags, i, sched_scratch);
    }
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

}  // namespace gfl

using namespace gfl;

extern "C" {

static inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// other translation units
size_t gfl_loss_workspace_bytes(int W, int H);

static inline int fit_nblk(int N) { return (N + BIN_BLOCK - 1) / BIN_BLOCK; }

// GFL_EWA_MFMA=1: the measured alternative for the J Sigma J^T contraction (fused_preprocess_fwd_kernel<true>)
static bool ewa_on_mfma() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_EWA_MFMA");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

int gfl_ewa_on_mfma(void) { return ewa_on_mfma() ? 1 : 0; }

// GFL_CAMERA_TAIL=1: the camera / depth-affine step as the ticketed tail of the per-splat launch (camera_tail) instead of
// a launch of its own.  Measured (round 3, one box, alternating): per-splat launch 22.3 -> 29.5 us with the tail against
// 22.3 + a 5-7 us launch without it, step 0.2100 against 0.2088 ms: the serial chain (agent-scope ticket, agent-scope
// loads of the rows, fold, pose chain rule) costs the same wherever it runs.  Off by default.
static bool camera_own_launch() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_CAMERA_TAIL");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// XCD-local bands + LPT in rounds (gfl_sched.hpp: schedule_tiles_xcd) -- the default; GFL_SCHED_XCD=0: the batched LPT over
// all queues of rounds 1-2 (schedule_tiles)
static int sched_xcd() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_SCHED_XCD");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

// GFL_BWD_ROTATE=1: the segments of a heavy first tile turn the block plan by one SIMD each, and the scheduler counts such a
// tile's blocks a quarter each on every SIMD.  Built on the trace's finding that a CU ends with its busiest SIMD (a pile
// in one 8x8 block puts 8 segments x 150 units on one SIMD); measured on one box, alternating, 4-frame clip fits: median
// 0.4852 / 0.4869 s with it against 0.4782 / 0.4838 s without, bench step 0.2061 against 0.2057 ms.  Off.
static int bwd_rotate() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_BWD_ROTATE");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v;
}

// GFL_SCHED_NEXT=0: the tile queues are built in line by the scatter launch in every iteration (rounds 1-2), not at the end
// of the previous iteration
static bool next_sched_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_SCHED_NEXT");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// GFL_BWD_GEOM_ONLY=0: every stage runs the full backward blend (sums that nobody reads formed, reduced and dropped)
static bool bwd_geom_only() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_BWD_GEOM_ONLY");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// GFL_POSE_FROZEN_FAST=0: iterations that do not move the camera keep the camera launch and the pose gradient (rounds 1-3)
static bool pose_frozen_fast() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_POSE_FROZEN_FAST");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// GFL_SORT_ORDER=0: the tile sort takes the tiles in their own order (rounds 1-3), not every XCD's longest lists first
static bool sort_heavy_first() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_SORT_ORDER");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

// list length from which the forward blend walks a queue's first tile as four blocks (GFL_FWD_SPLIT_MIN overrides)
static int fwd_split_min() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_FWD_SPLIT_MIN");
        v = e ? atoi(e) : FWD_SPLIT_MIN;
    }
    return v;
}

// one tile queue per CU (the dispatcher places workgroup b on CU b % CUs, tools/placement_probe.hip)
static int blend_queues() {
    static int nq = 0;
    if (!nq) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        nq = cus < 64 ? 64 : (cus < SCHED_MAX_QUEUES ? cus : SCHED_MAX_QUEUES);
    }
    return nq;
}

// workgroups of a blend launch: up to 8 per queue (all resident), fewer for small tile grids
// (GFL_FWD_WG / GFL_BWD_WG: fewer workgroups per CU than fit -- the rest of a queue is then pulled as workgroups finish)
static int env_wg(const char* name, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    return v > 0 && v < dflt ? v : dflt;
}

static int blend_grid(int T, int max_per_cu = BLEND_WG_PER_CU) {
    static const int fwd_wg = env_wg("GFL_FWD_WG", FWD_WG_PER_CU), bwd_wg = env_wg("GFL_BWD_WG", BLEND_WG_PER_CU);
    max_per_cu = max_per_cu == FWD_WG_PER_CU ? fwd_wg : bwd_wg;
    const int nq = blend_queues();
    int per = (T + nq - 1) / nq + 1;
    if (per > max_per_cu) per = max_per_cu;
    return nq * per;
}

// GFL_BWD_ROWS=1: the "rows" formulation of the backward blend (fused_blend_bwd_rows_kernel: a step = sixteen splats x
// four pixels, recurrences as row scans, no cross-lane reduction per splat) instead of the kernel of rounds 1-3.  Parity-
// green, measured in round 4 on the bench window (K = 264 k): 127.5 us against 71 us -- 30.8 M VALU instructions per launch
// against 32.2 M (its steps are 70 % full and cost ~290 instructions each) at 39 % VALU busy against 76 %.  Off by default.
static bool bwd_rows() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_BWD_ROWS");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}
int gfl_bwd_rows_on(void) { return bwd_rows() ? 1 : 0; }

size_t gfl_fit_workspace_bytes(int cap, int K_cap, int W, int H) {
    if (cap < 0 || K_cap < 0 || W <= 0 || H <= 0) return 0;
    const size_t T = (size_t)((W + GFL_TILE - 1) / GFL_TILE) * ((H + GFL_TILE - 1) / GFL_TILE);
    return up256((size_t)fit_nblk(cap > 0 ? cap : 1) * T * sizeof(int32_t))      // hist / bases
           + up256((size_t)K_cap * sizeof(unsigned long long))                      // keys
           + up256((size_t)reduce_rows(cap > 0 ? cap : 1) * 12 * sizeof(float))    // extr partials
           + up256(T * sizeof(int32_t))                                             // tile totals
           + up256((size_t)K_cap * PG * sizeof(float))                             // per-pair gradient rows
           + up256((size_t)(cap > 0 ? cap : 1) * SLOT_MAX * sizeof(int32_t))        // slot -> list position
           + 256 + up256((size_t)K_cap * sizeof(int32_t))                          // counters + slot pool
           + up256(4 * T * sizeof(int32_t))                                         // scheduler: work feedback per 8x8 block
           + up256((size_t)SCHED_MAX_QUEUES * sched_queue_capacity((int)T, 64) * sizeof(int32_t))   // queue items
           + 2 * up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t))                      // queue lengths, pull counters
           + up256((size_t)SCHED_MAX_QUEUES * (HEAVY_PARTS - 1) * 5 * 256 * sizeof(float))                  // heavy-tile checkpoints
           + up256(4 * T * sizeof(int32_t)) + up256(T * sizeof(int32_t))            // forward schedule: work feedback; first_slot
           + up256((size_t)SCHED_MAX_QUEUES * sched_queue_capacity((int)T, 64) * sizeof(int32_t))   // ... queue items
           + up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t))                          // ... queue lengths
           + up256(gfl_loss_workspace_bytes(W, H)) + 256
           + up256((size_t)6 * W * H * sizeof(float))                                  // SSIM statistics of the target
           + up256((size_t)fit_nblk(cap > 0 ? cap : 1) * sizeof(int32_t))              // rows of the scale term per block
           + up256((T * 4 + SORT_ORDER_TRAILER) * sizeof(int32_t))                     // the tile sort's order (+ its split list)
           + up256((size_t)K_cap * sizeof(int32_t))                                    // second slot pool   } iterations take
           + up256((size_t)fit_nblk(cap > 0 ? cap : 1) * sizeof(int32_t))              // second scale rows  } turns ("next preprocess")
           + up256((T * 4 + SORT_ORDER_TRAILER) * sizeof(int32_t))                     // reserved tile regions: the next sort order,
           + up256(T * sizeof(int4)) + up256(T * sizeof(int32_t));                      //   {start, capacity, position} per tile, fill counters
}

struct FitWs {
    int32_t* hist;
    unsigned long long* keys;
    float* partial;
    int32_t* tile_counts;
    float* pair_grad;
    int32_t* slot_inv;
    int32_t* slot_pool;
    int32_t* slot_pool2;     // iterations take turns (parity): the tail of iteration i prepares the pool of iteration i + 1
    int32_t* pool_counter;   //   while other workgroups of the same launch still gather through the pool of iteration i
    int32_t* pre_valid;      // != 0: the last per-splat launch ran the next iteration's preprocess in its tail
    int32_t* sched_valid;    // != 0: the tile queues in the workspace were built at the end of the last iteration
    Sched sched;             // tile queues of the backward blend; sched.work persists between calls
    Sched sched_fwd;         // ... and of the forward blend (its own work feedback)
    float* ckpt;             // [queue][boundary][T a0 a1 a2 a3][256 pixels] forward state at the heavy tile's segment boundaries
    void* loss_ws;
    size_t loss_ws_bytes;
    float* gt_stats;         // [3][2][H][W] conv(y), conv(y^2) of the current target (gfl_fit_prepare_targets)
    int32_t* scale_cnt;      // [blocks of the preprocess launch] rows of the scale term (lambda_scale)
    int32_t* scale_cnt2;     //   (second set, by parity like slot_pool2)
    int4* sort_order;        // [T] {tile, start, end, 0}: the order the tile sort takes the tiles in (fused_scatter_kernel)
    // reserved tile regions (fused_preprocess_bin_kernel): written at the end of an iteration for the next one
    int4* sort_order_next;   // [T] {tile, start, capacity, split} + trailer
    int4* region;            // [T] {start, capacity, position in sort_order_next}
    int32_t* fill;           // [T] keys binned so far, by position
    int32_t* regions_valid;  // != 0: the three arrays above are those of the coming iteration
    int32_t* extent;         // one past the last list position of the last forward (exact path: the number of pairs)
    int32_t* extent_next;    // ... of the regions
};

static FitWs carve(const gfl_fit_state* st) {
    const size_t T = (size_t)((st->W + GFL_TILE - 1) / GFL_TILE) * ((st->H + GFL_TILE - 1) / GFL_TILE);
    char* p = (char*)st->workspace;
    FitWs w;
    w.hist = (int32_t*)p;
    p += up256((size_t)fit_nblk(st->cap > 0 ? st->cap : 1) * T * sizeof(int32_t));
    w.keys = (unsigned long long*)p;
    p += up256((size_t)st->K_cap * sizeof(unsigned long long));
    w.partial = (float*)p;
    p += up256((size_t)reduce_rows(st->cap > 0 ? st->cap : 1) * 12 * sizeof(float));
    w.tile_counts = (int32_t*)p;
    p += up256(T * sizeof(int32_t));
    w.pair_grad = (float*)p;
    p += up256((size_t)st->K_cap * PG * sizeof(float));
    w.slot_inv = (int32_t*)p;
    p += up256((size_t)(st->cap > 0 ? st->cap : 1) * SLOT_MAX * sizeof(int32_t));
    w.pool_counter = (int32_t*)p;
    w.sched_valid = w.pool_counter + 16;
    w.pre_valid = w.pool_counter + 24;
    w.regions_valid = w.pool_counter + 32;
    w.extent = w.pool_counter + 40;
    w.extent_next = w.pool_counter + 48;
    p += 256;
    w.slot_pool = (int32_t*)p;
    p += up256((size_t)st->K_cap * sizeof(int32_t));
    w.sched.work = (int32_t*)p;
    p += up256(4 * T * sizeof(int32_t));
    w.sched.list = (int32_t*)p;
    p += up256((size_t)SCHED_MAX_QUEUES * sched_queue_capacity((int)T, 64) * sizeof(int32_t));
    w.sched.count = (int32_t*)p;
    p += up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t));
    w.sched.counters = (int32_t*)p;
    p += up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t));
    w.ckpt = (float*)p;
    p += up256((size_t)SCHED_MAX_QUEUES * (HEAVY_PARTS - 1) * 5 * 256 * sizeof(float));
    w.sched.nq = blend_queues();
    // (the list is sized for 512 queues: with fewer queues each may hold more -- a band of the XCD-local schedule
    //  can have many more tiles than T / 8)
    w.sched.cap_q = (int)(((size_t)SCHED_MAX_QUEUES * sched_queue_capacity((int)T, 64)) / w.sched.nq);
    w.sched.split_min = 0;
    w.sched.xcd = sched_xcd();
    w.sched.rotate = 0;
    w.sched_fwd = w.sched;
    w.sched.rotate = bwd_rotate();
    w.sched_fwd.work = (int32_t*)p;
    p += up256(4 * T * sizeof(int32_t));
    w.sched_fwd.list = (int32_t*)p;
    p += up256((size_t)SCHED_MAX_QUEUES * sched_queue_capacity((int)T, 64) * sizeof(int32_t));
    w.sched_fwd.count = (int32_t*)p;
    p += up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t));
    w.sched.first_slot = (int32_t*)p;
    p += up256(T * sizeof(int32_t));
    w.sched_fwd.first_slot = nullptr;
    w.sched_fwd.split_min = fwd_split_min();
    w.loss_ws = p;
    w.loss_ws_bytes = up256(gfl_loss_workspace_bytes(st->W, st->H));
    w.gt_stats = (float*)((char*)p + w.loss_ws_bytes + 256);
    w.scale_cnt = (int32_t*)((char*)w.gt_stats + up256((size_t)6 * st->W * st->H * sizeof(float)));
    w.sort_order = (int4*)((char*)w.scale_cnt + up256((size_t)fit_nblk(st->cap > 0 ? st->cap : 1) * sizeof(int32_t)));
    w.slot_pool2 = (int32_t*)((char*)w.sort_order + up256((T * 4 + SORT_ORDER_TRAILER) * sizeof(int32_t)));
    w.scale_cnt2 = (int32_t*)((char*)w.slot_pool2 + up256((size_t)st->K_cap * sizeof(int32_t)));
    w.sort_order_next = (int4*)((char*)w.scale_cnt2 + up256((size_t)fit_nblk(st->cap > 0 ? st->cap : 1) * sizeof(int32_t)));
    w.region = (int4*)((char*)w.sort_order_next + up256((T * 4 + SORT_ORDER_TRAILER) * sizeof(int32_t)));
    w.fill = (int32_t*)((char*)w.region + up256(T * sizeof(int4)));
    return w;
}

// the next iteration's tile queues are built by two extra workgroups of the per-splat launch when the XCD-local scheduler
// can run on REDUCE_BLOCK threads (otherwise the scatter launch keeps building them in line, every iteration)
static bool next_sched_ok(const FitWs& w, int T) {
    return w.sched.xcd && T <= SCHED_PLAN_TILES && w.sched.nq % 8 == 0 && w.sched.nq / 8 <= 64 && w.sched.nq <= REDUCE_BLOCK &&
           next_sched_enabled();
}
static int next_sched_blocks(const FitWs& w, int T) { return next_sched_ok(w, T) ? 2 : 0; }
// dynamic LDS of a launch that carries scheduling workgroups: T ints of scratch (the launch's own histogram / cursors) and,
// for tile grids of up to SCHED_PLAN_TILES tiles, the block plans' SCHED_PLAN_TILES words behind them
static size_t sched_dyn_lds(int T) {
    return (size_t)T * sizeof(int32_t) + (T <= SCHED_PLAN_TILES ? (size_t)SCHED_PLAN_TILES * sizeof(uint32_t) : 0);
}
static size_t next_sched_lds(const FitWs& w, int T) { return next_sched_ok(w, T) ? sched_dyn_lds(T) : 0; }
static NextSched next_sched(const FitWs& w, int rows, int T) {
    NextSched ns;
    ns.rows = rows;
    ns.T = T;
    ns.tile_counts = w.tile_counts;
    ns.bwd = w.sched;
    ns.fwd = w.sched_fwd;
    ns.valid = w.sched_valid;
    ns.reserve = 0;
    ns.order_next = nullptr; ns.ro = ReserveOut{}; ns.regions_valid = nullptr; ns.pool_counter = nullptr;
    ns.pull_counters = nullptr; ns.n_pull = 0;
    return ns;
}

// GFL_RESERVED=0 switches the reserved tile regions off (every iteration takes the exact binning path).
static bool reserved_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_RESERVED");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}
static bool next_pre_enabled();
// Reserved tile regions: grids of up to 4096 tiles (the region workgroup holds a tile's count in registers: 16 per lane on
// REDUCE_BLOCK lanes; the binning kernel 8 per lane on BIN_BLOCK), the tile sort's order in use, the scheduling workgroups
// in the per-splat launch (the region workgroup is the third of them), not together with the "next preprocess" experiment.
static bool fit_reserved_ok(const FitWs& w, int T) {
    return reserved_enabled() && !next_pre_enabled() && next_sched_ok(w, T) && T <= 4096 && sort_heavy_first() &&
           (2 * (size_t)T + 64) * sizeof(int32_t) <= 57 * 1024;
}
static NextSched next_sched_reserving(const gfl_fit_state* st, const FitWs& w, int rows, int T) {
    NextSched ns = next_sched(w, rows, T);
    if (!fit_reserved_ok(w, T)) return ns;
    ns.reserve = 1;
    ns.order_next = w.sort_order_next;
    ns.ro.region = w.region; ns.ro.fill = w.fill; ns.ro.extent_next = w.extent_next; ns.ro.total = st->tile_offsets + T;
    ns.ro.overflow = st->overflow; ns.ro.K_cap = st->K_cap;
    ns.regions_valid = w.regions_valid;
    ns.pool_counter = w.pool_counter;
    ns.pull_counters = w.sched.counters; ns.n_pull = 2 * w.sched.nq;
    return ns;
}

static int fit_check(const gfl_fit_state* st, const gfl_fit_hyper* hp) {
    if (!st || !hp) return GFL_ERR_INVALID;
    if (st->N < 0 || st->N > st->cap || st->W <= 0 || st->H <= 0 || st->K_cap < 0) return GFL_ERR_INVALID;
    if (!st->params || !st->rec || !st->d_rec || !st->pose || !st->intr || !st->extr || !st->render || !st->final_T ||
        !st->n_contrib || !st->tile_offsets || !st->ids || !st->tile_range || !st->overflow || !st->workspace)
        return GFL_ERR_INVALID;
    if (st->workspace_bytes < gfl_fit_workspace_bytes(st->cap, st->K_cap, st->W, st->H)) return GFL_ERR_WORKSPACE;
    return GFL_OK;
}

// kernels of gfl_bin.hip / gfl_loss.hip reused through their C entry points
// (gfl_loss_fwd_bwd, gfl_tile_sort_only: declared in gflow_hip.h)

// GFL_NEXT_PRE=1: between two plain iterations of one gfl_fit_iterations call the next preprocess runs in the tail of the
// per-splat launch (round 4).  Records and lists are bit-identical to the stand-alone launch; measured NEUTRAL on the bench
// window (the per-splat launch needs the binning's 512-splat workgroups for it: 22.0 -> 25.2 us, + 7.4 us of tail, against
// the 10.8 us launch it replaces), and the two instantiations of the per-splat kernel round their chain rule differently
// in the last bit, which Adam's first steps amplify: fits that group their iterations differently no longer agree to the
// order of the LDS adds.  Off by default.
static bool next_pre_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_NEXT_PRE");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

static PreArgs pre_args(const gfl_fit_state* st, const gfl_fit_hyper* hp, const FitWs& w, int gx, int gy, int op_mode,
                        int parity) {
    PreArgs a;
    a.intr = st->intr; a.pose = st->pose;
    a.N = st->N; a.W = st->W; a.H = st->H;
    a.nearest = hp->nearest; a.extent = hp->extent;
    a.gx = gx; a.gy = gy;
    a.rec = st->rec; a.slot_inv = w.slot_inv; a.hist_g = w.hist; a.extr_out = st->extr; a.overflow = st->overflow;
    a.slot_pool = parity ? w.slot_pool2 : w.slot_pool; a.pool_counter = w.pool_counter; a.pool_cap = st->K_cap;
    a.op_mode = op_mode;
    a.scale_rows_mode = (!op_mode && hp->lambda_scale != 0.f) ? (hp->freeze_all_splats ? 2 : 1) : 0;
    a.scale_cnt = parity ? w.scale_cnt2 : w.scale_cnt;
    a.pre_valid = w.pre_valid;
    return a;
}

// pre_done: this forward's preprocess has been run by the previous iteration's per-splat launch (its tail).
// parity: which of the two slot pools / scale-row sets this iteration uses.
// reserved: the tile regions the previous iteration's last launch reserved are used (one launch instead of three).
static int fit_forward_impl(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream, int op_mode,
                            int pre_done = 0, int parity = 0, int reserved = 0) {
    int rc = fit_check(st, hp);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (st->W + GFL_TILE - 1) / GFL_TILE, gy = (st->H + GFL_TILE - 1) / GFL_TILE, T = gx * gy;
    const FitWs w = carve(st);
    const int nblk = fit_nblk(st->N > 0 ? st->N : 1);
    const size_t lds = (size_t)T * sizeof(int32_t);
    // tile grid too large for the LDS histogram: 64 KB less the scheduling workgroups' ~6 KB of static state (their block
    // plans' 16 KB exist only for grids of up to 4096 tiles).  14 592 tiles: 2560 x 1440 has 14 400.
    if (lds > 57 * 1024) return GFL_ERR_INVALID;
    int32_t* slot_pool = parity ? w.slot_pool2 : w.slot_pool;
    if (reserved) {
        if (pre_done || op_mode || !fit_reserved_ok(w, T)) return GFL_ERR_INVALID;
        {
            StageScope p(ST_PREPROCESS, s);
            BinArgs b;
            b.region = w.region; b.fill = w.fill; b.keys = w.keys; b.K_cap = st->K_cap;
            b.regions_valid = w.regions_valid; b.extent_next = w.extent_next; b.extent = w.extent;
            b.pull_counters = w.sched.counters; b.n_pull = 2 * w.sched.nq;
            auto kern = ewa_on_mfma() ? fused_preprocess_bin_kernel<true> : fused_preprocess_bin_kernel<false>;
            kern<<<nblk, BIN_BLOCK, 2 * lds + 64 * sizeof(int32_t), s>>>(st->params, pre_args(st, hp, w, gx, gy, op_mode, parity), st->row_flags, b);
        }
        {
            StageScope p(ST_TILE_SORT, s);
            rc = gfl_tile_sort_reserved((const int32_t*)w.sort_order_next, w.fill, w.tile_counts, st->overflow + 2, st->W, st->H,
                                        st->K_cap, w.keys, st->ids, st->tile_range, st->rec, w.slot_inv, slot_pool, stream);
        }
        if (rc) return rc;
    }
    if (!pre_done && !reserved) {
        StageScope p(ST_PREPROCESS, s);
        auto kern = ewa_on_mfma() ? fused_preprocess_fwd_kernel<true> : fused_preprocess_fwd_kernel<false>;
        kern<<<nblk, BIN_BLOCK, lds, s>>>(st->params, pre_args(st, hp, w, gx, gy, op_mode, parity), st->row_flags);
    }
    if (!reserved) {
        StageScope p(ST_COLSCAN, s);
        bin_colscan_kernel<<<(T + CS_TILES - 1) / CS_TILES, 256, 0, s>>>(w.hist, nblk, T, w.tile_counts, w.pool_counter,
                                                                         w.sched.counters, 2 * w.sched.nq, w.pre_valid,
                                                                         pre_done, st->overflow);
    }
    // (the order of the tile sort is built by one workgroup with up to eight tiles per lane in registers: beyond 4096 tiles
    //  -- 1080p has 8160 -- the sort takes the tiles in their own order)
    const bool ordered = sort_heavy_first() && T <= 8 * BIN_BLOCK;
    if (!reserved) {
        StageScope p(ST_SCATTER, s);
        fused_scatter_kernel<<<nblk + 2 + (ordered ? 1 : 0), BIN_BLOCK, sched_dyn_lds(T), s>>>(st->rec, st->N, gx, gy, w.hist, w.tile_counts,
                                                              st->tile_offsets, st->K_cap, w.keys, st->overflow, w.sched, w.sched_fwd,
                                                              w.sched_valid, ordered ? w.sort_order : nullptr, w.extent);
    }
    if (!reserved) {
        StageScope p(ST_TILE_SORT, s);
        if (ordered)
            rc = gfl_tile_sort_ordered((const int32_t*)w.sort_order, st->W, st->H, st->K_cap, w.keys, st->ids, st->tile_range,
                                       st->rec, w.slot_inv, slot_pool, stream);
        else
            rc = gfl_tile_sort_with_slots(st->tile_offsets, st->W, st->H, st->K_cap, w.keys, st->ids, st->tile_range, st->rec,
                                          w.slot_inv, slot_pool, stream);
    }
    if (rc) return rc;
    {
        StageScope p(ST_BLEND_FWD, s);
        const TileQueue q = {w.sched_fwd.list, w.sched_fwd.count, w.sched.counters, w.sched.nq, w.sched.cap_q, 0};
        fused_blend_fwd_kernel<<<blend_grid(T, FWD_WG_PER_CU), 256, 0, s>>>(st->rec, st->ids, st->tile_range, hp->bg, st->W, st->H, gx,
                                                             st->render, st->final_T, st->n_contrib, q, w.ckpt, 0, nullptr,
                                                             nullptr, fwd_split_min(), w.sched_fwd.work, w.sched.first_slot);
        if (st->foot_flags) {
            // keep is in/out here: the footprint of this iteration's flagged splats is cleared from it, so it
            // carries the running union over the iterations of the stage exactly like the reference, which
            // rebinds move_mask = move_gs_mask | move_mask inside its loop (trainer.py:451).  The caller
            // initialises keep = !move_mask (all zero for a non-black background, where every pixel of the
            // extra render is > 0).
            // (Round 4: the footprint needs the sorted lists, not the render -- launched BESIDE the forward blend on a second
            //  stream, forked and joined with events inside the captured graph, 4-frame clip fits took 0.455-0.456 s against
            //  0.445-0.455 s with it behind the forward: like the snapshot before it, a fork inside a graph does not pay.)
            if (!st->keep) return GFL_ERR_INVALID;
            if (!(hp->bg > 0.f) && st->N > 0)
                footprint_kernel<<<T, 256, 0, s>>>(st->rec, st->ids, st->tile_range, st->foot_flags, st->W, st->H, gx,
                                                   st->keep);
        }
    }
    return check_launch();
}

int gfl_fit_forward(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream) {
    return fit_forward_impl(st, hp, stream, 0);
}

namespace gfl {
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
            const float x = fminf(fmaxf(src[k][(size_t)ch * P + i], 0.f), 1.f) * 255.f;
            out[((size_t)k * P + i) * 3 + ch] = (uint8_t)(x != x ? 0.f : x);
        }
    }
}
}  // namespace gfl

size_t gfl_fit_snapshot_workspace_bytes(int N, int W, int H) {
    if (N < 0 || W <= 0 || H <= 0) return 0;
    return 256 + up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t)) + 2 * up256((size_t)4 * W * H * sizeof(float)) +
           up256((size_t)W * H * sizeof(float)) + up256((size_t)W * H * sizeof(int32_t));
}

int gfl_fit_snapshot(const gfl_fit_state* st, const gfl_fit_hyper* hp, const float* lut, uint8_t* out_u8, void* workspace,
                     size_t workspace_bytes, gfl_stream_t stream) {
    int rc = fit_check(st, hp);
    if (rc) return rc;
    if (!lut || !out_u8 || !workspace) return GFL_ERR_INVALID;
    if (workspace_bytes < gfl_fit_snapshot_workspace_bytes(st->N, st->W, st->H)) return GFL_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (st->W + GFL_TILE - 1) / GFL_TILE, gy = (st->H + GFL_TILE - 1) / GFL_TILE, T = gx * gy;
    const int P = st->W * st->H;
    const FitWs w = carve(st);
    char* p = (char*)workspace;
    // head of the workspace, cleared by ONE memset: the depth range's two words and a set of queue pull counters for each
    // of the two composites (four memset launches of ~6 us each before: a snapshot every tenth iteration)
    unsigned* mm = (unsigned*)p;                 p += 256;
    int32_t* pull = (int32_t*)p;                 p += up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t));
    const size_t head_bytes = (size_t)(p - (char*)workspace);
    float* img_dc = (float*)p;                   p += up256((size_t)4 * P * sizeof(float));
    float* img_c = (float*)p;                    p += up256((size_t)4 * P * sizeof(float));
    float* fT = (float*)p;                       p += up256((size_t)P * sizeof(float));
    int32_t* nc = (int32_t*)p;
    rc = check(hipMemsetAsync(workspace, 0, head_bytes, s));
    if (rc) return rc;
    // (few blocks: a thousand waves hitting the two result words with atomics took 23 us)
    if (st->N > 0) rec_depth_range_kernel<<<min((st->N + 255) / 256, 32), 256, 0, s>>>(st->rec, st->N, mm);
    for (int mode = 1; mode <= 2; ++mode) {
        // (the forward launch of the iteration used up the engine's own pull counters)
        const TileQueue q = {w.sched_fwd.list, w.sched_fwd.count, pull + (mode - 1) * SCHED_MAX_QUEUES, w.sched.nq, w.sched.cap_q, 0};
        // (fewer workgroups per CU for these two launches, so that they disturb the fit's own kernels less, was measured in
        //  round 4: one per CU 0.871-0.886 s per 8-frame clip fit against 0.858-0.865 with five, three the same as five)
        fused_blend_fwd_kernel<<<blend_grid(T, FWD_WG_PER_CU), 256, 0, s>>>(st->rec, st->ids, st->tile_range, hp->bg, st->W, st->H,
                                                                            gx, mode == 1 ? img_dc : img_c, fT, nc, q, w.ckpt,
                                                                            mode, mm, lut, fwd_split_min(), w.sched_fwd.work,
                                                                            w.sched.first_slot);
    }
    snapshot_u8_kernel<<<(P + 255) / 256, 256, 0, s>>>(st->render, img_dc, img_c, P, out_u8);
    return check_launch();
}

}  // extern "C"

namespace gfl {
// Everything gfl_fit_snapshot reads of a forward -- records, sorted ids, tile ranges, the rgb planes of the render, the
// forward's tile queues -- copied from one engine to another in ONE launch (gfl_fit_snapshot_stage).  The number of ids
// is a device value (tile_offsets[T]).
struct StageSeg { const uint32_t* src; uint32_t* dst; unsigned n; };      // n: 32-bit words
struct StageCopy { StageSeg seg[7]; const int32_t* k_ptr; unsigned ids_cap; };
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
}  // namespace gfl

extern "C" {

int gfl_fit_snapshot_stage(const gfl_fit_state* src, const gfl_fit_state* dst, gfl_stream_t stream) {
    if (!src || !dst || !src->workspace || !dst->workspace || !src->rec || !dst->rec || !src->ids || !dst->ids ||
        !src->tile_range || !dst->tile_range || !src->render || !dst->render || !src->tile_offsets)
        return GFL_ERR_INVALID;
    if (src->W != dst->W || src->H != dst->H || src->N != dst->N || dst->N > dst->cap || src->N > src->cap)
        return GFL_ERR_INVALID;
    if (src->workspace_bytes < gfl_fit_workspace_bytes(src->cap, src->K_cap, src->W, src->H) ||
        dst->workspace_bytes < gfl_fit_workspace_bytes(dst->cap, dst->K_cap, dst->W, dst->H))
        return GFL_ERR_WORKSPACE;
    const int gx = (src->W + GFL_TILE - 1) / GFL_TILE, gy = (src->H + GFL_TILE - 1) / GFL_TILE, T = gx * gy;
    const size_t P = (size_t)src->W * src->H;
    const FitWs a = carve(src), b = carve(dst);
    if (a.sched_fwd.nq != b.sched_fwd.nq || a.sched_fwd.cap_q != b.sched_fwd.cap_q) return GFL_ERR_INVALID;
    StageCopy c;
    c.k_ptr = a.extent;
    c.ids_cap = (unsigned)min(src->K_cap, dst->K_cap);
    c.seg[0] = {(const uint32_t*)src->ids, (uint32_t*)dst->ids, 0u};
    c.seg[1] = {(const uint32_t*)src->rec, (uint32_t*)dst->rec, (unsigned)((size_t)src->N * REC)};
    c.seg[2] = {(const uint32_t*)src->tile_range, (uint32_t*)dst->tile_range, (unsigned)(2 * T)};
    c.seg[3] = {(const uint32_t*)src->render, (uint32_t*)dst->render, (unsigned)(3 * P)};
    c.seg[4] = {(const uint32_t*)a.sched_fwd.list, (uint32_t*)b.sched_fwd.list,
                (unsigned)((size_t)a.sched_fwd.nq * a.sched_fwd.cap_q)};
    c.seg[5] = {(const uint32_t*)a.sched_fwd.count, (uint32_t*)b.sched_fwd.count, (unsigned)a.sched_fwd.nq};
    // (first_slot -- which queue holds a tile as its first item: where the forward kernel leaves its checkpoints)
    c.seg[6] = {(const uint32_t*)a.sched.first_slot, (uint32_t*)b.sched.first_slot, (unsigned)T};
    snapshot_stage_kernel<<<1024, 256, 0, (hipStream_t)stream>>>(c);
    return check_launch();
}

int gfl_render_fwd(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream) {
    if (st && st->foot_flags) return GFL_ERR_INVALID;
    return fit_forward_impl(st, hp, stream, 1);
}

int gfl_render_bwd(const gfl_fit_state* st, const gfl_fit_hyper* hp, const float* d_render, const float* d_uv,
                   const float* d_depth, float* d_params, float* d_extr, gfl_stream_t stream) {
    int rc = fit_check(st, hp);
    if (rc) return rc;
    if (!d_render || !d_params || !d_extr) return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (st->W + GFL_TILE - 1) / GFL_TILE, gy = (st->H + GFL_TILE - 1) / GFL_TILE, T = gx * gy;
    const FitWs w = carve(st);
    {
        StageScope p(ST_BLEND_BWD, s);
        const TileQueue q = {w.sched.list, w.sched.count, w.sched.counters + w.sched.nq, w.sched.nq, w.sched.cap_q, w.sched.rotate};
        fused_blend_bwd_kernel<10><<<blend_grid(T), 256, 0, s>>>(st->rec, st->ids, st->tile_range, hp->bg, st->W, st->H, gx,
                                                                    st->final_T, st->n_contrib, d_render, w.pair_grad, q,
                                                                    w.sched.work, w.ckpt, st->render, LossTail{});
    }
    const int rows = reduce_rows(st->N > 0 ? st->N : 1);
    RegCfg rcfg = {};
    AdamCfg ac = {};
    {
        StageScope p(ST_PRE_BWD_ADAM, s);
        const NextSched ns = next_sched(w, rows, T);
        fused_preprocess_bwd_adam_kernel<true, REDUCE_BLOCK, false><<<rows + next_sched_blocks(w, T), REDUCE_BLOCK, next_sched_lds(w, T), s>>>(
            st->params, nullptr, nullptr, st->intr, st->extr, st->rec, st->d_rec, w.pair_grad, w.slot_pool, st->tile_range,
            w.slot_inv, gx, gy, st->N, st->W, st->H, nullptr, nullptr, nullptr, nullptr, nullptr, rcfg, ac, nullptr,
            w.partial, d_uv, d_depth, d_params, nullptr, CamTail{}, ns, PreArgs{}, nullptr);
        fold_partials_kernel<12><<<1, 256, 0, s>>>(w.partial, rows, d_extr);
    }
    return check_launch();
}

// do_next: run the next iteration's preprocess in the tail of the per-splat launch (only honoured when the iteration
// qualifies: fit_next_pre_ok); parity: the slot pool / scale rows of THIS iteration (the next one gets the other set)
static bool fit_next_pre_ok(const gfl_fit_state* st, const gfl_fit_hyper* hp, const FitWs& w, int T);

static int fit_backward_step_impl(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream, int do_next,
                                  int parity) {
    int rc = fit_check(st, hp);
    if (rc) return rc;
    if (!st->adam_m || !st->adam_v || !st->pose_m || !st->pose_v || !st->depth_ab || !st->depth_ab_m ||
        !st->depth_ab_v || !st->step || !st->gt_rgb || !st->d_render || !st->err_px || !st->sums || !st->d_extr)
        return GFL_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (st->W + GFL_TILE - 1) / GFL_TILE, gy = (st->H + GFL_TILE - 1) / GFL_TILE, T = gx * gy;
    const FitWs w = carve(st);
    const float *p_ssim = nullptr, *p_grad = nullptr;
    int n_ssim = 0, n_grad = 0;
    {
        StageScope p(ST_LOSS, s);
        // the footprint mask changes keep (and with it the masked target) every iteration
        const float* gt_stats = (st->gt_cached && !st->foot_flags) ? w.gt_stats : nullptr;
        rc = gfl_loss_fwd_bwd_partials_cached(st->render, st->gt_rgb, st->gt_depth, st->keep, st->depth_ab,
                                              hp->lambda_rgb, hp->lambda_depth, st->W, st->H, st->d_render,
                                              st->err_px, w.loss_ws, w.loss_ws_bytes, gt_stats, &p_ssim, &n_ssim,
                                              &p_grad, &n_grad, stream);
    }
    if (rc) return rc;
    AdamCfg ac = {hp->lr, hp->beta1, hp->beta2, hp->eps, hp->lr_end_factor, hp->total_iters};
    AdamCfg ac_cam = ac;
    ac_cam.lr = hp->lr_camera;
    // The camera does not move in this iteration (see LossTail): no pose gradient, no camera launch.  step_camera = 2 asks
    // for the gradient (d_extr) although nothing is stepped with it; GFL_POSE_FROZEN_FAST=0 switches the short cut off.
    const bool frozen = pose_frozen_fast() && hp->step_camera != 2 && (hp->step_camera == 0 || hp->lr_camera == 0.f) &&
                        camera_own_launch();
    LossTail lt = {};
    if (frozen) {
        lt.enabled = 1;
        lt.p_ssim = p_ssim; lt.n_ssim = n_ssim; lt.p_grad = p_grad; lt.n_grad = n_grad;
        lt.depth_ab = st->depth_ab; lt.ab_m = st->depth_ab_m; lt.ab_v = st->depth_ab_v;
        lt.sums = st->sums; lt.ac_ab = ac; lt.step_affine = hp->step_camera != 0;
        lt.d_step = st->step; lt.d_extr_out = st->d_extr; lt.overflow = st->overflow;
    }
    {
        StageScope p(ST_BLEND_BWD, s);
        const TileQueue q = {w.sched.list, w.sched.count, w.sched.counters + w.sched.nq, w.sched.nq, w.sched.cap_q, w.sched.rotate};
        const int sums = !bwd_geom_only() ? 10 : (hp->freeze_all_splats ? 6 : (hp->freeze_rgb ? 7 : 10));
        if (bwd_rows()) {
            auto kern = sums == 6 ? fused_blend_bwd_rows_kernel<6> : (sums == 7 ? fused_blend_bwd_rows_kernel<7> : fused_blend_bwd_rows_kernel<10>);
            kern<<<blend_grid(T), 256, 0, s>>>(st->rec, st->ids, st->tile_range, hp->bg, st->W, st->H, gx, st->final_T,
                                                     st->n_contrib, st->d_render, w.pair_grad, q, w.sched.work, w.ckpt, st->render, lt);
        } else {
            auto kern = sums == 6 ? fused_blend_bwd_kernel<6> : (sums == 7 ? fused_blend_bwd_kernel<7> : fused_blend_bwd_kernel<10>);
            kern<<<blend_grid(T), 256, 0, s>>>(st->rec, st->ids, st->tile_range, hp->bg, st->W, st->H, gx, st->final_T, st->n_contrib,
                                               st->d_render, w.pair_grad, q, w.sched.work, w.ckpt, st->render, lt);
        }
    }
    const int rows = reduce_rows(st->N > 0 ? st->N : 1);
    RegCfg rcfg;
    rcfg.lambda_scale = hp->lambda_scale;
    rcfg.scale_blocks = fit_nblk(st->N > 0 ? st->N : 1);
    rcfg.lambda_var = st->N > 0 ? hp->lambda_var / (float)st->N : 0.f;
    rcfg.lambda_flow = hp->lambda_flow;
    rcfg.lambda_still = hp->lambda_still;
    rcfg.freeze_rgb = hp->freeze_rgb;
    rcfg.freeze_all = hp->freeze_all_splats;
    rcfg.no_pose_grad = frozen ? 1 : 0;
    // the camera / depth-affine step: a launch of its own, or (GFL_CAMERA_TAIL=1) the tail of the per-splat launch
    const bool own_launch = camera_own_launch();
    CamTail tail = {};
    if (!own_launch) {
        tail.p_ssim = p_ssim; tail.n_ssim = n_ssim; tail.p_grad = p_grad; tail.n_grad = n_grad;
        tail.pose = st->pose; tail.pose_m = st->pose_m; tail.pose_v = st->pose_v;
        tail.depth_ab = st->depth_ab; tail.ab_m = st->depth_ab_m; tail.ab_v = st->depth_ab_v;
        tail.sums = st->sums; tail.ac_cam = ac_cam; tail.ac_ab = ac; tail.step_camera = hp->step_camera;
        tail.d_step = st->step; tail.d_extr_out = st->d_extr; tail.ticket = w.pool_counter + 8; tail.overflow = st->overflow;
    }
    const int32_t* slot_pool = parity ? w.slot_pool2 : w.slot_pool;
    const int32_t* scale_cnt = parity ? w.scale_cnt2 : w.scale_cnt;
    if (do_next && frozen && fit_next_pre_ok(st, hp, w, T)) {
        // the per-splat launch with 512-splat workgroups (= the binning's blocks) and the next preprocess in its tail
        StageScope p(ST_PRE_BWD_ADAM, s);
        const int rows512 = fit_nblk(st->N > 0 ? st->N : 1);
        rcfg.scale_blocks = rows512;
        const NextSched ns = next_sched(w, rows512, T);
        const PreArgs next = pre_args(st, hp, w, gx, gy, 0, parity ^ 1);
        fused_preprocess_bwd_adam_kernel<false, BIN_BLOCK, true><<<rows512 + next_sched_blocks(w, T), BIN_BLOCK, sched_dyn_lds(T), s>>>(
            st->params, st->adam_m, st->adam_v, st->intr, st->pose, st->rec, st->d_rec, w.pair_grad, slot_pool,
            st->tile_range, w.slot_inv, gx, gy, st->N, st->W, st->H, st->flow_target, st->flow_w, st->still_target,
            st->still_w, st->row_flags, rcfg, ac, st->step, w.partial, nullptr, nullptr, nullptr, scale_cnt, tail, ns, next, st->overflow);
        return check_launch();
    }
    {
        StageScope p(ST_PRE_BWD_ADAM, s);
        const NextSched ns = next_sched_reserving(st, w, rows, T);
        fused_preprocess_bwd_adam_kernel<false, REDUCE_BLOCK, false><<<rows + next_sched_blocks(w, T) + ns.reserve, REDUCE_BLOCK, next_sched_lds(w, T), s>>>(
            st->params, st->adam_m, st->adam_v, st->intr, st->pose, st->rec, st->d_rec, w.pair_grad, slot_pool,
            st->tile_range, w.slot_inv, gx, gy, st->N, st->W, st->H, st->flow_target, st->flow_w, st->still_target,
            st->still_w, st->row_flags, rcfg, ac, st->step, w.partial, nullptr, nullptr, nullptr, scale_cnt, tail, ns, PreArgs{}, st->overflow);
    }
    if (own_launch && !frozen) {
        StageScope p(ST_CAMERA, s);
        fused_camera_adam_kernel<<<1, 1024, 0, s>>>(w.partial, rows, p_ssim, n_ssim, p_grad, n_grad, st->pose, st->pose_m,
                                                    st->pose_v, st->depth_ab, st->depth_ab_m, st->depth_ab_v, st->sums,
                                                    ac_cam, ac, hp->step_camera != 0, st->step, st->d_extr, st->overflow);
    }
    return check_launch();
}

int gfl_fit_backward_step(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream) {
    return fit_backward_step_impl(st, hp, stream, 0, 0);
}

// The iteration qualifies for the next preprocess in its tail: a plain fit iteration whose camera cannot move (the pose the
// tail projects with is the pose of the next iteration), splats that are stepped (not the camera-only stage), no
// footprint mask, the XCD scheduler's 512-thread form available for the launch's two scheduling workgroups.
static bool fit_next_pre_ok(const gfl_fit_state* st, const gfl_fit_hyper* hp, const FitWs& w, int T) {
    const bool frozen = pose_frozen_fast() && hp->step_camera != 2 && (hp->step_camera == 0 || hp->lr_camera == 0.f) &&
                        camera_own_launch();
    const bool sched512 = !next_sched_enabled() || (w.sched.xcd && T <= SCHED_PLAN_TILES && w.sched.nq % 8 == 0 &&
                                                    w.sched.nq / 8 <= 64 && w.sched.nq <= BIN_BLOCK);
    return next_pre_enabled() && frozen && !hp->freeze_all_splats && !st->foot_flags && !ewa_on_mfma() && sched512 &&
           st->N > 0 && (size_t)T * sizeof(int32_t) <= 57 * 1024;
}

int gfl_fit_iterations(const gfl_fit_state* st, const gfl_fit_hyper* hp, int count, int flags, gfl_stream_t stream) {
    if (count < 1 || (flags & ~(GFL_ITER_PRE_DONE | GFL_ITER_PRE_NEXT | GFL_ITER_ODD | GFL_ITER_RESERVED))) return GFL_ERR_INVALID;
    int rc = fit_check(st, hp);
    if (rc) return rc;
    const int T = ((st->W + GFL_TILE - 1) / GFL_TILE) * ((st->H + GFL_TILE - 1) / GFL_TILE);
    const FitWs w = carve(st);
    const bool ok = fit_next_pre_ok(st, hp, w, T);
    if ((flags & GFL_ITER_PRE_DONE) && !ok) return GFL_ERR_INVALID;
    const bool res_ok = fit_reserved_ok(w, T) && st->N > 0;
    if ((flags & GFL_ITER_RESERVED) && !res_ok) return GFL_ERR_INVALID;
    int parity = (flags & GFL_ITER_ODD) ? 1 : 0;
    for (int j = 0; j < count; ++j) {
        const int pre_done = j == 0 ? ((flags & GFL_ITER_PRE_DONE) ? 1 : 0) : (ok ? 1 : 0);
        const int do_next = ok && (j + 1 < count || (flags & GFL_ITER_PRE_NEXT));
        // (every iteration of a call but the first follows a full iteration: its tile regions are reserved)
        const int reserved = res_ok && (j > 0 || (flags & GFL_ITER_RESERVED)) ? 1 : 0;
        rc = fit_forward_impl(st, hp, stream, 0, pre_done, parity, reserved);
        if (rc) return rc;
        rc = fit_backward_step_impl(st, hp, stream, do_next, parity);
        if (rc) return rc;
        if (ok) parity ^= 1;
    }
    return GFL_OK;
}

int gfl_fit_reserved_supported(const gfl_fit_state* st, const gfl_fit_hyper* hp) {
    if (fit_check(st, hp) || st->N <= 0) return 0;
    const int T = ((st->W + GFL_TILE - 1) / GFL_TILE) * ((st->H + GFL_TILE - 1) / GFL_TILE);
    return fit_reserved_ok(carve(st), T) ? 1 : 0;
}

int gfl_fit_next_preprocess_supported(const gfl_fit_state* st, const gfl_fit_hyper* hp) {
    if (fit_check(st, hp)) return 0;
    const int T = ((st->W + GFL_TILE - 1) / GFL_TILE) * ((st->H + GFL_TILE - 1) / GFL_TILE);
    return fit_next_pre_ok(st, hp, carve(st), T) ? 1 : 0;
}

int gfl_fit_prepare_targets(const gfl_fit_state* st, gfl_stream_t stream) {
    if (!st || st->W <= 0 || st->H <= 0 || !st->gt_rgb || !st->workspace) return GFL_ERR_INVALID;
    if (st->workspace_bytes < gfl_fit_workspace_bytes(st->cap, st->K_cap, st->W, st->H)) return GFL_ERR_WORKSPACE;
    const FitWs w = carve(st);
    return gfl_loss_prepare_gt(st->gt_rgb, st->keep, st->W, st->H, w.gt_stats, stream);
}

int gfl_fit_schedule_info(const gfl_fit_state* st, int* n_queues, int* queue_capacity, const int32_t** d_lists,
                          const int32_t** d_counts) {
    if (!st || !n_queues || !queue_capacity || !d_lists || !d_counts || !st->workspace) return GFL_ERR_INVALID;
    if (st->workspace_bytes < gfl_fit_workspace_bytes(st->cap, st->K_cap, st->W, st->H)) return GFL_ERR_WORKSPACE;
    const FitWs w = carve(st);
    *n_queues = w.sched.nq;
    *queue_capacity = w.sched.cap_q;
    *d_lists = w.sched.list;
    *d_counts = w.sched.count;
    return GFL_OK;
}

int gfl_fit_schedule_info_fwd(const gfl_fit_state* st, int* n_queues, int* queue_capacity, const int32_t** d_lists,
                              const int32_t** d_counts) {
    if (!st || !n_queues || !queue_capacity || !d_lists || !d_counts || !st->workspace) return GFL_ERR_INVALID;
    if (st->workspace_bytes < gfl_fit_workspace_bytes(st->cap, st->K_cap, st->W, st->H)) return GFL_ERR_WORKSPACE;
    const FitWs w = carve(st);
    *n_queues = w.sched_fwd.nq;
    *queue_capacity = w.sched_fwd.cap_q;
    *d_lists = w.sched_fwd.list;
    *d_counts = w.sched_fwd.count;
    return GFL_OK;
}

int gfl_fit_iteration(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream) {
    int rc = gfl_fit_forward(st, hp, stream);
    if (rc) return rc;
    return gfl_fit_backward_step(st, hp, stream);
}

#ifdef GFL_TRACE
int gfl_debug_read_fwd_trace(long long* out, int n_tiles) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gfl::g_fwd_trace), (size_t)n_tiles * 8 * sizeof(long long));
}
int gfl_debug_read_fwd_trace2(long long* out, int n_values) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gfl::g_fwd_trace2), (size_t)n_values * sizeof(long long));
}
int gfl_debug_read_phase_trace(long long* out, int n_values) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gfl::g_phase_trace), (size_t)n_values * sizeof(long long));
}
int gfl_debug_read_bwd_trace(long long* out, int n_tiles) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gfl::g_bwd_trace), (size_t)n_tiles * 8 * sizeof(long long));
}
#endif

}  // extern "C"
