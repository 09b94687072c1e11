// Fused fit iteration -- what the stage files (gfl_fit_bin / _fwd / _bwd / _splat .hip) and the host file (gfl_fit.hip)
// share: the data layout's constants, the device functions more than one stage inlines (camera from pose, activations, the
// culling disc, the alpha test), the argument structs of the kernels, the carved workspace, and the launchers through which
// the host file reaches a stage's kernels (a kernel is launched from the translation unit that defines it).
//
// The whole of gflow/trainer.py:387-558 (activations, render, losses, backward, gradient masking, Adam) is ~6-9 kernel
// launches issued by ONE library call, with no host read-back and every launch sized by N, T or the image (never by the
// data-dependent pair count K) -> graph-capturable.
//
// Data layout in HBM (288 GB: capacity-based, nothing is reallocated when N grows)
//   params / adam_m / adam_v : [cap][16] f32, one 64-byte row per splat
//        x y z | sx sy sz | qw qx qy qz | opacity | r g b | pad pad      (raw values)
//   rec   : [cap][12] f32, what a pixel needs from a splat (3 x 16-byte loads)
//        u v A B | C opacity r g | b depth cutoff radius(int bits)
//   d_rec : [cap][12] f32, gradient of the loss wrt the first 10 entries of rec
//   keys  : [K_cap] u64 (depth bits << 32 | id), ids : [K_cap] i32, tile_range [T][2]
//   hist  : [n_bin_blocks][T] i32 per-block tile histogram -> per-block base offsets
#pragma once
#include "gfl_math.hpp"
#include "gfl_profile.hpp"
#include "gfl_sched.hpp"

#include <stdlib.h>

namespace gfl {

constexpr int BIN_BLOCK = 512;
// splats per workgroup of the one-launch binning on reserved tile regions (fused_preprocess_bin_kernel).  That launch is a chain of
// latencies with ONE splat per lane -- 60 000 splats are 940 waves for 1 024 SIMDs, and 512-lane workgroups put them two to a SIMD
// on 118 of the 256 CUs -- but 256-lane workgroups on 235 CUs are SLOWER: 23.0 us against 19.7 (tools/quick_trace.sh, round 5;
// every workgroup clears its own T-int histogram, requests the T regions and reserves its part of every tile it touches: that
// share of the launch doubles, the preprocess in between is a fifth of it).  -DGFL_RBIN_BLOCK=256 keeps the measurement repeatable.
#ifndef GFL_RBIN_BLOCK
#define GFL_RBIN_BLOCK 512
#endif
constexpr int RBIN_BLOCK = GFL_RBIN_BLOCK;
static_assert(RBIN_BLOCK == 256 || RBIN_BLOCK == 512, "two scale-row partials per 512 splats (PreArgs.scale_cnt)");
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
// (one copy of the table per translation unit: rows 0-2 live in gfl_fit_bin.hip, row 3 in gfl_fit_splat.hip)
#ifdef GFL_TRACE
constexpr int PHASE_WAVES = 4096;
static __device__ long long g_phase_trace[4 * PHASE_WAVES * 8];
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
    u = gridf(u); v = gridf(v);
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
// (and a barrier has passed): preprocess_block (gfl_fit_bin.hip).  Two callers: the stand-alone launch of the exact binning
// path (fused_preprocess_fwd_kernel) and the one-launch binning on reserved tile regions (fused_preprocess_bin_kernel).
// A block: splats [blockIdx.x * BIN_BLOCK, + BIN_BLOCK) (the scatter re-walks the same blocks).
// EWA_MFMA: the J Sigma J^T contraction on the matrix cores (cov2d_mfma, gfl_math.hpp) instead of 30 FMAs in the lane
// -- the variant north_star names; selected with GFL_EWA_MFMA=1, measured in DESIGN.md section 4 (docs/history.md section 4 for the full account), off by default.
struct PreArgs {
    const float* intr; const float* pose;
    int N, W, H;
    float nearest, extent;
    int gx, gy;
    float* rec; int32_t* wide_off; int32_t* hist_g; float* extr_out; int32_t* overflow;
    int32_t* pool_counter; int pool_cap; int op_mode;
    int scale_rows_mode; int32_t* scale_cnt;
};

// BINNED (reserved tile regions, fused_preprocess_bin_kernel): the block's histogram stays in LDS -- the caller reserves the
// block's part of every tile's region with it and scatters the keys itself -- and what the scatter needs of the splat comes
// back in `po`.
struct PreOut { float u, v, cutoff, depth; int rad; };
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
// what the region-reserving workgroup at the end of an iteration writes for the next one (build_sort_order<.., true>)
struct ReserveOut { int4* region; int32_t* fill; int32_t* extent_next; int32_t* total; int K_cap; };

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

// "every lane of the workgroup says yes" with ONE barrier and no dependence on the block's shape (HIP's __syncthreads_and reads
// blockDim / threadIdx.y for a flat thread id: loop-invariant values the forward blend then carried through its tile loop --
// in scratch, 16 B per lane).  Two LDS counters that only grow, used in turn: a wave cannot reach its add of round r + 2
// before every wave has read round r's total (the barrier of round r + 1 lies between).  waves = blockDim.x / 64.
struct WgVote {
    int32_t* cnt;        // [2] in LDS, zeroed once before the first vote (behind a barrier)
    int base0, base1, round;
};
__device__ __forceinline__ WgVote wg_vote_init(int32_t* lds2) {
    WgVote v;
    v.cnt = lds2; v.base0 = 0; v.base1 = 0; v.round = 0;
    return v;
}
__device__ __forceinline__ bool wg_all(WgVote& v, bool pred, int waves) {
    const int r = v.round & 1;
    if (__all(pred) && (threadIdx.x & 63) == 0) atomicAdd(&v.cnt[r], 1);
    __syncthreads();
    const int now = v.cnt[r];
    const bool all = now - (r ? v.base1 : v.base0) == waves;
    if (r) v.base1 = now; else v.base0 = now;
    ++v.round;
    return all;
}

// tile -> (tx, ty) without an integer division: inv_gx = 2^32 / gx rounded up (exact for tile < 2^16, which every tile id
// in a queue item is)
__device__ __forceinline__ void tile_xy(int tile, int gx, unsigned inv_gx, int& tx, int& ty) {
    ty = (int)__umulhi((unsigned)tile, inv_gx);
    tx = tile - ty * gx;
}

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

// a float image value as render2img stores it (render.py:158-166): clamp to [0, 1], x 255, truncate
__device__ __forceinline__ uint8_t img_u8(float v) {
    const float x = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
    return (uint8_t)(x != x ? 0.f : x);
}

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

// ------------------------------------------------- preprocess backward + Adam (A13)
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

// ------------------------------------------------- the carved workspace (gfl_fit.hip: carve)
struct FitWs {
    int32_t* hist;
    unsigned long long* keys;
    float* partial;
    int32_t* tile_counts;
    float* pair_grad;
    // pair rows (round 5): the backward blend writes the gradient row of the pair (splat g, tile) at a place the per-splat
    // launch finds WITHOUT a table -- row g * SLOT_MAX + (the tile's index inside g's tile rectangle); splats covering more than
    // SLOT_MAX tiles (a handful per frame) get a run of the rows behind those, at wide_base + wide_off[g].  A row carries the
    // STAMP of the forward it belongs to in its eleventh float: rows of older iterations, and of pairs nobody walked (behind the
    // tile's deepest contributor, culled by the exact disc test), are simply not this iteration's.
    int32_t* wide_off;       // [cap] offset of a wide splat's run (written by the preprocess for wide splats only)
    long long wide_base;     // first row of the runs = cap * SLOT_MAX
    int32_t* stamp;          // the current forward's number (the forward blend's launch advances it)
    unsigned* snap_mm;       // a snapshot iteration's depth range (two words, cmap_nonzero_lookup)
    int32_t* pool_counter;   // rows of the runs handed out by this iteration's preprocess
    int32_t* sched_valid;    // != 0: the tile queues in the workspace were built at the end of the last iteration
    Sched sched;             // tile queues of the backward blend; sched.work persists between calls
    Sched sched_fwd;         // ... and of the forward blend (its own work feedback)
    float* ckpt;             // [queue][boundary][T a0 a1 a2 a3][256 pixels] forward state at the heavy tile's segment boundaries
    void* loss_ws;
    size_t loss_ws_bytes;
    float* gt_stats;         // [3][2][H][W] conv(y), conv(y^2) of the current target (gfl_fit_prepare_targets)
    int32_t* scale_cnt;      // [blocks of the preprocess launch] rows of the scale term (lambda_scale)
    int4* sort_order;        // [T] {tile, start, end, 0}: the order the tile sort takes the tiles in (fused_scatter_kernel)
    // reserved tile regions (fused_preprocess_bin_kernel): written at the end of an iteration for the next one
    int4* sort_order_next;   // [T] {tile, start, capacity, split} + trailer
    int4* region;            // [T] {start, capacity, position in sort_order_next}
    int32_t* fill;           // [T] keys binned so far, by position
    int32_t* regions_valid;  // != 0: the three arrays above are those of the coming iteration
    int32_t* extent;         // one past the last list position of the last forward (exact path: the number of pairs)
    int32_t* extent_next;    // ... of the regions
};

// dynamic LDS of a launch that carries scheduling workgroups: T ints of scratch (the launch's own histogram / cursors) and,
// for tile grids of up to SCHED_PLAN_TILES tiles, the block plans' SCHED_PLAN_TILES words behind them
inline size_t sched_dyn_lds(int T) {
    return (size_t)T * sizeof(int32_t) + (T <= SCHED_PLAN_TILES ? (size_t)SCHED_PLAN_TILES * sizeof(uint32_t) : 0);
}

// gfl_fit_snapshot_stage: what one launch copies from one engine to another (segment 0 = the ids, its length a device value)
struct StageSeg { const uint32_t* src; uint32_t* dst; unsigned n; };      // n: 32-bit words
struct StageCopy { StageSeg seg[7]; const int32_t* k_ptr; unsigned ids_cap; };

// ------------------------------------------------- launchers: a stage's kernels as seen from the host file
// gfl_fit_bin.hip
void launch_preprocess_fwd(const float* params, const PreArgs& a, const uint8_t* row_flags, int nblk, bool mfma, hipStream_t s);
void launch_preprocess_bin(const float* params, const PreArgs& a, const uint8_t* row_flags, const BinArgs& b, int nblk, bool mfma,
                           hipStream_t s);
void launch_colscan(const FitWs& w, int nblk, int T, int32_t* overflow, hipStream_t s);
void launch_scatter(const gfl_fit_state* st, const FitWs& w, int nblk, int gx, int gy, bool ordered, hipStream_t s);
// gfl_fit_fwd.hip
inline unsigned inv_of(int gx) { return (unsigned)((((unsigned long long)1 << 32) + (unsigned)gx - 1) / (unsigned)gx); }
void launch_blend_fwd(const gfl_fit_state* st, float bg, int gx, int grid, float* out, float* final_T, int32_t* n_contrib,
                      const TileQueue& q, const FitWs& w, int mode, const unsigned* cmap_mm, const float* cmap_lut, int split_min,
                      hipStream_t s, uint8_t* snap_u8 = nullptr);
void launch_footprint(const gfl_fit_state* st, int gx, int T, hipStream_t s);
void launch_center_blend(const gfl_fit_state* st, float bg, int gx, int T, float* out, uint8_t* out_u8, hipStream_t s);
void launch_rec_depth_range(const float* rec, int N, unsigned* mm, hipStream_t s);
void launch_snapshot_u8(const float* a, const float* b, const float* c, int P, uint8_t* out, hipStream_t s);
void launch_snapshot_stage(const StageCopy& c, hipStream_t s);
// gfl_fit_bwd.hip
void launch_blend_bwd(const gfl_fit_state* st, float bg, int gx, int gy, int grid, int sums, const float* d_out, const TileQueue& q,
                      const FitWs& w, const LossTail& lt, hipStream_t s);
// gfl_fit_splat.hip
void launch_splat_bwd_adam(const gfl_fit_state* st, const FitWs& w, int gx, int gy, const RegCfg& rc, const AdamCfg& ac,
                           const NextSched& ns, int extra, size_t lds, hipStream_t s);
void launch_splat_bwd_op(const gfl_fit_state* st, const FitWs& w, int gx, int gy, const NextSched& ns, int extra, size_t lds,
                         const float* d_uv, const float* d_depth, float* d_params, float* d_extr, hipStream_t s);
void launch_camera_adam(const gfl_fit_state* st, const FitWs& w, int rows, const float* p_ssim, int n_ssim, const float* p_grad,
                        int n_grad, const AdamCfg& ac_cam, const AdamCfg& ac_ab, int step_camera, hipStream_t s);
#ifdef GFL_TRACE
int read_phase_trace_bin(long long* out, int n_values);
int read_phase_trace_splat(long long* out, int n_values);
int read_fwd_trace(long long* out, int n_tiles);
int read_fwd_trace2(long long* out, int n_values);
int read_bwd_trace(long long* out, int n_tiles);
#endif

}  // namespace gfl
