// Fused fit iteration, host side: the C entry points of include/gflow_hip.h for the iteration level (gfl_fit_*), the fused
// differentiable operator (gfl_render_*) and the snapshot (gfl_fit_snapshot*) -- workspace carving, argument checks, the order
// of the launches.  The kernels live in the stage files (gfl_fit_bin / _fwd / _bwd / _splat .hip) and are reached through
// the launchers declared in gfl_fit.hpp; tile sort and loss kernels through their own C entry points (gfl_bin.hip,
// gfl_ssim.hip).  Environment switches read here, once per process -- ALL the library has: GFL_EWA_MFMA, GFL_RESERVED,
// GFL_FWD_SPLIT_MIN (include/gflow_hip.h).
#include "gfl_fit.hpp"

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

// list length from which the forward blend walks a queue's first tile as four blocks (GFL_FWD_SPLIT_MIN overrides; scheduling
// only: results do not depend on it)
static int fwd_split_min() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GFL_FWD_SPLIT_MIN");
        v = e ? atoi(e) : FWD_SPLIT_MIN;
    }
    return v;
}

// one tile queue per CU (the dispatcher places workgroup b on CU b % CUs, tools/placement_probe.hip)
static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}
// (st->cu_count: the share of the device this state's stream may use -- a CU-masked stream, gfl_fit_state; a multiple of 8,
//  one queue per CU and XCD band, so that the XCD-local scheduler keeps working)
static int blend_queues(const gfl_fit_state* st) {
    int cus = (st && st->cu_count > 0) ? (st->cu_count / 8) * 8 : device_cus();
    if (cus <= 0) cus = 8;
    if (st && st->cu_count > 0) return cus < SCHED_MAX_QUEUES ? cus : SCHED_MAX_QUEUES;
    return cus < 64 ? 64 : (cus < SCHED_MAX_QUEUES ? cus : SCHED_MAX_QUEUES);
}

// workgroups of a blend launch: up to max_per_cu per queue (all resident), fewer for small tile grids.  (Fewer than fit, so
// that part of a queue is pulled as workgroups finish, was measured slower for both launches: docs/history.md section 7.)
static int blend_grid(const gfl_fit_state* st, int T, int max_per_cu = BLEND_WG_PER_CU) {
    const int nq = blend_queues(st);
    int per = (T + nq - 1) / nq + 1;
    if (per > max_per_cu) per = max_per_cu;
    return nq * per;
}


size_t gfl_fit_workspace_bytes(int cap, int K_cap, int W, int H) {
    if (cap < 0 || K_cap < 0 || W <= 0 || H <= 0) return 0;
    const size_t T = (size_t)((W + GFL_TILE - 1) / GFL_TILE) * ((H + GFL_TILE - 1) / GFL_TILE);
    return up256((size_t)fit_nblk(cap > 0 ? cap : 1) * T * sizeof(int32_t))      // hist / bases
           + up256((size_t)K_cap * sizeof(unsigned long long))                      // keys
           + up256((size_t)reduce_rows(cap > 0 ? cap : 1) * 12 * sizeof(float))    // extr partials
           + up256(T * sizeof(int32_t))                                             // tile totals
           + up256(((size_t)(cap > 0 ? cap : 1) * SLOT_MAX + (size_t)K_cap) * PG * sizeof(float))   // pair rows: SLOT_MAX per splat + the wide splats' runs
           + up256((size_t)(cap > 0 ? cap : 1) * sizeof(int32_t))                   // a wide splat's run
           + 256                                                                    // counters
           + up256(4 * T * sizeof(int32_t))                                         // scheduler: work feedback per 8x8 block
           + up256((size_t)SCHED_MAX_QUEUES * sched_queue_capacity((int)T, 64) * sizeof(int32_t))   // queue items
           + 2 * up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t))                      // queue lengths, pull counters
           + up256((size_t)SCHED_MAX_QUEUES * (HEAVY_PARTS - 1) * 5 * 256 * sizeof(float))                  // heavy-tile checkpoints
           + up256(4 * T * sizeof(int32_t)) + up256(T * sizeof(int32_t))            // forward schedule: work feedback; first_slot
           + up256((size_t)SCHED_MAX_QUEUES * sched_queue_capacity((int)T, 64) * sizeof(int32_t))   // ... queue items
           + up256(2 * SCHED_MAX_QUEUES * sizeof(int32_t))                          // ... queue lengths
           + up256(gfl_loss_workspace_bytes(W, H)) + 256
           + up256((size_t)6 * W * H * sizeof(float))                                  // SSIM statistics of the target
           + up256((size_t)2 * fit_nblk(cap > 0 ? cap : 1) * sizeof(int32_t))          // rows of the scale term per 256 splats
           + up256((T * 4 + SORT_ORDER_TRAILER) * sizeof(int32_t))                     // the tile sort's order (+ its split list)
           + up256((T * 4 + SORT_ORDER_TRAILER) * sizeof(int32_t))                     // reserved tile regions: the next sort order,
           + up256(T * sizeof(int4)) + up256(T * sizeof(int32_t));                      //   {start, capacity, position} per tile, fill counters
}


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
    w.wide_base = (long long)(st->cap > 0 ? st->cap : 1) * SLOT_MAX;
    p += up256(((size_t)(st->cap > 0 ? st->cap : 1) * SLOT_MAX + (size_t)st->K_cap) * PG * sizeof(float));
    w.wide_off = (int32_t*)p;
    p += up256((size_t)(st->cap > 0 ? st->cap : 1) * sizeof(int32_t));
    w.pool_counter = (int32_t*)p;
    w.stamp = w.pool_counter + 56;
    w.snap_mm = (unsigned*)(w.pool_counter + 58);
    w.sched_valid = w.pool_counter + 16;
    w.regions_valid = w.pool_counter + 32;
    w.extent = w.pool_counter + 40;
    w.extent_next = w.pool_counter + 48;
    p += 256;
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
    w.sched.nq = blend_queues(st);
    // (the list is sized for 512 queues: with fewer queues each may hold more -- a band of the XCD-local schedule
    //  can have many more tiles than T / 8)
    w.sched.cap_q = (int)(((size_t)SCHED_MAX_QUEUES * sched_queue_capacity((int)T, 64)) / w.sched.nq);
    w.sched.split_min = 0;
    w.sched.xcd = 1;         // XCD-local bands + LPT in rounds wherever the grid allows it (sched_xcd_usable, next_sched_ok)
    w.sched_fwd = w.sched;
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
    w.sort_order = (int4*)((char*)w.scale_cnt + up256((size_t)2 * fit_nblk(st->cap > 0 ? st->cap : 1) * sizeof(int32_t)));
    w.sort_order_next = (int4*)((char*)w.sort_order + up256((T * 4 + SORT_ORDER_TRAILER) * sizeof(int32_t)));
    w.region = (int4*)((char*)w.sort_order_next + up256((T * 4 + SORT_ORDER_TRAILER) * sizeof(int32_t)));
    w.fill = (int32_t*)((char*)w.region + up256(T * sizeof(int4)));
    return w;
}

// the next iteration's tile queues are built by two extra workgroups of the per-splat launch when the XCD-local scheduler
// can run on REDUCE_BLOCK threads (otherwise the scatter launch keeps building them in line, every iteration)
static bool next_sched_ok(const FitWs& w, int T) {
    return w.sched.xcd && T <= SCHED_PLAN_TILES && w.sched.nq % 8 == 0 && w.sched.nq / 8 <= 64 && w.sched.nq <= REDUCE_BLOCK;
}
static int next_sched_blocks(const FitWs& w, int T) { return next_sched_ok(w, T) ? 2 : 0; }
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
// Reserved tile regions: grids of up to 4096 tiles (the region workgroup holds a tile's count in registers: 16 per lane on
// REDUCE_BLOCK lanes; the binning kernel 8 per lane on BIN_BLOCK), the scheduling workgroups in the per-splat launch (the
// region workgroup is the third of them).
static bool fit_reserved_ok(const FitWs& w, int T) {
    return reserved_enabled() && next_sched_ok(w, T) && T <= 4096 && (2 * (size_t)T + 64) * sizeof(int32_t) <= 57 * 1024;
}
static NextSched next_sched_reserving(const gfl_fit_state* st, const FitWs& w, int rows, int T) {
    NextSched ns = next_sched(w, rows, T);
    if (!fit_reserved_ok(w, T)) return ns;
    ns.reserve = 1;
    ns.order_next = w.sort_order_next;
    ns.ro.region = w.region; ns.ro.fill = w.fill; ns.ro.extent_next = w.extent_next; ns.ro.total = st->tile_offsets + T;
    ns.ro.K_cap = st->K_cap;
    ns.regions_valid = w.regions_valid;
    ns.pool_counter = w.pool_counter;
    ns.pull_counters = w.sched.counters; ns.n_pull = 2 * w.sched.nq;
    return ns;
}

static int fit_check(const gfl_fit_state* st, const gfl_fit_hyper* hp) {
    if (!st || !hp) return GFL_ERR_INVALID;
    if (st->N < 0 || st->N > st->cap || st->W <= 0 || st->H <= 0 || st->K_cap < 0) return GFL_ERR_INVALID;
    if (!st->params || !st->rec || !st->pose || !st->intr || !st->extr || !st->render || !st->final_T ||
        !st->n_contrib || !st->tile_offsets || !st->ids || !st->tile_range || !st->overflow || !st->workspace)
        return GFL_ERR_INVALID;
    if (st->workspace_bytes < gfl_fit_workspace_bytes(st->cap, st->K_cap, st->W, st->H)) return GFL_ERR_WORKSPACE;
    return GFL_OK;
}

// kernels of gfl_bin.hip / gfl_loss.hip reused through their C entry points
// (gfl_loss_fwd_bwd, gfl_tile_sort_only: declared in gflow_hip.h)

static PreArgs pre_args(const gfl_fit_state* st, const gfl_fit_hyper* hp, const FitWs& w, int gx, int gy, int op_mode) {
    PreArgs a;
    a.intr = st->intr; a.pose = st->pose;
    a.N = st->N; a.W = st->W; a.H = st->H;
    a.nearest = hp->nearest; a.extent = hp->extent;
    a.gx = gx; a.gy = gy;
    a.rec = st->rec; a.wide_off = w.wide_off; a.hist_g = w.hist; a.extr_out = st->extr; a.overflow = st->overflow;
    a.pool_counter = w.pool_counter; a.pool_cap = st->K_cap;
    a.op_mode = op_mode;
    a.scale_rows_mode = (!op_mode && hp->lambda_scale != 0.f) ? (hp->freeze_all_splats ? 2 : 1) : 0;
    a.scale_cnt = w.scale_cnt;
    return a;
}

// reserved: the tile regions the previous iteration's last launch reserved are used (one launch instead of three).
// snap_lut / snap_u8 (gfl_fit_iteration_snapshot): this forward also leaves the three snapshot images, as uint8.
static int fit_forward_impl(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream, int op_mode, int reserved = 0,
                            const float* snap_lut = nullptr, uint8_t* snap_u8 = nullptr) {
    int rc = fit_check(st, hp);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int gx = (st->W + GFL_TILE - 1) / GFL_TILE, gy = (st->H + GFL_TILE - 1) / GFL_TILE, T = gx * gy;
    const FitWs w = carve(st);
    const int nblk = fit_nblk(st->N > 0 ? st->N : 1);
    // tile grid too large for the LDS histogram: 64 KB less the scheduling workgroups' ~6 KB of static state (their block
    // plans' 16 KB exist only for grids of up to 4096 tiles).  14 592 tiles: 2560 x 1440 has 14 400.
    if ((size_t)T * sizeof(int32_t) > 57 * 1024) return GFL_ERR_INVALID;
    const PreArgs pa = pre_args(st, hp, w, gx, gy, op_mode);
    if (reserved) {
        if (op_mode || !fit_reserved_ok(w, T)) return GFL_ERR_INVALID;
        {
            StageScope p(ST_PREPROCESS, s);
            BinArgs b;
            b.region = w.region; b.fill = w.fill; b.keys = w.keys; b.K_cap = st->K_cap;
            b.regions_valid = w.regions_valid; b.extent_next = w.extent_next; b.extent = w.extent;
            b.pull_counters = w.sched.counters; b.n_pull = 2 * w.sched.nq;
            launch_preprocess_bin(st->params, pa, st->row_flags, b, nblk, ewa_on_mfma(), s);
        }
        {
            StageScope p(ST_TILE_SORT, s);
            rc = gfl_tile_sort_reserved((const int32_t*)w.sort_order_next, w.fill, w.tile_counts, st->overflow + 2, st->W, st->H,
                                        st->K_cap, w.keys, st->ids, st->tile_range, stream);
        }
        if (rc) return rc;
    } else {
        {
            StageScope p(ST_PREPROCESS, s);
            launch_preprocess_fwd(st->params, pa, st->row_flags, nblk, ewa_on_mfma(), s);
        }
        {
            StageScope p(ST_COLSCAN, s);
            launch_colscan(w, nblk, T, st->overflow, s);
        }
        // (the order of the tile sort is built by one workgroup with up to eight tiles per lane in registers: beyond 4096 tiles
        //  -- 1080p has 8160 -- the sort takes the tiles in their own order)
        const bool ordered = T <= 8 * BIN_BLOCK;
        {
            StageScope p(ST_SCATTER, s);
            launch_scatter(st, w, nblk, gx, gy, ordered, s);
        }
        {
            StageScope p(ST_TILE_SORT, s);
            if (ordered)
                rc = gfl_tile_sort_ordered((const int32_t*)w.sort_order, st->W, st->H, st->K_cap, w.keys, st->ids, st->tile_range,
                                           stream);
            else
                rc = gfl_tile_sort_only(st->tile_offsets, gx * gy, st->K_cap, w.keys, st->ids, st->tile_range, stream);
        }
        if (rc) return rc;
    }
    {
        StageScope p(ST_BLEND_FWD, s);
        const TileQueue q = {w.sched_fwd.list, w.sched_fwd.count, w.sched.counters, w.sched.nq, w.sched.cap_q};
        if (snap_u8) {
            // the range of the splats' depths (the turbo map's normalisation), then the forward that composes rgb and
            // depth_map_color in one walk, then "center" over the same lists
            rc = check(hipMemsetAsync(w.snap_mm, 0, 2 * sizeof(unsigned), s));
            if (rc) return rc;
            if (st->N > 0) launch_rec_depth_range(st->rec, st->N, w.snap_mm, s);
            launch_blend_fwd(st, hp->bg, gx, blend_grid(st, T, FWD_WG_PER_CU), st->render, st->final_T, st->n_contrib, q, w, 2,
                             w.snap_mm, snap_lut, fwd_split_min(), s, snap_u8);
            launch_center_blend(st, hp->bg, gx, T, nullptr, snap_u8 + (size_t)2 * st->W * st->H * 3, s);
        }
        // (camera-only stage, black background: the footprint's workgroups ride behind the blend's in the same launch, mode 3 --
        //  `fused_blend_fwd<3>` 53.0 us where the plain forward takes 50.5 and the footprint launch behind it took 9.8: 8-frame
        //  clip fits 0.824-0.831 s against 0.826-0.843 s, three alternations on one box; a snapshot iteration keeps the launch)
        const bool foot = st->foot_flags && !(hp->bg > 0.f) && st->N > 0;
        if (foot && !st->keep) return GFL_ERR_INVALID;
        const bool foot_inside = foot && !snap_u8;
        if (!snap_u8)
            launch_blend_fwd(st, hp->bg, gx, blend_grid(st, T, FWD_WG_PER_CU), st->render, st->final_T, st->n_contrib, q, w,
                             foot_inside ? 3 : 0, nullptr, nullptr, fwd_split_min(), s);
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
            if (foot && !foot_inside) launch_footprint(st, gx, T, s);
        }
    }
    return check_launch();
}

int gfl_fit_forward(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream) {
    return fit_forward_impl(st, hp, stream, 0);
}

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
    if (st->N > 0) launch_rec_depth_range(st->rec, st->N, mm, s);
    // center first: a kernel of its own (gfl_fit_fwd.hip), short-lived workgroups.  On a side stream (trainer.py: _snapshot_async)
    // these launches run beside the NEXT iteration's binning and tile sort, and the persistent workgroups of the blend kernel
    // below hold every CU's registers for their ~50 us: the tile sort beside them took 44 us instead of 14
    // (tools/snapshot_gaps.py); behind this kernel they start when that iteration is past its sort.
    launch_center_blend(st, hp->bg, gx, T, img_c, nullptr, s);
    {
        // depth_map_color: the blend kernel over the forward's queues (the iteration's own launch used up the engine's pull counters)
        const TileQueue q = {w.sched_fwd.list, w.sched_fwd.count, pull, w.sched.nq, w.sched.cap_q};
        // (fewer workgroups per CU for this launch, so that it disturbs the fit's own kernels less, was measured in
        //  round 4: one per CU 0.871-0.886 s per 8-frame clip fit against 0.858-0.865 with five, three the same as five)
        launch_blend_fwd(st, hp->bg, gx, blend_grid(st, T, FWD_WG_PER_CU), img_dc, fT, nc, q, w, 1, mm, lut, fwd_split_min(), s);
    }
    launch_snapshot_u8(st->render, img_dc, img_c, P, out_u8, s);
    return check_launch();
}

// Everything gfl_fit_snapshot reads of a forward -- records, sorted ids, tile ranges, the rgb planes of the render, the
// forward's tile queues -- copied from one engine to another in ONE launch.  The number of ids is a device value (the
// extent of the last forward's lists).
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
    launch_snapshot_stage(c, (hipStream_t)stream);
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
        const TileQueue q = {w.sched.list, w.sched.count, w.sched.counters + w.sched.nq, w.sched.nq, w.sched.cap_q};
        launch_blend_bwd(st, hp->bg, gx, gy, blend_grid(st, T), 10, d_render, q, w, LossTail{}, s);
    }
    {
        StageScope p(ST_PRE_BWD_ADAM, s);
        const NextSched ns = next_sched(w, reduce_rows(st->N > 0 ? st->N : 1), T);
        launch_splat_bwd_op(st, w, gx, gy, ns, next_sched_blocks(w, T), next_sched_lds(w, T), d_uv, d_depth, d_params, d_extr, s);
    }
    return check_launch();
}

int gfl_fit_backward_step(const gfl_fit_state* st, const gfl_fit_hyper* hp, gfl_stream_t stream) {
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
    // for the gradient (d_extr) although nothing is stepped with it.
    const bool frozen = hp->step_camera != 2 && (hp->step_camera == 0 || hp->lr_camera == 0.f);
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
        const TileQueue q = {w.sched.list, w.sched.count, w.sched.counters + w.sched.nq, w.sched.nq, w.sched.cap_q};
        // sums that nobody reads are not formed: 6 in the camera-only stage, 7 while the colours are frozen (see the kernel)
        const int sums = hp->freeze_all_splats ? 6 : (hp->freeze_rgb ? 7 : 10);
        launch_blend_bwd(st, hp->bg, gx, gy, blend_grid(st, T), sums, st->d_render, q, w, lt, s);
    }
    const int rows = reduce_rows(st->N > 0 ? st->N : 1);
    RegCfg rcfg;
    rcfg.lambda_scale = hp->lambda_scale;
    rcfg.scale_blocks = 2 * fit_nblk(st->N > 0 ? st->N : 1);      // (one partial per 256 splats: preprocess_block)
    rcfg.lambda_var = st->N > 0 ? hp->lambda_var / (float)st->N : 0.f;
    rcfg.lambda_flow = hp->lambda_flow;
    rcfg.lambda_still = hp->lambda_still;
    rcfg.freeze_rgb = hp->freeze_rgb;
    rcfg.freeze_all = hp->freeze_all_splats;
    rcfg.no_pose_grad = frozen ? 1 : 0;
    {
        StageScope p(ST_PRE_BWD_ADAM, s);
        const NextSched ns = next_sched_reserving(st, w, rows, T);
        launch_splat_bwd_adam(st, w, gx, gy, rcfg, ac, ns, next_sched_blocks(w, T) + ns.reserve, next_sched_lds(w, T), s);
    }
    if (!frozen) {
        StageScope p(ST_CAMERA, s);
        launch_camera_adam(st, w, rows, p_ssim, n_ssim, p_grad, n_grad, ac_cam, ac, hp->step_camera != 0, s);
    }
    return check_launch();
}

int gfl_fit_iterations(const gfl_fit_state* st, const gfl_fit_hyper* hp, int count, int flags, gfl_stream_t stream) {
    if (count < 1 || (flags & ~GFL_ITER_RESERVED)) return GFL_ERR_INVALID;
    int rc = fit_check(st, hp);
    if (rc) return rc;
    const int T = ((st->W + GFL_TILE - 1) / GFL_TILE) * ((st->H + GFL_TILE - 1) / GFL_TILE);
    const FitWs w = carve(st);
    const bool res_ok = fit_reserved_ok(w, T) && st->N > 0;
    if ((flags & GFL_ITER_RESERVED) && !res_ok) return GFL_ERR_INVALID;
    for (int j = 0; j < count; ++j) {
        // (every iteration of a call but the first follows a full iteration: its tile regions are reserved)
        const int reserved = res_ok && (j > 0 || (flags & GFL_ITER_RESERVED)) ? 1 : 0;
        rc = fit_forward_impl(st, hp, stream, 0, reserved);
        if (rc) return rc;
        rc = gfl_fit_backward_step(st, hp, stream);
        if (rc) return rc;
    }
    return GFL_OK;
}

int gfl_fit_iteration_snapshot(const gfl_fit_state* st, const gfl_fit_hyper* hp, const float* lut, uint8_t* out_u8,
                               gfl_stream_t stream) {
    if (!lut || !out_u8) return GFL_ERR_INVALID;
    int rc = fit_forward_impl(st, hp, stream, 0, 0, lut, out_u8);
    if (rc) return rc;
    return gfl_fit_backward_step(st, hp, stream);
}

int gfl_fit_reserved_supported(const gfl_fit_state* st, const gfl_fit_hyper* hp) {
    if (fit_check(st, hp) || st->N <= 0) return 0;
    const int T = ((st->W + GFL_TILE - 1) / GFL_TILE) * ((st->H + GFL_TILE - 1) / GFL_TILE);
    return fit_reserved_ok(carve(st), T) ? 1 : 0;
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
int gfl_debug_read_fwd_trace(long long* out, int n_tiles) { return read_fwd_trace(out, n_tiles); }
int gfl_debug_read_fwd_trace2(long long* out, int n_values) { return read_fwd_trace2(out, n_values); }
int gfl_debug_read_phase_trace(long long* out, int n_values) {
    const int rc = read_phase_trace_bin(out, n_values);
    return rc ? rc : read_phase_trace_splat(out, n_values);
}
int gfl_debug_read_bwd_trace(long long* out, int n_tiles) { return read_bwd_trace(out, n_tiles); }
#endif

}  // extern "C"
