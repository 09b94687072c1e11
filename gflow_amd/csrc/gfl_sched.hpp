// Tile scheduler of the fused blend kernels -- included by gfl_fused.hip.
//
// Why: a 480p frame has 1620 tiles and the chip 256 CUs x 8 resident workgroups, so every tile is
// resident from the first cycle and the hardware never gets to balance anything: workgroup b
// simply lands on CU b % 256 (measured with tools/placement_probe.hip).  The blend kernels are
// instruction-issue bound per CU (a CU's finishing time follows the number of (splat, 8x8 block)
// units it was dealt with correlation 0.92, tools/bwd_trace.py), and with tiles dealt in index
// order the busiest CU gets 1.4x the mean -- the launch lasts 1.4x longer than the work needs.
// The second limit is the longest tile: its waves run one dependent chain each, however idle
// the rest of the chip is.
//
// What: one workgroup (an otherwise idle CU during the key scatter) deals the tiles into one queue
// per CU:
//   weight    the units the blend kernel counted on that tile in the PREVIOUS iteration (the
//             scene moves slowly); the list length where there is no history yet.  The forward and the backward
//             pass count their own units and get their own queues (two scheduling workgroups side by side): a tile
//             whose pixels saturate early is cheap forward and dear backward, and with the backward's weights the
//             forward's CUs were dealt 529 to 2 661 units on a real fit (mean 1 505; the launch lasted 55 us, the
//             mean CU 30 us);
//   rounds    tiles in order of descending weight; in every round the queues whose load is
//             within twice the next tile's weight of the smallest load take one tile each, the
//             least loaded queue the heaviest (greedy LPT in batches: one ranking of the queues
//             per round instead of 1620 sequential steps; a queue that already holds a 1000-unit
//             tile sits out until the others have caught up);
//   first     the first tile of every queue is one of the NQ heaviest.  Its waves raise their
//             instruction priority, and in the backward pass it is walked as up to eight segments
//             of its list by as many workgroups of the CU side by side (the forward pass leaves
//             a per-pixel checkpoint (T, C) at every segment boundary; a segment starts from
//             T and S = sum_c g_c (out_c - C_c)).
// The 8 workgroups resident on a CU pull from that CU's queue (first pull = slot number, later
// pulls through a per-queue counter that only those 8 contend for).
// The schedule decides WHERE a tile is processed, never what is computed: results do not depend on
// it, and a dispatcher that places workgroups differently only loses the balance.
#pragma once

namespace gfl {

constexpr int SCHED_BLOCK = 512;          // threads of the scheduling workgroup (= BIN_BLOCK)
constexpr int SCHED_BINS = 2 * SCHED_BLOCK;
constexpr int SCHED_MAX_QUEUES = 512;
constexpr int SCHED_MAX_WEIGHT = 65535;   // weights and tile ids are kept as 16-bit values in LDS

struct Sched {
    int32_t* work;       // [T][4]     feedback: units the blend kernel counted per 8x8 BLOCK of each tile in the last iteration (0: none)
    int32_t* list;       // [nq][cap_q] items of queue c: tile | block plan << 16 | priority << 28 (ITEM_* below)
    int32_t* count;      // [nq]       items in each queue
    int32_t* counters;   // [2 nq]     pull counters, forward then backward
    int32_t* first_slot; // [T] or null: queue whose FIRST item the tile is, -1 for the others (backward schedule: the
                         //            forward pass leaves checkpoints for exactly those tiles)
    int nq;              // queues (= CUs)
    int cap_q;           // capacity of one queue
    int split_min;       // forward schedule: a first tile with a longer list is walked on four CUs; 0: never
};

__host__ __device__ inline int sched_queue_capacity(int T, int nq) { return 2 * ((T + nq - 1) / nq) + 8; }

// ---- block plan (round 3).  A blend workgroup has one wave on each of its CU's four SIMDs, a wave walks ONE 8x8 block of the
// tile, and the blend kernels are issue bound: a CU is done when its BUSIEST SIMD is done.  Measured on a real fit
// (tools/bwd_trace.py): a CU's finishing time follows the units of its busiest SIMD (r = 0.86-0.90; the CU's total adds
// nothing), the busiest SIMD carried 7.5 % more than a quarter of its CU on average and up to 46 % more (blocks of one
// tile differ: [50 64 62 73], [49 0 52 0] at the image's edge), and which SIMD runs wave k of a workgroup is the
// dispatcher's choice.  So the scheduler also decides, per item, WHICH SIMD walks WHICH block: it keeps the four SIMD
// loads of every queue and hands the heaviest block of the next tile to the least loaded SIMD and so on (weights = the
// units each block counted in the last iteration).  A wave reads its SIMD id (HW_REG_HW_ID) and takes the block the
// plan names for that SIMD.  Encoding: 2 bits per SIMD, plan >> (2 * simd) & 3 = block; identity = 0xE4.
constexpr int ITEM_PLAN_SHIFT = 16;
constexpr unsigned ITEM_PLAN_IDENTITY = 0xE4u;
constexpr int SCHED_PLAN_TILES = 4096;     // tiles with a higher index keep the identity plan (LDS: 4 bytes per tile)

// blocks in descending weight meet SIMDs in ascending load.  key[] carries the four SIMD loads as load * 4 + simd id
// (any order; updated); bw[] the four block weights.  Two 5-comparator sorting networks on packed keys.
__device__ __forceinline__ void plan_cswap(int& a, int& b, bool descending) {
    const int lo = min(a, b), hi = max(a, b);
    a = descending ? hi : lo;
    b = descending ? lo : hi;
}
__device__ __forceinline__ void plan_sort4(int (&k)[4], bool descending) {
    plan_cswap(k[0], k[1], descending); plan_cswap(k[2], k[3], descending);
    plan_cswap(k[0], k[2], descending); plan_cswap(k[1], k[3], descending);
    plan_cswap(k[1], k[2], descending);
}
__device__ __forceinline__ unsigned plan_blocks(const int (&bw)[4], int (&key)[4]) {
    int b[4] = {(bw[0] >> 2) * 4, (bw[1] >> 2) * 4 + 1, (bw[2] >> 2) * 4 + 2, (bw[3] >> 2) * 4 + 3};
    plan_sort4(b, true);
    plan_sort4(key, false);
    unsigned plan = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        plan |= (unsigned)(b[i] & 3) << (2 * (key[i] & 3));
        key[i] += b[i] & ~3;
    }
    return plan;
}

// exclusive scan of one int per thread over the workgroup; `total` = sum over all threads
__device__ __forceinline__ int sched_block_scan(int v, int32_t* wsum, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int sc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(sc, off);
        if (lane >= off) sc += n;
    }
    __syncthreads();
    if (lane == 63) wsum[wid] = sc;
    __syncthreads();
    int wprefix = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < SCHED_BLOCK / 64; ++w) {
        const int x = wsum[w];
        if (w < wid) wprefix += x;
        total += x;
    }
    return wprefix + sc - v;
}

// lds: scratch of T ints (used as two 16-bit arrays); wsum: LDS scratch of SCHED_BLOCK / 64 ints.
// Whole workgroup.
__device__ void schedule_tiles(const int32_t* __restrict__ tile_counts, int T, const Sched sc, int32_t* lds,
                               int32_t* wsum) {
    __shared__ int32_t bins[SCHED_BINS];
    __shared__ int32_t s_max, s_lo;
    __shared__ uint32_t frac4[SCHED_PLAN_TILES];                       // share of each block in its tile's weight, 4 x 8 bits
    unsigned short* w16 = reinterpret_cast<unsigned short*>(lds);      // weight of tile t
    unsigned short* ord16 = w16 + T;                                     // tiles by descending weight
    const int tid = threadIdx.x;
    const int NQ = sc.nq;
    // ---- 1. weights
    if (tid == 0) s_max = 1;
    for (int b = tid; b < SCHED_BINS; b += SCHED_BLOCK) bins[b] = 0;
    for (int c = tid; c < 2 * NQ; c += SCHED_BLOCK) sc.counters[c] = 0;
    if (sc.first_slot)
        for (int t = tid; t < T; t += SCHED_BLOCK) sc.first_slot[t] = -1;
    __syncthreads();
    int local = 0, lmax = 1;
    for (int t = tid; t < T; t += SCHED_BLOCK) {
        int4* w4 = reinterpret_cast<int4*>(sc.work) + t;
        const int4 b = *w4;
        int x = b.x + b.y + b.z + b.w;
        uint32_t fr = 0x40404040u;       // no history: four equal blocks
        if (x > 0) {
            const float inv = 255.f / (float)x;
            fr = (uint32_t)((float)b.x * inv) | (uint32_t)((float)b.y * inv) << 8 | (uint32_t)((float)b.z * inv) << 16 |
                 (uint32_t)((float)b.w * inv) << 24;
        } else {
            x = max(tile_counts[t], 1);
        }
        x = min(x, SCHED_MAX_WEIGHT);
        *w4 = make_int4(0, 0, 0, 0);     // the blend kernel adds this iteration's units
        w16[t] = (unsigned short)x;
        if (t < SCHED_PLAN_TILES) frac4[t] = fr;
        local += x;
        lmax = max(lmax, x);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, __shfl_xor(lmax, off));
    if ((tid & 63) == 0) atomicMax(&s_max, lmax);
    int W_total;
    sched_block_scan(local, wsum, W_total);          // (contains the barriers that publish s_max, w16)
    int shift = 0;
    while ((s_max >> shift) >= SCHED_BINS) ++shift;
    // ---- 2. tiles by descending weight (counting sort on the quantised weight; order within a
    //         bin is whatever the LDS atomics produce)
    for (int t = tid; t < T; t += SCHED_BLOCK) atomicAdd(&bins[SCHED_BINS - 1 - (w16[t] >> shift)], 1);
    __syncthreads();
    {
        const int a = bins[2 * tid], b = bins[2 * tid + 1];
        int tot;
        const int excl = sched_block_scan(a + b, wsum, tot);
        bins[2 * tid] = excl;
        bins[2 * tid + 1] = excl + a;
    }
    __syncthreads();
    for (int t = tid; t < T; t += SCHED_BLOCK) {
        const int pos = atomicAdd(&bins[SCHED_BINS - 1 - (w16[t] >> shift)], 1);
        ord16[pos] = (unsigned short)t;
    }
    __syncthreads();
    // (Round 3 measured two ways of tightening the CU balance further, both rejected.  A shared pool of the lightest tiles --
    // 10-40 % of the weight, pulled with one atomic per tile by whichever workgroup runs out of work -- made the backward
    // SLOWER, 63 -> 73 / 83 / 100 us for 10 / 20 / 40 %: the pulls of two thousand workgroups queue up on one address.  A
    // narrower eligibility window of the rounds below (0.25-1.5 x the next tile's weight instead of 2 x) left the
    // backward at 62.5-63 us while this scheduling workgroup, and with it the scatter launch, grew from 17 to 19-45 us:
    // what is left of the launch's tail is not the sums of the CUs but the chains of the heaviest tiles' segments.)
    // ---- 3. greedy LPT in batches; thread c < NQ owns queue c
    const int target = (W_total + NQ - 1) / NQ;
    int my_load = 0, my_cnt = 0;
    int simd_load[4] = {0, 1, 2, 3};                  // units planned onto each SIMD of this queue's CU, as load * 4 + simd id
    int32_t* my_list = sc.list + (size_t)min(tid, NQ - 1) * sc.cap_q;
    int next = 0;
    bool force = false;
    while (next < T) {                               // (uniform)
        const int avail = T - next;
        const int w_next = w16[ord16[next]];
        int rank = tid, m = NQ;                      // first round: all loads are zero
        if (next > 0) {
            __syncthreads();
            if (tid == 0) s_lo = 0x7fffffff;
            bins[2 * tid] = 0;
            bins[2 * tid + 1] = 0;
            __syncthreads();
            {
                // (256 LDS atomics on one word cost ~4 us: reduce in the wave first)
                int mn = (tid < NQ && my_cnt < sc.cap_q) ? my_load : 0x7fffffff;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) mn = min(mn, __shfl_xor(mn, off));
                if ((tid & 63) == 0) atomicMin(&s_lo, mn);
            }
            __syncthreads();
            const int lo = s_lo;
            const int span = force ? 0x3fffffff : 2 * w_next;
            const bool elig = tid < NQ && my_cnt < sc.cap_q && my_load - lo <= span;
            // rank of the eligible queues by load: counting sort over [lo, lo + span]
            const int bin = elig ? (int)(((long long)(my_load - lo) * (SCHED_BINS - 1)) / (span + 1)) : 0;
            if (elig) atomicAdd(&bins[bin], 1);
            __syncthreads();
            {
                const int x = bins[2 * tid], y = bins[2 * tid + 1];
                const int excl = sched_block_scan(x + y, wsum, m);
                bins[2 * tid] = excl;
                bins[2 * tid + 1] = excl + x;
            }
            __syncthreads();
            rank = elig ? atomicAdd(&bins[bin], 1) : 0x3fffffff;
        } else if (tid >= NQ) {
            rank = 0x3fffffff;
        }
        if (rank < avail && rank < m) {
            const int tile = ord16[next + rank];
            const int wt = w16[tile];
            const int prio = next > 0 ? 0 : (wt * 5 >= target * 2 ? 3 : (wt * 4 >= target ? 2 : 1));
            unsigned plan = ITEM_PLAN_IDENTITY;
            if (tile < SCHED_PLAN_TILES) {
                const uint32_t fr = frac4[tile];
                const int bw[4] = {(int)(fr & 255u) * wt, (int)((fr >> 8) & 255u) * wt, (int)((fr >> 16) & 255u) * wt,
                                   (int)(fr >> 24) * wt};
                plan = plan_blocks(bw, simd_load);
            }
            my_list[my_cnt++] = tile | (int)(plan << ITEM_PLAN_SHIFT) | (prio << 28);
            my_load += wt;
            if (next == 0 && sc.first_slot) sc.first_slot[tile] = tid;       // (the barrier after the clearing loop has passed)
        }
        if (next == 0 && tid < NQ && sc.split_min > 0) {
            // forward schedule: a long first tile costs its own CU a quarter, the other three quarters go to the queues
            // that help it (next_item: items 1..3 of queue q walk blocks of the first tile of queue q + p nq/4)
            if (tid < avail && tile_counts[ord16[tid]] > sc.split_min) my_load -= w16[ord16[tid]] - (w16[ord16[tid]] >> 2);
#pragma unroll
            for (int b = 1; b < 4; ++b) {
                const int owner = (tid + b * (NQ / 4)) % NQ;
                if (owner < avail && tile_counts[ord16[owner]] > sc.split_min) my_load += w16[ord16[owner]] >> 2;
            }
        }
        force = (m == 0);                            // every queue in reach is full: open the round to all
        next += min(m, avail);
    }
    if (tid < NQ) sc.count[tid] = my_cnt;
}

// ---- consumer side
struct TileQueue {
    const int32_t* list;
    const int32_t* count;
    int32_t* counter;    // [nq] for this launch
    int nq;
    int cap_q;
};

// Item of a queue.  part: -1 = the whole tile; 0 .. HEAVY_PARTS-1 = that segment of the queue's
// first (heaviest) tile, nearest first (backward launch only).
struct TileItem {
    int tile;     // -1: the queue is empty
    int part;
    int queue;
    unsigned plan;   // block plan (ITEM_PLAN_*): plan >> (2 * simd) & 3 = the block the wave on that SIMD walks
};

// The heaviest tile of a queue is walked in up to HEAVY_PARTS segments of its list by as many
// workgroups of the CU (backward); the forward pass leaves a checkpoint at every segment boundary.
// heavy_parts: number of segments; heavy_seg: their length, a multiple of 64 (the last is shorter).
#ifndef GFL_HEAVY_PARTS
#define GFL_HEAVY_PARTS 8
#endif
#ifndef GFL_HEAVY_SEG
#define GFL_HEAVY_SEG 160
#endif
constexpr int HEAVY_PARTS = GFL_HEAVY_PARTS;
__device__ __forceinline__ int heavy_parts(int total) {
    return total <= 128 ? 1 : min(HEAVY_PARTS, max(2, (total + GFL_HEAVY_SEG - 1) / GFL_HEAVY_SEG));
}
__device__ __forceinline__ int heavy_seg(int total, int parts) { return ((total + parts - 1) / parts + 63) & ~63; }

// next item of this workgroup's queue.  Whole workgroup.  `split`: backward launch.
__device__ __forceinline__ TileItem next_item(const TileQueue& q, int32_t* s_ticket, bool first, bool split) {
    TileItem it;
    it.queue = blockIdx.x % q.nq;
    it.part = -1;
    it.tile = -1;
    it.plan = ITEM_PLAN_IDENTITY;
    int idx = blockIdx.x / q.nq;                     // first pull: the slot number, no atomic
    if (!first) {
        __syncthreads();                             // the previous tile's LDS traffic is complete
        if (threadIdx.x == 0) *s_ticket = (int)(gridDim.x / q.nq) + atomicAdd(&q.counter[it.queue], 1);
        __syncthreads();
        idx = *s_ticket;
    }
    // backward: the first tile of the queue is HEAVY_PARTS items
    const int k = split ? max(idx - (HEAVY_PARTS - 1), 0) : idx;
    if (split && idx < HEAVY_PARTS) it.part = idx;   // (also when this queue is empty: the forward pass helps other queues)
    if (k >= q.count[it.queue]) return it;
    const int item = q.list[(size_t)it.queue * q.cap_q + k];
    it.tile = item & 0xffff;
    it.plan = ((unsigned)item >> ITEM_PLAN_SHIFT) & 0xffu;
    const int prio = (item >> 28) & 3;
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    return it;
}

}  // namespace gfl
