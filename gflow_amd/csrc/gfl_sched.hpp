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
    int xcd;             // 1: XCD-local bands + snake deal (schedule_tiles_xcd) instead of the batched LPT
};

__host__ __device__ inline int sched_queue_capacity(int T, int nq) { return 2 * ((T + nq - 1) / nq) + 8; }

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

// exclusive scan of one int per thread over the workgroup of BLOCK threads; `total` = sum over all threads
template <int BLOCK>
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
    for (int w = 0; w < BLOCK / 64; ++w) {
        const int x = wsum[w];
        if (w < wid) wprefix += x;
        total += x;
    }
    return wprefix + sc - v;
}

// exclusive scan over the SCHED_BINS bins in place (BLOCK threads, SCHED_BINS / BLOCK consecutive bins each); returns the total
template <int BLOCK>
__device__ __forceinline__ int sched_scan_bins(int32_t* bins, int32_t* wsum) {
    constexpr int PER = SCHED_BINS / BLOCK;
    const int tid = threadIdx.x;
    int v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = bins[PER * tid + k]; sum += v[k]; }
    int tot;
    int run = sched_block_scan<BLOCK>(sum, wsum, tot);
#pragma unroll
    for (int k = 0; k < PER; ++k) { bins[PER * tid + k] = run; run += v[k]; }
    return tot;
}

// static LDS of the scheduling workgroup, declared ONCE in the kernel that calls either scheduler
struct SchedLds {
    int32_t bins[SCHED_BINS];
    int32_t s_max, s_lo;
    int32_t g_base[9];
    int32_t rank[SCHED_BLOCK];
};

// lds: scratch of T ints (used as two 16-bit arrays); wsum: LDS scratch of SCHED_BLOCK / 64 ints.  SCHED_BLOCK threads.
// Whole workgroup.
// frac4: SCHED_PLAN_TILES words of LDS for the block plans (share of each block in its tile's weight, 4 x 8 bits), or null:
// every item then keeps the identity plan.  (The caller carves it out of its dynamic LDS only for tile grids of up to
// SCHED_PLAN_TILES tiles: a 1440p frame needs 56 KB of histogram there, and 16 KB more would not fit.)
__device__ void schedule_tiles(const int32_t* __restrict__ tile_counts, int T, const Sched sc, int32_t* lds,
                               int32_t* wsum, SchedLds& sl, uint32_t* frac4) {
    int32_t (&bins)[SCHED_BINS] = sl.bins;
    int32_t& s_max = sl.s_max;
    int32_t& s_lo = sl.s_lo;
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
        if (frac4 && t < SCHED_PLAN_TILES) frac4[t] = fr;
        local += x;
        lmax = max(lmax, x);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, __shfl_xor(lmax, off));
    if ((tid & 63) == 0) atomicMax(&s_max, lmax);
    int W_total;
    sched_block_scan<SCHED_BLOCK>(local, wsum, W_total);          // (contains the barriers that publish s_max, w16)
    int shift = 0;
    while ((s_max >> shift) >= SCHED_BINS) ++shift;
    // ---- 2. tiles by descending weight (counting sort on the quantised weight; order within a
    //         bin is whatever the LDS atomics produce)
    for (int t = tid; t < T; t += SCHED_BLOCK) atomicAdd(&bins[SCHED_BINS - 1 - (w16[t] >> shift)], 1);
    __syncthreads();
    {
        const int a = bins[2 * tid], b = bins[2 * tid + 1];
        int tot;
        const int excl = sched_block_scan<SCHED_BLOCK>(a + b, wsum, tot);
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
                const int excl = sched_block_scan<SCHED_BLOCK>(x + y, wsum, m);
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
            if (frac4 && tile < SCHED_PLAN_TILES) {
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

// value of lane (lane ^ D), D in {1, 2, 4, 8, 16}, on the VALU only (DPP / v_permlane16_swap; see gfl_tile_sort.hpp)
template <int D>
__device__ __forceinline__ unsigned sched_xor_lane(unsigned v, int lane) {
    const int x = (int)v;
    if constexpr (D == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
    else if constexpr (D == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
    else if constexpr (D == 4) {
        int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);
        t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);
        return (unsigned)t;
    } else if constexpr (D == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);
    else {
        float a = __builtin_bit_cast(float, v), b = a;
        permlane16_swap(a, b);
        return __builtin_bit_cast(unsigned, (lane & 16) ? a : b);
    }
}
template <int D>
__device__ __forceinline__ void sched_cx(unsigned& key, int k, int lane) {
    const unsigned other = sched_xor_lane<D>(key, lane);
    const bool take_min = ((lane & D) == 0) == ((lane & 31 & k) == 0);
    key = ((key < other) == take_min) ? key : other;
}
// ascending bitonic sort of one unique 32-bit key per lane inside each 32-lane half of the wave (15 exchange steps)
__device__ __forceinline__ unsigned sched_sort32(unsigned key, int lane) {
    sched_cx<1>(key, 2, lane);
    sched_cx<2>(key, 4, lane); sched_cx<1>(key, 4, lane);
    sched_cx<4>(key, 8, lane); sched_cx<2>(key, 8, lane); sched_cx<1>(key, 8, lane);
    sched_cx<8>(key, 16, lane); sched_cx<4>(key, 16, lane); sched_cx<2>(key, 16, lane); sched_cx<1>(key, 16, lane);
    sched_cx<16>(key, 32, lane); sched_cx<8>(key, 32, lane); sched_cx<4>(key, 32, lane); sched_cx<2>(key, 32, lane);
    sched_cx<1>(key, 32, lane);
    return key;
}

// ---- XCD-local schedule (round 3; GFL_SCHED_XCD=1).
// The dispatcher places workgroup b on XCD b % 8 and every XCD has its own L2.  With the queues above a tile lands on
// whichever CU balances the loads, so every XCD ends up walking tiles from all over the image and pulls (nearly) ALL
// splat records through its own L2: the blend kernels fetched the 2.9 MB of records eight times (19.7 MB read by the
// forward against 10.7 algorithmic, rocprofv3 FETCH_SIZE).  Here the image is first cut into eight horizontal BANDS of
// equal weight (a prefix sum of the tile weights in row-major order), band x belongs to the queues q with q % 8 == x, and
// inside a band the tiles are dealt in order of descending weight in a snake over the band's queues (stripe k forwards,
// stripe k + 1 backwards): no iterative rounds at all -- the scheduling workgroup is no longer the long pole of the
// scatter launch.  The first tile of every queue is still one of the band's heaviest (segments / four-CU walk), the
// block plans are made per queue in order.  Needs nq % 8 == 0 and nq / 8 <= 64; otherwise the caller uses schedule_tiles.
template <int BLOCK>
__device__ void schedule_tiles_xcd(const int32_t* __restrict__ tile_counts, int T, const Sched sc, int32_t* lds, int32_t* wsum,
                                   SchedLds& sl, uint32_t* frac4) {
    int32_t (&bins)[SCHED_BINS] = sl.bins;
    int32_t& s_max = sl.s_max;
    int32_t (&g_base)[9] = sl.g_base;
    unsigned short* w16 = reinterpret_cast<unsigned short*>(lds);      // weight of tile t
    unsigned short* ord16 = w16 + T;                                     // tiles by (band, descending weight)
    const int tid = threadIdx.x;
    const int NQ = sc.nq, NQG = NQ / 8;
    if (tid == 0) s_max = 1;
    for (int b = tid; b < SCHED_BINS; b += BLOCK) bins[b] = 0;
    for (int c = tid; c < 2 * NQ; c += BLOCK) sc.counters[c] = 0;
    if (sc.first_slot)
        for (int t = tid; t < T; t += BLOCK) sc.first_slot[t] = -1;
    __syncthreads();
    // ---- weights; thread tid owns the CONTIGUOUS tiles [tid * per, (tid + 1) * per) (row-major order = bands)
    const int per = (T + BLOCK - 1) / BLOCK;
    const int t_lo = tid * per, t_hi = min(T, t_lo + per);
    int local = 0, lmax = 1;
    for (int t = t_lo; t < t_hi; ++t) {
        int4* w4 = reinterpret_cast<int4*>(sc.work) + t;
        const int4 b = *w4;
        int x = b.x + b.y + b.z + b.w;
        uint32_t fr = 0x40404040u;
        if (x > 0) {
            const float inv = 255.f / (float)x;
            fr = (uint32_t)((float)b.x * inv) | (uint32_t)((float)b.y * inv) << 8 | (uint32_t)((float)b.z * inv) << 16 |
                 (uint32_t)((float)b.w * inv) << 24;
        } else {
            x = max(tile_counts[t], 1);
        }
        x = min(x, SCHED_MAX_WEIGHT);
        *w4 = make_int4(0, 0, 0, 0);
        w16[t] = (unsigned short)x;
        if (frac4 && t < SCHED_PLAN_TILES) frac4[t] = fr;
        local += x;
        lmax = max(lmax, x);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, __shfl_xor(lmax, off));
    if ((tid & 63) == 0) atomicMax(&s_max, lmax);
    int W_total;
    int run = sched_block_scan<BLOCK>(local, wsum, W_total);      // weight in front of this thread's tiles (+ the barriers)
    int shift = 0;
    while ((s_max >> shift) >= 128) ++shift;
    // ---- counting sort by (band, descending weight); the band of a tile = where its prefix weight falls
    const long long Wt = max(W_total, 1);
    for (int t = t_lo; t < t_hi; ++t) {
        const int band = min(7, (int)(((long long)run * 8) / Wt));
        run += w16[t];
        ord16[t] = (unsigned short)band;                   // (parked here until the scatter below)
        atomicAdd(&bins[band * 128 + 127 - (w16[t] >> shift)], 1);
    }
    __syncthreads();
    sched_scan_bins<BLOCK>(bins, wsum);
    __syncthreads();
    if (tid < 8) g_base[tid] = bins[tid * 128];
    if (tid == 8) g_base[8] = T;
    // (T <= 4096 = SCHED_PLAN_TILES here, i.e. at most MAXPER tiles per thread: every parked band is read before any
    //  sorted position is written)
    constexpr int MAXPER = SCHED_PLAN_TILES / BLOCK;
    int my_pos[MAXPER];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) {
        const int t = t_lo + k;
        if (k < per && t < t_hi) my_pos[k] = atomicAdd(&bins[(int)ord16[t] * 128 + 127 - (w16[t] >> shift)], 1);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) {
        const int t = t_lo + k;
        if (k < per && t < t_hi) ord16[my_pos[k]] = (unsigned short)t;
    }
    __syncthreads();
    // ---- deal, band by band: LPT in rounds.  Thread x * NQG + j owns queue q = j * 8 + x, so the NQG queues of a band sit
    // in NQG consecutive lanes (NQG <= 64 and a power of two dividing 64, or the band's ranks are taken with the snake
    // below).  Round k hands the band's tiles k * NQG ... (k + 1) * NQG - 1 (descending weight) to the queues in order of
    // ASCENDING load: the least loaded queue takes the heaviest tile of the round (ranks by wave shuffles, no LDS, no
    // barrier -- the batched LPT above needs six barriers per round).
    const int target = (W_total + NQ - 1) / NQ;
    const bool ranked = NQG == 32;                       // (ranks by a 32-lane register sort; other queue counts: the snake)
    if (tid < NQ) {
        const int x = tid / NQG, j = tid % NQG;
        const int q = j * 8 + x;
        const int b0 = g_base[x], cnt = g_base[x + 1] - b0;
        int32_t* my_list = sc.list + (size_t)q * sc.cap_q;
        int key[4] = {0, 1, 2, 3};
        int load = 0, n_items = 0;
        const int lane = tid & 63;
        // Rounds.  A queue whose load has reached the mean final load (minus half the next tile) SITS OUT: every queue
        // taking a tile in every round gave the queue that holds a 1 000-unit pile five more tiles, the lightest of their
        // rounds -- 1 793 units against a mean of 1 471 on a real fit; with this rule the busiest queue has 1 568
        // (offline replay of the same weights).  m = queues taking a tile this round; the round consumes m tiles.
        int next = 0;                                                  // (uniform over the band's lanes)
        for (int k = 0; next < cnt; ++k) {                             // (uniform: a queue's capacity only makes it sit out)
            const int w_next = w16[ord16[b0 + next]];
            const bool room = n_items < sc.cap_q;                      // (a band's queues hold >= 2 T items together)
            bool elig = room && (k == 0 || load + (w_next >> 1) < target || !ranked);
            unsigned bm = (unsigned)(__ballot(elig) >> (lane & 32));   // this band's 32 lanes (NQG == 32) ...
            if (!ranked) bm = 0xffffffffu;
            if (bm == 0u) {                                            // everyone has its share: open the round to all
                elig = room;
                bm = (unsigned)(__ballot(elig) >> (lane & 32));
                if (bm == 0u) break;
            }
            const int m = ranked ? __popc(bm) : NQG;
            int rank = (k & 1) ? NQG - 1 - j : j;                      // snake (also the first round: all loads are zero)
            if (ranked && k > 0) {
                // rank of this queue's load among its band's 32: a register bitonic sort of (load, lane) over the
                // half-wave -- 15 DPP / permlane steps -- and the inverse permutation through LDS.  (32 ds_bpermute
                // shuffles per round made this workgroup 13 us slower, 64 v_readlane broadcasts 20 us: its four waves
                // are alone on their SIMDs and issue every ~8 cycles.)  Queues that sit out sort to the end.
                const unsigned mine = elig ? (((unsigned)max(load, 0) << 6) | (unsigned)j) : (0xffffffc0u | (unsigned)j);
                const unsigned sorted = sched_sort32(mine, lane);
                sl.rank[(tid - j) + (int)(sorted & 63u)] = j;          // lane j of the half now holds the rank-j queue's key
                rank = sl.rank[tid];
            }
            const int p = next + rank;
            if (rank < m && p < cnt && room) {
                const int tile = ord16[b0 + p];
                const int wt = w16[tile];
                const int prio = k > 0 ? 0 : (wt * 5 >= target * 2 ? 3 : (wt * 4 >= target ? 2 : 1));
                unsigned plan = ITEM_PLAN_IDENTITY;
                if (frac4 && tile < SCHED_PLAN_TILES) {
                    const uint32_t fr = frac4[tile];
                    const int bw[4] = {(int)(fr & 255u) * wt, (int)((fr >> 8) & 255u) * wt, (int)((fr >> 16) & 255u) * wt,
                                       (int)(fr >> 24) * wt};
                    plan = plan_blocks(bw, key);
                }
                my_list[n_items] = tile | (int)(plan << ITEM_PLAN_SHIFT) | (prio << 28);
                if (n_items == 0 && sc.first_slot) sc.first_slot[tile] = q;
                ++n_items;
                load += wt;
            }
            if (k == 0 && sc.split_min > 0) {
                // forward schedule: a long first tile costs its own CU a quarter, the other three quarters go to the queues
                // that help it (next_item: items 1..3 of queue q walk blocks of the first tile of queue q + p nq/4)
                if (cnt > j && tile_counts[ord16[b0 + j]] > sc.split_min) load -= w16[ord16[b0 + j]] - (w16[ord16[b0 + j]] >> 2);
#pragma unroll
                for (int b = 1; b < 4; ++b) {
                    const int owner = (q + b * (NQ / 4)) % NQ;
                    const int ox = owner & 7, oj = owner >> 3;
                    if (g_base[ox + 1] - g_base[ox] > oj) {
                        const int ot = ord16[g_base[ox] + oj];
                        if (tile_counts[ot] > sc.split_min) load += w16[ot] >> 2;
                    }
                }
            }
            next += min(m, cnt - next);
        }
        sc.count[q] = n_items;
    }
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

// item number idx of this workgroup's queue when the queue's first tile counts as `parts` items
__device__ __forceinline__ TileItem item_at(const TileQueue& q, int idx, int parts) {
    TileItem it;
    it.queue = blockIdx.x % q.nq;
    it.part = -1;
    it.tile = -1;
    it.plan = ITEM_PLAN_IDENTITY;
    const int k = max(idx - (parts - 1), 0);
    if (idx < parts) it.part = idx;                  // (also when this queue is empty: the forward pass helps other queues)
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
// next item of this workgroup's queue.  Whole workgroup.  `parts`: the queue's first tile is that many items (the forward
// launch: four blocks -- with the backward's HEAVY_PARTS there too, every queue's tickets 4..7 were drawn for nothing, a round trip
// each at the start of the launch; the backward launch: the segments its first tile really has, see there).
// grid: the workgroups that pull from the queues (a launch may carry others behind them).
__device__ __forceinline__ TileItem next_item(const TileQueue& q, int32_t* s_ticket, bool first, int parts, unsigned grid) {
    int idx = blockIdx.x / q.nq;                     // first pull: the slot number, no atomic
    if (!first) {
        __syncthreads();                             // the previous tile's LDS traffic is complete
        if (threadIdx.x == 0) *s_ticket = (int)(grid / q.nq) + atomicAdd(&q.counter[blockIdx.x % q.nq], 1);
        __syncthreads();
        idx = *s_ticket;
    }
    return item_at(q, idx, parts);
}

}  // namespace gfl
