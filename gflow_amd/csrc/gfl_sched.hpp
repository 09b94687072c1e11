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
//   weight    the units the backward blend counted on that tile in the PREVIOUS iteration (the
//             scene moves slowly); the list length where there is no history yet;
//   rounds    tiles in order of descending weight, NQ at a time; in every round the queue with
//             the smallest load so far gets the heaviest tile of the round (rank matching: one
//             256-way ranking per round instead of the 1620 sequential steps of greedy LPT;
//             simulated on measured weights: busiest queue 1.06x the mean, LPT 1.03x, plain 1.4x);
//   round 0   the NQ heaviest tiles.  Their waves raise their instruction priority, and in the
//             backward pass each of them is split in two halves of its list that two workgroups
//             of the CU walk side by side (the forward pass leaves a per-pixel checkpoint at the
//             split position).
// The 8 workgroups resident on a CU pull from that CU's queue (first pull = slot number, later
// pulls through a per-queue counter that only those 8 contend for).
// The schedule decides WHERE a tile is processed, never what is computed: results do not depend on
// it, and a dispatcher that places workgroups differently only loses the balance.
#pragma once

namespace gfl {

constexpr int SCHED_BLOCK = 512;          // threads of the scheduling workgroup (= BIN_BLOCK)
constexpr int SCHED_BINS = 2 * SCHED_BLOCK;
constexpr int SCHED_MAX_QUEUES = 512;
constexpr int SCHED_MAX_WEIGHT = 1 << 20;

struct Sched {
    int32_t* work;       // [T]        feedback: units of the last backward blend per tile (0: none)
    int32_t* order;      // [T]        scratch: tiles by descending weight
    int32_t* seq;        // [rounds nq] item r of queue c at seq[r * nq + c]: tile | priority << 28, or -1
    int32_t* counters;   // [2 nq]     pull counters, forward then backward
    int nq;              // queues (= CUs)
};

__host__ __device__ inline int sched_rounds(int T, int nq) { return (T + nq - 1) / nq; }

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

// w: LDS scratch of T ints; wsum: LDS scratch of SCHED_BLOCK / 64 ints.  Whole workgroup.
__device__ void schedule_tiles(const int32_t* __restrict__ tile_counts, int T, const Sched sc, int32_t* w,
                               int32_t* wsum) {
    __shared__ int32_t bins[SCHED_BINS];
    __shared__ int32_t chunk[SCHED_MAX_QUEUES];
    __shared__ int32_t s_max, s_lo, s_hi;
    const int tid = threadIdx.x;
    const int NQ = sc.nq;
    // ---- 1. weights
    if (tid == 0) s_max = 1;
    for (int b = tid; b < SCHED_BINS; b += SCHED_BLOCK) bins[b] = 0;
    for (int c = tid; c < 2 * NQ; c += SCHED_BLOCK) sc.counters[c] = 0;
    __syncthreads();
    int local = 0, lmax = 1;
    for (int t = tid; t < T; t += SCHED_BLOCK) {
        int x = sc.work[t];
        if (x <= 0 || x > SCHED_MAX_WEIGHT) x = min(max(tile_counts[t], 1), SCHED_MAX_WEIGHT);
        sc.work[t] = 0;                  // the backward blend adds this iteration's units
        w[t] = x;
        local += x;
        lmax = max(lmax, x);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, __shfl_xor(lmax, off));
    if ((tid & 63) == 0) atomicMax(&s_max, lmax);
    int W_total;
    sched_block_scan(local, wsum, W_total);          // (contains the barriers that publish s_max, w)
    int shift = 0;
    while ((s_max >> shift) >= SCHED_BINS) ++shift;
    // ---- 2. tiles by descending weight (counting sort on the quantised weight; order within a
    //         bin is whatever the LDS atomics produce)
    for (int t = tid; t < T; t += SCHED_BLOCK) atomicAdd(&bins[SCHED_BINS - 1 - (w[t] >> shift)], 1);
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
        const int pos = atomicAdd(&bins[SCHED_BINS - 1 - (w[t] >> shift)], 1);
        sc.order[pos] = t;
    }
    __threadfence_block();
    __syncthreads();
    // ---- 3. rounds of rank matching; thread c < NQ owns queue c.  The ranking is a counting
    //         sort of the queue loads quantised to SCHED_BINS levels between the smallest and
    //         the largest load (an exact 256-way ranking by comparison took 5 us per round).
    const int target = (W_total + NQ - 1) / NQ;
    const int rounds = sched_rounds(T, NQ);
    int my_load = 0;
    // the round's tiles go through LDS (chunk), fetched from `order` one round ahead: a dependent
    // global load per round would cost more than the ranking
    int nxt = (tid < NQ && tid < T) ? sc.order[tid] : -1;
    for (int r = 0; r < rounds; ++r) {
        const int cur = nxt;
        {
            const int p = (r + 1) * NQ + tid;
            nxt = (r + 1 < rounds && tid < NQ && p < T) ? sc.order[p] : -1;
        }
        int rank = tid;                              // round 0: all loads are zero
        __syncthreads();
        if (tid < NQ) chunk[tid] = cur;
        if (r > 0) {
            if (tid == 0) { s_lo = 0x7fffffff; s_hi = 0; }
            bins[2 * tid] = 0;
            bins[2 * tid + 1] = 0;
            __syncthreads();
            {
                // (256 LDS atomics on one word cost ~4 us: reduce in the wave first)
                int mn = tid < NQ ? my_load : 0x7fffffff, mx = tid < NQ ? my_load : 0;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    mn = min(mn, __shfl_xor(mn, off));
                    mx = max(mx, __shfl_xor(mx, off));
                }
                if ((tid & 63) == 0) { atomicMin(&s_lo, mn); atomicMax(&s_hi, mx); }
            }
            __syncthreads();
            const int lo = s_lo;
            const float q = (float)(SCHED_BINS - 1) / (float)(s_hi - lo + 1);
            const int bin = tid < NQ ? (int)((float)(my_load - lo) * q) : 0;
            if (tid < NQ) atomicAdd(&bins[bin], 1);
            __syncthreads();
            {
                const int x = bins[2 * tid], y = bins[2 * tid + 1];
                int tot;
                const int excl = sched_block_scan(x + y, wsum, tot);
                bins[2 * tid] = excl;
                bins[2 * tid + 1] = excl + x;
            }
            __syncthreads();
            if (tid < NQ) rank = atomicAdd(&bins[bin], 1);
        } else {
            __syncthreads();
        }
        if (tid < NQ) {
            int item = -1;
            const int tile = rank < NQ ? chunk[rank] : -1;
            if (tile >= 0) {
                const int wt = w[tile];
                my_load += wt;
                const int prio = r > 0 ? 0 : (wt * 5 >= target * 2 ? 3 : (wt * 4 >= target ? 2 : 1));
                item = tile | (prio << 28);
            }
            sc.seq[r * NQ + tid] = item;
        }
    }
}

// ---- consumer side
struct TileQueue {
    const int32_t* seq;
    int32_t* counter;    // [nq] for this launch
    int nq;
    int rounds;
};

// Item of a queue.  part: 0 = the whole tile; 1 / 2 = far / near half of the queue's heaviest tile
// (backward launch only).
struct TileItem {
    int tile;     // -1: the queue is empty
    int part;
    int queue;
};

// list position where a heavy tile is split (a multiple of 64, 0 = not split)
__device__ __forceinline__ int heavy_split(int total) { return total > 128 ? ((total >> 1) + 63) & ~63 : 0; }

// next item of this workgroup's queue.  Whole workgroup.  `split`: backward launch.
__device__ __forceinline__ TileItem next_item(const TileQueue& q, int32_t* s_ticket, bool first, bool split) {
    TileItem it;
    it.queue = blockIdx.x % q.nq;
    it.part = 0;
    it.tile = -1;
    int idx = blockIdx.x / q.nq;                     // first pull: the slot number, no atomic
    if (!first) {
        __syncthreads();                             // the previous tile's LDS traffic is complete
        if (threadIdx.x == 0) *s_ticket = (int)(gridDim.x / q.nq) + atomicAdd(&q.counter[it.queue], 1);
        __syncthreads();
        idx = *s_ticket;
    }
    // backward: the round-0 tile is two items
    const int r = split ? max(idx - 1, 0) : idx;
    if (r >= q.rounds) return it;
    const int item = q.seq[r * q.nq + it.queue];
    if (item < 0) return it;                         // (only the last round has holes)
    it.tile = item & 0x0fffffff;
    if (split && idx < 2) it.part = idx + 1;
    const int prio = item >> 28;
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    return it;
}

}  // namespace gfl
