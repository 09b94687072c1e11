// The ORDER the tile sort takes the tiles in (every XCD's longest lists first, the piles' lists flagged for the two-workgroup
// split) and, with RESERVE, the tile regions of the next iteration.  Included by gfl_fit_bin.hip (one workgroup of the scatter
// launch builds the order of THIS iteration's sort) and by gfl_fit_splat.hip (one workgroup of the per-splat launch, at the
// end of an iteration, reserves the regions and writes the order of the NEXT one).
#pragma once
#include "gfl_fit.hpp"

namespace gfl {

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
// region, clamped to K_cap, *total = the pairs of the iteration that ends here.  (Regions that reach beyond K_cap are cut
// short by the binning launch that uses them, and THAT launch raises the lists' overflow flag if a key then does not fit:
// this workgroup runs inside the launch whose row blocks read the flag, so it must not write it -- a flag raised here made
// early row blocks step and late ones skip)
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

}  // namespace gfl
