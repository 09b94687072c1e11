// Per-tile sort of the 64-bit (depth bits << 32 | splat id) keys -- included by gfl_bin.hip.
//
// One workgroup of up to SORT_THREADS (512) lanes per tile, keys in REGISTERS: a lane holds E = 1, 2, 4 or 8
// consecutive keys (E chosen per tile from its length, up to 4096 keys), padded with +inf; the waves a tile does
// not need (four of the eight for the typical 140-key list) leave before the first barrier.  Round 1 ran 256 lanes:
// a 1 200-key tile then sat on four lone waves with 8 keys per lane (24 us, the launch's duration on real fits);
// with eight waves it has 4 keys per lane and two waves per SIMD: 24 -> 18 us there, 13.2 -> 12.5 us on the bench scene
// (sixteen waves: 21 and 17 us -- every tile then pays for the wider workgroup).
// Bitonic network; a compare-exchange partner is
//   - in the same lane            when the stride is below E        (register swap),
//   - in the same wave            when it is below 64 E             (DPP / permlane swap, no LDS),
//   - in another wave otherwise   (a few steps per sort)            (16 KB LDS exchange buffer, four keys per barrier pair).
// A first version sorted in LDS with a barrier per pass: a single 300-key tile then cost ~25 us
// of barrier latency and set the duration of the whole launch.
#pragma once

namespace gfl {

#ifdef GFL_TRACE
__device__ long long g_sort_trace[16384 * 4];   // analysis build: start / keys loaded / sorted / done, one row per tile
#define SORT_TRACE(slot) if (threadIdx.x == 0 && blockIdx.x < 16384) g_sort_trace[sort_trace_row * 4 + (slot)] = wall_clock64()
#else
#define SORT_TRACE(slot)
#endif

// Value of lane (lane ^ D) for a wave-uniform D in {1, 2, 4, 8, 16, 32}, on the VALU only: DPP
// quad permutes / row shifts / row rotate inside a 16-lane row, v_permlane16_swap / v_permlane32_swap
// (gfx950) across rows.  __shfl_xor goes through ds_bpermute (address VGPR + an LDS-pipe round trip
// per 32 bits); with ~40 dependent exchange steps per tile the sort network cost 14 of the
// launch's 24 us that way.
template <int D>
__device__ __forceinline__ unsigned xor_lane_u32(unsigned v, int lane) {
    const int x = (int)v;
    if constexpr (D == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    else if constexpr (D == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (D == 4) {
        int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);               // row_shl:4 -> banks 0, 2
        t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);                   // row_shr:4 -> banks 1, 3
        return (unsigned)t;
    } else if constexpr (D == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);   // row_ror:8
    else if constexpr (D == 16) {
        float a = __builtin_bit_cast(float, v), b = a;
        permlane16_swap(a, b);               // a = rows {0,0,2,2} of v, b = rows {1,1,3,3}
        return __builtin_bit_cast(unsigned, (lane & 16) ? a : b);
    } else {
        float a = __builtin_bit_cast(float, v), b = a;
        permlane32_swap(a, b);               // a = {lo, lo}, b = {hi, hi}
        return __builtin_bit_cast(unsigned, (lane & 32) ? a : b);
    }
}

// one compare-exchange step of the network with the partner D lanes away, all E keys of the lane
template <int E, int D>
__device__ __forceinline__ void exchange_in_wave(unsigned long long (&key)[E], int k, int tid) {
    const bool lower = (tid & D) == 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned lo = xor_lane_u32<D>((unsigned)key[e], tid), hi = xor_lane_u32<D>((unsigned)(key[e] >> 32), tid);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        const bool up = ((tid * E + e) & k) == 0;
        const bool take_min = lower == up;
        key[e] = ((key[e] < other) == take_min) ? key[e] : other;     // keys are unique (or equal padding)
    }
}

// (round 4, -DGFL_SORT_THREADS=256 on the build with the pivot split: bench-window sort 15.6 against 16.1 us by events,
//  4-frame clip fits 0.458 / 0.473 s against 0.452 / 0.459 s -- the halves of the piles' lists on four waves again)
#ifndef GFL_SORT_THREADS
#define GFL_SORT_THREADS 512
#endif
constexpr int SORT_THREADS = GFL_SORT_THREADS;
constexpr int SORT_XB = 4;         // keys of a lane exchanged through LDS per pair of barriers (16 KB of LDS)


template <int E>
__device__ __forceinline__ void sort_tile_regs(const unsigned long long* seg, int n, int npow,
                                               unsigned long long* __restrict__ sk, unsigned long long (&key)[E],
                                               int sort_trace_row, bool seg_in_sk = false) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = tid * E + e;
        key[e] = idx < n ? seg[idx] : ~0ull;
    }
    if (seg_in_sk) __syncthreads();      // (the keys were compacted into the exchange buffer: every lane has its own before it is re-used)
#ifdef GFL_TRACE
    if (key[0] == 1ull) return;      // (forces the loads to complete before the time stamp)
    SORT_TRACE(1);
#endif
    // One merge step per k; inside it the partner distance runs k/2, k/4, ..., 1: first the distances
    // that reach other waves, then the lane distances 32 .. 1 as six guarded blocks with the distance a
    // compile-time constant (a switch on a run-time distance cost as many scalar instructions per
    // step as the exchange itself), then the distances inside the lane.
    for (int k = 2; k <= npow; k <<= 1) {
        const int j0 = k >> 1;
        for (int j = j0; j >= 64 * E; j >>= 1) {
            // partner in another wave: through LDS, up to SORT_XB of the lane's keys per pair of barriers (one key per
            // pair made the 2 048-key tiles -- six such steps of 4 keys each, 48 barriers -- the tail of the launch)
            const int tj = j / E;
            const bool lower = (tid & tj) == 0;
            constexpr int XB = E < SORT_XB ? E : (E > 4 ? 2 : SORT_XB);     // (E = 8 with four in flight spills registers)
#pragma unroll
            for (int e0 = 0; e0 < E; e0 += XB) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < XB; ++e) sk[e * SORT_THREADS + tid] = key[e0 + e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < XB; ++e) {
                    const unsigned long long other = sk[e * SORT_THREADS + (tid ^ tj)];
                    const bool up = ((tid * E + e0 + e) & k) == 0;
                    const bool take_min = lower == up;
                    key[e0 + e] = ((key[e0 + e] < other) == take_min) ? key[e0 + e] : other;
                }
            }
        }
        if (j0 >= 32 * E) exchange_in_wave<E, 32>(key, k, tid);
        if (j0 >= 16 * E) exchange_in_wave<E, 16>(key, k, tid);
        if (j0 >= 8 * E) exchange_in_wave<E, 8>(key, k, tid);
        if (j0 >= 4 * E) exchange_in_wave<E, 4>(key, k, tid);
        if (j0 >= 2 * E) exchange_in_wave<E, 2>(key, k, tid);
        if (j0 >= E) exchange_in_wave<E, 1>(key, k, tid);
#pragma unroll
        for (int jj = E >> 1; jj >= 1; jj >>= 1) {
            if (j0 < jj) continue;
            // partner inside the lane
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (e & jj) continue;
                const int f = e | jj;
                const bool up = ((tid * E + e) & k) == 0;
                const unsigned long long x = key[e], y = key[f];
                const bool sw = (x > y) == up;
                key[e] = sw ? y : x;
                key[f] = sw ? x : y;
            }
        }
    }
}

template <int E>
__device__ __forceinline__ void sort_tile_and_emit(const unsigned long long* seg, int n, int npow,
                                                   unsigned long long* __restrict__ sk, int start, int tile,
                                                   int32_t* __restrict__ ids, bool seg_in_sk = false) {
    unsigned long long key[E];
    const int sort_trace_row = tile;
    (void)sort_trace_row;
    sort_tile_regs<E>(seg, n, npow, sk, key, tile, seg_in_sk);
    SORT_TRACE(2);
    const int tid = threadIdx.x;
    // the sorted ids.  (Until round 5 this also filled a table of every pair's list position for the per-splat launch's gather;
    //  the pair rows now lie where that launch finds them without one, gfl_fit.hpp: FitWs.pair_grad.)
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (tid * E + e < n) ids[start + tid * E + e] = (int32_t)(unsigned)(key[e] & 0xffffffffull);
}

// Register budget: 64 VGPRs (eight waves per SIMD).  With a 16-keys-per-lane variant in the same kernel the compiler
// needed 95 VGPRs; up to 8 keys per lane it needs 50.  Lists beyond 4096 keys are sorted in global memory.
__global__ void __launch_bounds__(SORT_THREADS, 8) bin_tile_sort_kernel(const int32_t* __restrict__ offsets, int K_cap,
                                                                        unsigned long long* __restrict__ keys,
                                                                        int32_t* __restrict__ ids,
                                                                        int32_t* __restrict__ tile_range, int gx, int gy,
                                                                        const int4* __restrict__ order,
                                                                        const int32_t* __restrict__ fill,
                                                                        int32_t* __restrict__ counts_out,
                                                                        int32_t* __restrict__ void_words) {
    __shared__ unsigned long long sk[SORT_XB * SORT_THREADS];      // cross-wave exchange buffer
    // XCD x sorts one contiguous range of tiles (the dispatcher places workgroup b on XCD b % 8): the records the slot
    // table needs (uv, radius of every key's splat) are those of neighbouring tiles; with block = tile every XCD pulled
    // all of them through its own L2 (22 MB read for 3.9 MB of keys, rocprofv3 FETCH_SIZE)
    // `order` (fused iteration): the XCD's tiles with the longest lists first -- {tile, start, end} per position, written
    // by the scatter launch.  Four 512-lane workgroups fit a CU at a time and a launch has six per CU: a 600-key tile
    // that started in the second round (4-6 us in) was the end of the launch (tools/sort_trace.py).
    // Round 4: a list of more than SORT_SPLIT_MIN keys -- the piles densification leaves in single tiles: 1 000-1 700 keys, a
    // 2 048-key network of 16 us on eight waves while every other tile is done after 7-10 us (tools/sort_trace.py --fit) --
    // is cut at a pivot key and sorted by TWO workgroups: this tile's own sorts the keys below the pivot, one of the
    // SORT_MAX_SPLIT extra workgroups at the FRONT of the grid the others, each a list of half the length (fewer network
    // steps, half the keys per lane), straight into its part of the tile's range.  No exchange between the two: both read
    // all keys, both find the same pivot (the median of 32 keys at fixed positions of the unsorted list) and count the same
    // lower half.  Which tiles: order[pos].w = 1 + j and trailer[1 + j] = pos for the j-th of them, trailer[0] = their number
    // (the trailer follows order[T]; written with the order by the scatter launch).  (Two workgroups for EVERY position,
    // the second leaving at once where there is nothing to split, cost the bench scene 6.7 us: 1 620 more 512-lane
    // workgroups to dispatch.)
    // `fill` (reserved tile regions, gfl_fused.hip): the order was written at the END of the iteration before, with every
    // tile's region {start, capacity} in place of {start, end}; the list's length is what the binning launch counted into
    // fill[position] (capped: what did not fit was not written, and the iteration steps nothing).  counts_out[tile] = what the
    // tile wanted (the next regions are sized by it).  void_words: {this iteration is void, a tile outgrew its region}: the
    // binning launch sets the second, this launch moves it into the first, where the update launches look -- a word nobody
    // writes while they run.
    if (void_words && blockIdx.x == 0 && threadIdx.x == 0) {
        void_words[0] = void_words[1];
        void_words[1] = 0;
    }
    const int T_all = gx * gy;
    int lb, half = 0;
    bool flagged = false;
    if (order) {
        const int32_t* trailer = reinterpret_cast<const int32_t*>(order + T_all);
        if ((int)blockIdx.x < SORT_MAX_SPLIT) {
            if ((int)blockIdx.x >= min(trailer[0], SORT_MAX_SPLIT)) return;
            lb = trailer[1 + blockIdx.x];
            half = 1;
        } else {
            lb = xcd_logical_block((int)blockIdx.x - SORT_MAX_SPLIT, T_all);
        }
    } else {
        lb = xcd_logical_block((int)blockIdx.x, (int)gridDim.x);
    }
    int tile = lb, start, end, wanted = 0;
    if (order) {
        wanted = fill ? fill[lb] : 0;                // (by position: requested beside the order entry, not behind it)
        const int4 it = order[lb];
        tile = it.x;
        start = min(it.y, K_cap);
        end = min(fill ? it.y + min(wanted, it.z) : it.z, K_cap);
        flagged = it.w != 0;
    } else {
        start = min(offsets[tile], K_cap);
        end = min(offsets[tile + 1], K_cap);
    }
    const int sort_trace_row = tile;
    (void)sort_trace_row;
    SORT_TRACE(0);
    const int n = end - start;
    if (threadIdx.x == 0 && half == 0) {
        tile_range[2 * tile] = n > 0 ? start : 0;
        tile_range[2 * tile + 1] = n > 0 ? end : 0;
        if (counts_out) counts_out[tile] = fill ? wanted : max(n, 0);
    }
    if (n <= 0) return;
    unsigned long long* seg = keys + start;
    const bool split = flagged && n >= 64 && n <= 2048;       // (both workgroups of a tile decide alike)
    if (half == 1 && !split) return;
    if (split) {
        // ---- pivot: the median of 32 keys at fixed positions (keys are unique: ranks are)
        __shared__ unsigned long long s_pivot;
        __shared__ int32_t s_cnt[SORT_THREADS / 64 + 1];
        const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
        if (wid == 0) {
            const unsigned long long mine = seg[(int)(((long long)n * (2 * (lane & 31) + 1)) >> 6)];
            int rank = 0;
            for (int j = 0; j < 32; ++j) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, j);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), j);
                rank += ((((unsigned long long)hi << 32) | lo) < mine) ? 1 : 0;
            }
            if (lane < 32 && rank == 16) s_pivot = mine;
        }
        __syncthreads();
        const unsigned long long pivot = s_pivot;
        // ---- this half's keys, compacted into the exchange buffer (order does not matter: they are sorted next)
        constexpr int EP = 2048 / SORT_THREADS;      // (n <= 2048)
        unsigned long long k4[EP];
        bool keep[EP];
        int mine_cnt = 0;
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            const int idx = tid + e * SORT_THREADS;              // (coalesced: the order inside the list is irrelevant here)
            k4[e] = idx < n ? seg[idx] : ~0ull;
            keep[e] = idx < n && ((k4[e] < pivot) == (half == 0));
            mine_cnt += (int)__popcll(__ballot(keep[e]));
        }
        if (lane == 0) s_cnt[wid] = mine_cnt;
        __syncthreads();
        int base = 0, m = 0;
#pragma unroll
        for (int w = 0; w < SORT_THREADS / 64; ++w) {
            const int c = s_cnt[w];
            base += w < wid ? c : 0;
            m += c;
        }
        const int n_lower = half == 0 ? m : n - m;
        // (a split list has at most 2048 keys: with SORT_XB * SORT_THREADS = 2048 slots in the exchange buffer -- the shipped 512
        //  lanes -- either half always fits and the fallback below does not exist in the binary; a 256-lane build keeps it)
        constexpr bool HALVES_ALWAYS_FIT = SORT_XB * SORT_THREADS >= 2048;
        if (HALVES_ALWAYS_FIT || (m <= SORT_XB * SORT_THREADS && n - m <= SORT_XB * SORT_THREADS)) {
#pragma unroll
            for (int e = 0; e < EP; ++e) {
                const unsigned long long bal = __ballot(keep[e]);
                if (keep[e]) sk[base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0))] = k4[e];
                base += (int)__popcll(bal);
            }
            __syncthreads();
            if (m <= 0) return;
            int np2 = 2;
            while (np2 < m) np2 <<= 1;
            const int E2 = np2 <= SORT_THREADS ? 1 : (np2 <= 2 * SORT_THREADS ? 2 : 4);
            const int start2 = start + (half == 0 ? 0 : n_lower);
            // (waves the shorter list does not need leave here; the barrier inside the key load counts only the others)
            if ((int)threadIdx.x >= max(np2 / E2, 64)) return;
            if (E2 == 1) sort_tile_and_emit<1>(sk, m, np2, sk, start2, tile, ids, true);
            else if (E2 == 2) sort_tile_and_emit<2>(sk, m, np2, sk, start2, tile, ids, true);
            else sort_tile_and_emit<4>(sk, m, np2, sk, start2, tile, ids, true);
            SORT_TRACE(3);
            return;
        }
        // (narrower builds only: a pivot so lopsided that one half does not fit the buffer -- sampled medians do not do that,
        //  but it must be right if they do --: the first workgroup sorts the whole list the ordinary way)
        if (half == 1) return;
        __syncthreads();
    }
    int npow = 2;
    while (npow < n) npow <<= 1;
    // keys per lane: the smallest E with npow <= SORT_THREADS * E; the lanes beyond npow / E are not needed (whole
    // waves of them leave here, before any barrier)
    const int E = npow <= SORT_THREADS ? 1 : (npow <= 2 * SORT_THREADS ? 2 : (npow <= 4 * SORT_THREADS ? 4 : 8));
    if (npow <= 8 * SORT_THREADS && (int)threadIdx.x >= max(npow / E, 64)) return;
    if (npow <= SORT_THREADS) {
        sort_tile_and_emit<1>(seg, n, npow, sk, start, tile, ids);
        SORT_TRACE(3);
    } else if (npow <= 2 * SORT_THREADS) {
        sort_tile_and_emit<2>(seg, n, npow, sk, start, tile, ids);
        SORT_TRACE(3);
    } else if (npow <= 4 * SORT_THREADS) {
        sort_tile_and_emit<4>(seg, n, npow, sk, start, tile, ids);
        SORT_TRACE(3);
    } else if (npow <= 8 * SORT_THREADS) {
        sort_tile_and_emit<8>(seg, n, npow, sk, start, tile, ids);
        SORT_TRACE(3);
    } else {
        // the all-ascending network directly on global memory (one CU, its own L1; the
        // barriers between passes order the accesses): slow, for lists no scene here produces
        bitonic_sort((volatile unsigned long long*)seg, n);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int g = (int32_t)(unsigned)(seg[i] & 0xffffffffull);
            ids[start + i] = g;
        }
    }
}

}  // namespace gfl
