// Per-tile sort of the 64-bit (depth bits << 32 | splat id) keys -- included by gfl_bin.hip.
//
// One workgroup of 256 lanes per tile, keys in REGISTERS: a lane holds E = 1, 2, 4, 8 or 16
// consecutive keys (E chosen per tile from its length, up to 4096 keys), padded with +inf.
// Bitonic network; a compare-exchange partner is
//   - in the same lane            when the stride is below E        (register swap),
//   - in the same wave            when it is below 64 E             (wave shuffle, no barrier),
//   - in another wave otherwise   (three steps per sort)            (2 KB LDS exchange buffer).
// A first version sorted in LDS with a barrier per pass: a single 300-key tile then cost ~25 us
// of barrier latency and set the duration of the whole launch.
#pragma once

namespace gfl {

template <int E>
__device__ __forceinline__ void sort_tile_regs(unsigned long long* __restrict__ seg, int n, int npow,
                                               unsigned long long* __restrict__ sk, unsigned long long (&key)[E]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = tid * E + e;
        key[e] = idx < n ? seg[idx] : ~0ull;
    }
    for (int k = 2; k <= npow; k <<= 1) {
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j < E) {
                // partner inside the lane: constant register indices for every possible j
#pragma unroll
                for (int jj = 1; jj < E; jj <<= 1) {
                    if (jj != j) continue;
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
            } else {
                const int tj = j / E;                  // lane distance of the partner
                const bool lower = (tid & tj) == 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    unsigned long long other;
                    if (tj >= 64) {
                        __syncthreads();
                        sk[tid] = key[e];
                        __syncthreads();
                        other = sk[tid ^ tj];
                    } else {
                        other = __shfl_xor(key[e], tj);
                    }
                    const bool up = ((tid * E + e) & k) == 0;
                    const bool take_min = lower == up;
                    const unsigned long long mn = key[e] < other ? key[e] : other;
                    const unsigned long long mx = key[e] < other ? other : key[e];
                    key[e] = take_min ? mn : mx;
                }
            }
        }
    }
}

template <int E>
__device__ __forceinline__ void sort_tile_and_emit(unsigned long long* __restrict__ seg, int n, int npow,
                                                   unsigned long long* __restrict__ sk, int start, int tile,
                                                   int32_t* __restrict__ ids, const float* __restrict__ slot_rec,
                                                   int32_t* __restrict__ slot_inv, int32_t* __restrict__ slot_pool,
                                                   int gx, int gy) {
    unsigned long long key[E];
    sort_tile_regs<E>(seg, n, npow, sk, key);
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = tid * E + e;
        if (idx < n) {
            seg[idx] = key[e];
            const int g = (int32_t)(unsigned)(key[e] & 0xffffffffull);
            ids[start + idx] = g;
            if (slot_inv) write_slot(slot_rec, slot_inv, slot_pool, g, tile, gx, gy, start + idx);
        }
    }
}

__global__ void __launch_bounds__(256) bin_tile_sort_kernel(const int32_t* __restrict__ offsets, int K_cap,
                                                            unsigned long long* __restrict__ keys,
                                                            int32_t* __restrict__ ids,
                                                            int32_t* __restrict__ tile_range,
                                                            const float* __restrict__ slot_rec,
                                                            int32_t* __restrict__ slot_inv,
                                                            int32_t* __restrict__ slot_pool, int gx, int gy) {
    __shared__ unsigned long long sk[256];
    const int tile = blockIdx.x;
    const int start = min(offsets[tile], K_cap);
    const int end = min(offsets[tile + 1], K_cap);
    const int n = end - start;
    if (threadIdx.x == 0) {
        tile_range[2 * tile] = n > 0 ? start : 0;
        tile_range[2 * tile + 1] = n > 0 ? end : 0;
    }
    if (n <= 0) return;
    unsigned long long* seg = keys + start;
    int npow = 2;
    while (npow < n) npow <<= 1;
    if (n <= 256) {
        sort_tile_and_emit<1>(seg, n, npow, sk, start, tile, ids, slot_rec, slot_inv, slot_pool, gx, gy);
    } else if (n <= 512) {
        sort_tile_and_emit<2>(seg, n, npow, sk, start, tile, ids, slot_rec, slot_inv, slot_pool, gx, gy);
    } else if (n <= 1024) {
        sort_tile_and_emit<4>(seg, n, npow, sk, start, tile, ids, slot_rec, slot_inv, slot_pool, gx, gy);
    } else if (n <= 2048) {
        sort_tile_and_emit<8>(seg, n, npow, sk, start, tile, ids, slot_rec, slot_inv, slot_pool, gx, gy);
    } else if (n <= 4096) {
        sort_tile_and_emit<16>(seg, n, npow, sk, start, tile, ids, slot_rec, slot_inv, slot_pool, gx, gy);
    } else {
        // oversized segment: the all-ascending network directly on global memory (one CU, its
        // own L1; the barriers between passes order the accesses)
        bitonic_sort((volatile unsigned long long*)seg, n);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int g = (int32_t)(unsigned)(seg[i] & 0xffffffffull);
            ids[start + i] = g;
            if (slot_inv) write_slot(slot_rec, slot_inv, slot_pool, g, tile, gx, gy, start + i);
        }
    }
}

}  // namespace gfl
