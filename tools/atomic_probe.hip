// Probe for gfx950 (analysis tool, not part of the library): what does it cost when every workgroup of a binning launch
// reserves its part of every tile's list with ONE returning global atomicAdd per (workgroup, tile) -- the step that would
// replace the histogram rows + column scan + scatter launches of the fused iteration by a single launch (DESIGN.md
// section 4, "reserved tile regions")?  118 workgroups x 512 lanes, 1 620 counters, every workgroup touching a share of them.
//   hipcc --offload-arch=gfx950 -O2 tools/atomic_probe.hip -o tools/atomic_probe.bin
//   gpurun -- ./tools/atomic_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// mode 0: no atomics (launch floor)   1: returning atomicAdd, counters 4 bytes apart   2: counters `stride` ints apart
__global__ void __launch_bounds__(512) reserve(int* __restrict__ fill, int T, int stride, int share_pct, int mode,
                                                int* __restrict__ out) {
    __shared__ int cursor[4096];
    const int tid = threadIdx.x;
    int got[8];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int t = tid + k * 512;
        got[k] = 0;
        if (t < T) {
            // a pseudo-random share of the tiles is touched by this workgroup
            const unsigned h = (unsigned)(t * 2654435761u) ^ (unsigned)(blockIdx.x * 40503u);
            const bool touched = (int)((h >> 8) % 100u) < share_pct;
            if (touched && mode) got[k] = atomicAdd(&fill[(size_t)t * stride], 3);
            ++n;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (tid + k * 512 < T) cursor[tid + k * 512] = got[k];
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = cursor[(blockIdx.x * 7) % T] + n;
}

int main() {
    const int T = 1620, NB = 118;
    int *fill, *out;
    hipMalloc(&fill, (size_t)T * 64 * sizeof(int));
    hipMalloc(&out, 4096 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { int mode, stride, pct; const char* what; } cases[] = {
        {0, 1, 100, "no atomics"},
        {1, 1, 100, "4-byte spacing, every tile"},
        {2, 16, 100, "64-byte spacing, every tile"},
        {2, 32, 100, "128-byte spacing, every tile"},
        {1, 1, 60, "4-byte spacing, 60 % of the tiles"},
        {2, 32, 60, "128-byte spacing, 60 % of the tiles"},
        {2, 32, 25, "128-byte spacing, 25 % of the tiles"},
    };
    for (auto& c : cases) {
        for (int nb : {NB, 2 * NB, 4 * NB}) {
            hipMemset(fill, 0, (size_t)T * 64 * sizeof(int));
            for (int w = 0; w < 3; ++w) reserve<<<nb, 512>>>(fill, T, c.stride, c.pct, c.mode, out);
            hipDeviceSynchronize();
            const int reps = 50;
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) reserve<<<nb, 512>>>(fill, T, c.stride, c.pct, c.mode, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-40s workgroups %4d   %7.2f us per launch (back to back)\n", c.what, nb, ms * 1000.f / reps);
        }
    }
    return 0;
}
