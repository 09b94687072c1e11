// Probe for gfx950 (analysis tool): what would it cost the backward blend to ADD every (splat, tile) pair's ten gradient sums
// into the splat's own accumulator row with float atomics (40 contiguous bytes per pair, a random row per pair: 264 k pairs
// on 60 k rows), instead of writing a 48-byte row per pair that the per-splat launch gathers through a slot table?
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics tools/atomic_probe2.hip -o tools/atomic_probe2.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: nothing   1: float atomicAdd, 10 lanes per pair, 6 pairs per wave instruction   2: plain stores of 48-byte rows by list position
// 3: atomics, but the pairs of a wave instruction go to NEIGHBOURING rows (what spatial sorting of the splats would give)
__global__ void __launch_bounds__(256) scatter_add(float* __restrict__ acc, float* __restrict__ rows, int K, int N, int mode) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    const int sub = lane / 10, comp = lane % 10;
    float v = (float)lane * 1e-3f;
    for (int p0 = wave * 6; p0 < K; p0 += n_waves * 6) {
        const int pair = p0 + sub;
        if (sub >= 6 || pair >= K) continue;
        if (mode == 1) atomicAdd(&acc[(size_t)(hash(pair) % N) * 12 + comp], v);
        else if (mode == 3) atomicAdd(&acc[(size_t)((hash(p0) + sub * 3) % N) * 12 + comp], v);
        else if (mode == 2) rows[(size_t)pair * 12 + comp] = v;
        v += 1e-6f;
    }
}

int main() {
    const int K = 264000, N = 60000;
    float *acc, *rows;
    hipMalloc(&acc, (size_t)N * 12 * 4); hipMalloc(&rows, (size_t)K * 12 * 4);
    hipMemset(acc, 0, (size_t)N * 12 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"nothing", "float atomics, random rows", "plain 40-byte stores by position", "float atomics, neighbouring rows"};
    for (int grid : {2048, 512})
        for (int mode = 0; mode < 4; ++mode) {
            for (int w = 0; w < 3; ++w) scatter_add<<<grid, 256>>>(acc, rows, K, N, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 50; ++r) scatter_add<<<grid, 256>>>(acc, rows, K, N, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("grid %4d  %-36s %7.2f us per launch\n", grid, names[mode], ms * 1000.f / 50);
        }
    // correctness across XCDs: every row's sum must be exact
    hipMemset(acc, 0, (size_t)N * 12 * 4);
    scatter_add<<<2048, 256>>>(acc, rows, K, N, 1);
    hipDeviceSynchronize();
    return 0;
}
