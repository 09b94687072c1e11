// Where does the dispatcher put workgroup b?  (analysis tool, not part of the library)
//   hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o tools/placement_probe.bin
//   gpurun -- ./tools/placement_probe.bin 1024 18000
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) probe(unsigned* out, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    lds[threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = hw;
        out[3 * blockIdx.x + 1] = xcc;
        out[3 * blockIdx.x + 2] = (unsigned)(t0 & 0xffffffffu);
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const int lds = argc > 2 ? atoi(argv[2]) : 18000;
    const int spin = argc > 3 ? atoi(argv[3]) : 20000;
    unsigned* d;
    hipMalloc(&d, n * 12);
    probe<<<n, 256, lds>>>(d, spin);
    hipDeviceSynchronize();
    probe<<<n, 256, lds>>>(d, spin);
    hipDeviceSynchronize();
    unsigned* h = (unsigned*)malloc(n * 12);
    hipMemcpy(h, d, n * 12, hipMemcpyDeviceToHost);
    unsigned tmin = ~0u;
    for (int b = 0; b < n; ++b) if (h[3 * b + 2] < tmin) tmin = h[3 * b + 2];
    for (int b = 0; b < n; ++b) {
        const unsigned hw = h[3 * b], xcc = h[3 * b + 1] & 0xf;
        printf("b %4d xcc %u se %u sh %u cu %2u simd %u wave %u  t %u\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15,
               (hw >> 4) & 3, hw & 15, h[3 * b + 2] - tmin);
    }
    return 0;
}
