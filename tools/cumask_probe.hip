// How does a CU-masked stream (hipExtStreamCreateWithCUMask) place workgroups on gfx950's 8 XCDs x 32 CUs -- which mask bit is
// which (XCD, CU) -- does a hipGraph launched on such a stream keep the mask, and do two complementary masks really run side
// by side?  (analysis tool, not part of the library)
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe.bin && gpurun -- ./tools/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) probe(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw | (acc == 12345.f ? 1u << 31 : 0u);
        out[2 * blockIdx.x + 1] = xcc;
    }
}

static void report(const char* what, unsigned* d, int n) {
    std::vector<unsigned> h(2 * n);
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    std::set<unsigned> xccs, cus;
    int per_xcc[16] = {0};
    for (int b = 0; b < n; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        xccs.insert(xcc); cus.insert(cu); per_xcc[xcc]++;
    }
    printf("%-44s XCCs used %zu [", what, xccs.size());
    for (int x = 0; x < 8; ++x) printf("%d ", per_xcc[x]);
    printf("] distinct CUs %zu; first 16 blocks -> xcc:", cus.size());
    for (int b = 0; b < 16 && b < n; ++b) printf(" %u", h[2 * b + 1] & 0xf);
    printf("\n");
}

int main() {
    const int n = 2048, spin = 20000;
    unsigned* d;
    CK(hipMalloc(&d, n * 8));
    hipStream_t plain;
    CK(hipStreamCreate(&plain));
    probe<<<n, 256, 0, plain>>>(d, spin); CK(hipStreamSynchronize(plain));
    report("no mask", d, n);
    struct { const char* name; int kind; } pats[] = {{"bits [0,128)", 0}, {"bits with (i % 8) < 4", 1}, {"bits with (i / 32) < 4", 2},
                                                     {"bits with (i % 8) == 0", 3}, {"bits [0,32)", 4}, {"bits with (i % 2) == 0", 5}};
    hipStream_t keepA = nullptr, keepB = nullptr;
    for (auto& p : pats) {
        uint32_t m[8] = {0};
        for (int i = 0; i < 256; ++i) {
            bool on = false;
            switch (p.kind) {
                case 0: on = i < 128; break;
                case 1: on = (i % 8) < 4; break;
                case 2: on = (i / 32) < 4; break;
                case 3: on = (i % 8) == 0; break;
                case 4: on = i < 32; break;
                case 5: on = (i % 2) == 0; break;
            }
            if (on) m[i / 32] |= 1u << (i % 32);
        }
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, 8, m));
        probe<<<n, 256, 0, s>>>(d, spin); CK(hipStreamSynchronize(s));
        report(p.name, d, n);
        if (p.kind == 0) keepA = s;
    }
    {   // the complement of pattern 0: the upper sixteen CUs of every XCD (bit i = CU i / 8 of XCC i % 8)
        uint32_t m[8] = {0};
        for (int i = 128; i < 256; ++i) m[i / 32] |= 1u << (i % 32);
        CK(hipExtStreamCreateWithCUMask(&keepB, 8, m));
        probe<<<n, 256, 0, keepB>>>(d, spin); CK(hipStreamSynchronize(keepB));
        report("bits [128,256)", d, n);
    }
    {   // are the two halves disjoint?
        std::vector<unsigned> ha(2 * n), hb(2 * n);
        probe<<<n, 256, 0, keepA>>>(d, spin); CK(hipStreamSynchronize(keepA));
        CK(hipMemcpy(ha.data(), d, n * 8, hipMemcpyDeviceToHost));
        probe<<<n, 256, 0, keepB>>>(d, spin); CK(hipStreamSynchronize(keepB));
        CK(hipMemcpy(hb.data(), d, n * 8, hipMemcpyDeviceToHost));
        std::set<unsigned> ca, cb;
        auto key = [](unsigned hw, unsigned xcc) { return ((xcc & 15) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15); };
        for (int b = 0; b < n; ++b) { ca.insert(key(ha[2 * b], ha[2 * b + 1])); cb.insert(key(hb[2 * b], hb[2 * b + 1])); }
        int common = 0;
        for (unsigned c : ca) common += cb.count(c);
        printf("halves: %zu and %zu CUs, %d in common\n", ca.size(), cb.size(), common);
    }
    // a graph captured on a masked stream and launched on it
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(keepA, hipStreamCaptureModeThreadLocal));
        probe<<<n, 256, 0, keepA>>>(d, spin);
        CK(hipStreamEndCapture(keepA, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, keepA)); CK(hipStreamSynchronize(keepA));
        report("GRAPH captured + launched on the [0,128) stream", d, n);
        CK(hipGraphLaunch(ge, keepB)); CK(hipStreamSynchronize(keepB));
        report("same GRAPH launched on the [128,256) stream", d, n);
        CK(hipGraphLaunch(ge, plain)); CK(hipStreamSynchronize(plain));
        report("same GRAPH launched on the plain stream", d, n);
    }
    // do two half-chip streams run side by side?  4 x 1024 workgroups of a long spin: one stream alone, both at once, plain
    unsigned* d2; CK(hipMalloc(&d2, n * 8));
    auto timeit = [&](const char* what, bool a, bool b, bool pl) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 4; ++r) {
            if (a) probe<<<1024, 256, 0, keepA>>>(d, 400000);
            if (b) probe<<<1024, 256, 0, keepB>>>(d2, 400000);
            if (pl) probe<<<1024, 256, 0, plain>>>(d, 400000);
        }
        CK(hipDeviceSynchronize());
        printf("%-44s %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    timeit("warm", true, true, true);
    timeit("half-chip stream A alone (4 launches)", true, false, false);
    timeit("A and B together (4 + 4 launches)", true, true, false);
    timeit("plain stream alone (4 launches)", false, false, true);
    return 0;
}
