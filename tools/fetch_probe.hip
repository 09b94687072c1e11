// What does rocprofv3's FETCH_SIZE count for the access patterns of the blend kernels?  (analysis tool, not part of the library)
// MI355X_MICROARCH.md calibrates "FETCH_SIZE x 2" on wide coalesced streaming reads (gfx950 counts a 128-byte request as 64);
// the blend kernels GATHER 48-byte records (three 16-byte loads per lane) through a coalesced id list.  Each kernel below moves a
// KNOWN number of bytes; run under `rocprofv3 --pmc FETCH_SIZE` (own pass) and compare (tools/fetch_probe.sh):
//   stream       : every lane reads consecutive float4s          -- N * 16 bytes, each byte once
//   gather_small : 48-byte rows of a 2.9 MB table (60 000 rows: the fit's `rec`) through a random id list of K entries --
//                  the table fits every XCD's 4 MB L2: HBM traffic >= ids (4 K) + table once per XCD (8 x 2.9 MB at most)
//   gather_huge  : the same through a 1.5 GB table (32 M rows): every row is a miss -- 4 K + K x (the 64-byte or 128-byte
//                  lines a 48-byte row at a 48-byte stride touches: 1.375 x 128 B or 1.75 x 64 B on average)
//   hipcc --offload-arch=gfx950 -O2 tools/fetch_probe.hip -o tools/fetch_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) stream_kernel(const float4* __restrict__ in, float* __restrict__ out, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void __launch_bounds__(256) gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                     float* __restrict__ out, int K) {
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) {
        const float4* r4 = reinterpret_cast<const float4*>(table + (size_t)ids[i] * 12);
        const float4 a = r4[0], b = r4[1], c = r4[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// the same name for the profiler's per-kernel table would fold the two gathers: a second symbol
__global__ void __launch_bounds__(256) gather_huge_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                                          float* __restrict__ out, int K) {
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) {
        const float4* r4 = reinterpret_cast<const float4*>(table + (size_t)ids[i] * 12);
        const float4 a = r4[0], b = r4[1], c = r4[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t stream_bytes = (size_t)256 << 20;
    const int K = 4 << 20, rows_small = 60000;
    const size_t rows_huge = (size_t)32 << 20;
    float *d_stream, *d_small, *d_huge, *d_out;
    int32_t *d_ids_small, *d_ids_huge;
    CK(hipMalloc(&d_stream, stream_bytes));
    CK(hipMalloc(&d_small, (size_t)rows_small * 48));
    CK(hipMalloc(&d_huge, rows_huge * 48));
    CK(hipMalloc(&d_out, 256));
    CK(hipMalloc(&d_ids_small, (size_t)K * 4));
    CK(hipMalloc(&d_ids_huge, (size_t)K * 4));
    CK(hipMemset(d_stream, 0, stream_bytes));
    CK(hipMemset(d_small, 0, (size_t)rows_small * 48));
    CK(hipMemset(d_huge, 0, rows_huge * 48));
    std::vector<int32_t> h(K);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < K; ++i) h[i] = (int32_t)(rnd() % rows_small);
    CK(hipMemcpy(d_ids_small, h.data(), (size_t)K * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < K; ++i) h[i] = (int32_t)(rnd() % rows_huge);
    CK(hipMemcpy(d_ids_huge, h.data(), (size_t)K * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        stream_kernel<<<2048, 256>>>((const float4*)d_stream, d_out, stream_bytes / 16);
        gather_kernel<<<2048, 256>>>(d_small, d_ids_small, d_out, K);
        gather_huge_kernel<<<2048, 256>>>(d_huge, d_ids_huge, d_out, K);
        CK(hipDeviceSynchronize());
    }
    printf("known bytes per launch: stream %zu; gather_small ids %zu + table %zu (x <= 8 XCDs); gather_huge ids %zu + rows %zu "
           "(48 B each; %.0f if whole 64-B lines, %.0f if whole 128-B lines)\n", stream_bytes, (size_t)K * 4, (size_t)rows_small * 48,
           (size_t)K * 4, (size_t)K * 48, K * 1.75 * 64.0, K * 1.375 * 128.0);
    return 0;
}
