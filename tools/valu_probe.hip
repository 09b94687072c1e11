// Issue-rate probe for gfx950 (analysis tool, not part of the library): how many cycles does a
// SIMD spend per wave64 VALU instruction of each kind, as a function of the waves resident on it?
// The blend kernels are instruction-issue bound; this fixes the price list they are designed
// against.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_probe.hip -o tools/valu_probe.bin
//   gpurun -- ./tools/valu_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REP16(x) x x x x x x x x x x x x x x x x

enum Kind { FMA_DEP, FMA_IND, FMA_SGPR, PK_FMA, EXP, RCP, CNDMASK, DPP_ADD, PERMSWAP, LDS_BCAST128, LDS_B64, SMEM16, MIXED };

template <int KIND>
__global__ void __launch_bounds__(256) probe(float* __restrict__ out, long long* __restrict__ cyc, const float* __restrict__ tbl, int iters, int stride) {
    __shared__ float4 lds[1024];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const float sg = tbl[blockIdx.x & 7];          // wave-uniform -> SGPR
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned soff = (blockIdx.x * 4 + wave) * 64;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == FMA_DEP) {
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));)
        } else if (KIND == FMA_IND) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == FMA_SGPR) {
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sg), "v"(c));
        } else if (KIND == PK_FMA) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b}, cc = {c, c};
            asm volatile(
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (KIND == EXP) {
            asm volatile(
                "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == RCP) {
            asm volatile(
                "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == CNDMASK) {
            asm volatile(
                "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
        } else if (KIND == DPP_ADD) {
            asm volatile(
                "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == PERMSWAP) {
            asm volatile(
                "s_nop 1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                "s_nop 1\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                "s_nop 1\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                "s_nop 1\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == LDS_BCAST128) {
            // 16 wave-uniform 16-byte reads (what the round-1 blend loops do three times per splat)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 v = lds[(i * 16 + k * stride) & 1023];
                a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
            }
        } else if (KIND == LDS_B64) {
            // per-lane 8-byte reads, conflict-free (lane-consecutive)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float2 v = reinterpret_cast<const float2*>(lds)[(threadIdx.x & 63) + ((i + k) & 15) * 64];
                a0 += v.x; a1 += v.y;
            }
        } else if (KIND == SMEM16) {
            // 16 x 64-byte scalar loads per trip at wave-uniform addresses walking a table with `stride` rows
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* p = tbl + (size_t)((soff + (i * 4 + k) * stride) & 0xffff) * 16;
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = p[q];       // uniform address + readonly -> s_load_dwordx16
#pragma unroll
                for (int q = 0; q < 16; q += 4) { a0 = fmaf(a0, v[q], v[q + 1]); a1 = fmaf(a1, v[q + 2], v[q + 3]); }
            }
        } else if (KIND == MIXED) {
            // the shape of one forward unit: ~20 dependent-ish VALU incl. one exp
            float dx = a0 - sg, dy = a1 - sg;
            float q = fmaf(b * dx, dx, (c * dy) * dy);
            float pw = fmaf(-0.5f, q, -((b * dx) * dy));
            float G = __expf(fminf(pw, 0.f));
            float al = fminf(0.99f, c * G);
            bool val = (pw <= 0.f) & (al >= 1.f / 255.f);
            float a = val ? al : 0.f;
            float tt = a2 * (1.f - a);
            bool stop = tt < 1e-4f;
            float w = stop ? 0.f : a * a2;
            a3 = fmaf(b, w, a3); a4 = fmaf(c, w, a4); a5 = fmaf(b, w, a5); a6 = fmaf(c, w, a6);
            a2 = stop ? 0.f : tt;
            a0 += 1e-3f; a1 -= 1e-3f;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter, int wg_per_cu, float* out, long long* cyc, const float* tbl, int iters,
                int stride = 1) {
    const int grid = 256 * wg_per_cu;
    // LDS padding limits residency: 160 KB / wg_per_cu per workgroup (static 16 KB + dynamic)
    const int dyn = wg_per_cu >= 8 ? 0 : (160 * 1024 / wg_per_cu - 16 * 1024 - 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<grid, 256, dyn>>>(out, cyc, tbl, 8, stride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND><<<grid, 256, dyn>>>(out, cyc, tbl, iters, stride);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long* h = (long long*)malloc(grid * 8);
    hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < grid; ++i) mean += h[i];
    mean /= grid;
    free(h);
    // waves per SIMD = wg_per_cu (4 waves per workgroup, one per SIMD)
    const double inst_per_wave = (double)iters * per_iter;
    // s_memtime ticks at 100 MHz; derive shader cycles from the wall time at 2.4 GHz as well
    printf("%-14s waves/SIMD %d  stride %3d: %8.3f us  -> %6.2f cyc@2.4GHz per wave-instr per SIMD (memtime ticks/instr/SIMD %.3f)\n",
           name, wg_per_cu, stride, ms * 1e3, ms * 1e-3 * 2.4e9 / (inst_per_wave * wg_per_cu), mean / (inst_per_wave * wg_per_cu));
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    float* out; long long* cyc; float* tbl;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&cyc, 256 * 8 * 8);
    hipMalloc(&tbl, 65536 * 64);
    hipMemset(tbl, 0, 65536 * 64);
    const int it = 4000;
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) { run<FMA_DEP>("fma dep", 16, 1, out, cyc, tbl, it); run<FMA_IND>("fma indep", 16, 1, out, cyc, tbl, it); run<FMA_SGPR>("fma sgpr", 16, 1, out, cyc, tbl, it); run<PK_FMA>("pk_fma", 16, 1, out, cyc, tbl, it); run<EXP>("exp", 16, 1, out, cyc, tbl, it); run<RCP>("rcp", 16, 1, out, cyc, tbl, it); run<CNDMASK>("cmp+cndmask", 16, 1, out, cyc, tbl, it); run<DPP_ADD>("add_dpp", 16, 1, out, cyc, tbl, it); run<PERMSWAP>("permlane swap", 16, 1, out, cyc, tbl, it); run<LDS_BCAST128>("lds b128 bcast", 16, 1, out, cyc, tbl, it); run<LDS_B64>("lds b64 lane", 16, 1, out, cyc, tbl, it); run<SMEM16>("smem x16", 4, 1, out, cyc, tbl, it, 1); run<SMEM16>("smem x16", 4, 1, out, cyc, tbl, it, 37); run<MIXED>("mixed unit", 24, 1, out, cyc, tbl, it); }
        if (w == 2) { run<FMA_DEP>("fma dep", 16, 2, out, cyc, tbl, it); run<FMA_IND>("fma indep", 16, 2, out, cyc, tbl, it); run<PK_FMA>("pk_fma", 16, 2, out, cyc, tbl, it); run<EXP>("exp", 16, 2, out, cyc, tbl, it); run<MIXED>("mixed unit", 24, 2, out, cyc, tbl, it); run<SMEM16>("smem x16", 4, 2, out, cyc, tbl, it, 37); }
        if (w == 4) { run<FMA_DEP>("fma dep", 16, 4, out, cyc, tbl, it); run<FMA_IND>("fma indep", 16, 4, out, cyc, tbl, it); run<PK_FMA>("pk_fma", 16, 4, out, cyc, tbl, it); run<EXP>("exp", 16, 4, out, cyc, tbl, it); run<CNDMASK>("cmp+cndmask", 16, 4, out, cyc, tbl, it); run<DPP_ADD>("add_dpp", 16, 4, out, cyc, tbl, it); run<PERMSWAP>("permlane swap", 16, 4, out, cyc, tbl, it); run<LDS_BCAST128>("lds b128 bcast", 16, 4, out, cyc, tbl, it); run<LDS_B64>("lds b64 lane", 16, 4, out, cyc, tbl, it); run<SMEM16>("smem x16", 4, 4, out, cyc, tbl, it, 37); run<MIXED>("mixed unit", 24, 4, out, cyc, tbl, it); }
        if (w == 8) { run<FMA_DEP>("fma dep", 16, 8, out, cyc, tbl, it); run<FMA_IND>("fma indep", 16, 8, out, cyc, tbl, it); run<FMA_SGPR>("fma sgpr", 16, 8, out, cyc, tbl, it); run<PK_FMA>("pk_fma", 16, 8, out, cyc, tbl, it); run<EXP>("exp", 16, 8, out, cyc, tbl, it); run<RCP>("rcp", 16, 8, out, cyc, tbl, it); run<CNDMASK>("cmp+cndmask", 16, 8, out, cyc, tbl, it); run<DPP_ADD>("add_dpp", 16, 8, out, cyc, tbl, it); run<PERMSWAP>("permlane swap", 16, 8, out, cyc, tbl, it); run<LDS_BCAST128>("lds b128 bcast", 16, 8, out, cyc, tbl, it); run<LDS_B64>("lds b64 lane", 16, 8, out, cyc, tbl, it); run<SMEM16>("smem x16", 4, 8, out, cyc, tbl, it, 1); run<SMEM16>("smem x16", 4, 8, out, cyc, tbl, it, 37); run<MIXED>("mixed unit", 24, 8, out, cyc, tbl, it); }
    }
    return 0;
}
