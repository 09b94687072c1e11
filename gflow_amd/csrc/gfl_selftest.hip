// Device self-tests of wave-level primitives, exported so the GPU test-suite can pin
// them independently of the kernels that use them.
#include "gfl_common.hpp"

namespace gfl {

// one wave: in[64][10] -> out[10] via wave_reduce_scatter10; out2[10] via ten wave_sum calls
__global__ void __launch_bounds__(64) selftest_reduce10_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               float* __restrict__ out2) {
    const int lane = threadIdx.x;
    float v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = in[lane * 10 + k];
    const int comp = reduce_scatter10_component(lane);
    const float mine = wave_reduce_scatter10(v, lane);
    if (comp >= 0) out[comp] = mine;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float s = wave_sum(v[k]);
        if (lane == 0) out2[k] = s;
    }
}

}  // namespace gfl

extern "C" int gfl_selftest_reduce10(const float* in, float* out_scatter, float* out_dpp, gfl_stream_t stream) {
    if (!in || !out_scatter || !out_dpp) return GFL_ERR_INVALID;
    gfl::selftest_reduce10_kernel<<<1, 64, 0, (hipStream_t)stream>>>(in, out_scatter, out_dpp);
    return gfl::check_launch();
}

// sizes of the plain structs of the fused ABI, so FFI bindings can check their mirrors
extern "C" int gfl_abi_sizes(int* sizeof_fit_state, int* sizeof_fit_hyper) {
    if (!sizeof_fit_state || !sizeof_fit_hyper) return GFL_ERR_INVALID;
    *sizeof_fit_state = (int)sizeof(gfl_fit_state);
    *sizeof_fit_hyper = (int)sizeof(gfl_fit_hyper);
    return GFL_OK;
}
