// Optional per-stage timing with HIP events recorded ON THE STREAM THE KERNELS RUN ON.
// bench.py enables a stage mask for the timed region and reads the averages afterwards
// (gfl_profile_enable / gfl_profile_read in include/gflow_hip.h).
#pragma once
#include <hip/hip_runtime.h>

namespace gfl {

enum Stage {
    ST_PREPROCESS = 0, ST_COLSCAN, ST_SCATTER, ST_TILE_SORT, ST_BLEND_FWD, ST_LOSS, ST_BLEND_BWD, ST_PRE_BWD_ADAM,
    ST_CAMERA, ST_COUNT
};

unsigned profile_mask();
void profile_begin(int stage, hipStream_t s);
void profile_end(int stage, hipStream_t s);

struct StageScope {
    int stage;
    hipStream_t s;
    bool on;
    StageScope(int st, hipStream_t str) : stage(st), s(str), on((profile_mask() >> st) & 1u) {
        if (on) profile_begin(stage, s);
    }
    ~StageScope() {
        if (on) profile_end(stage, s);
    }
};

}  // namespace gfl
