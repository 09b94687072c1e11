#include "gfl_profile.hpp"

#include <mutex>
#include <vector>

#include "gfl_common.hpp"

namespace gfl {

static unsigned g_mask = 0;
static std::mutex g_mu;
struct Rec { int stage; hipEvent_t a, b; };
static std::vector<Rec> g_recs;
static hipEvent_t g_open[ST_COUNT];

unsigned profile_mask() { return g_mask; }

void profile_begin(int stage, hipStream_t s) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    g_open[stage] = e;
}

void profile_end(int stage, hipStream_t s) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back({stage, g_open[stage], e});
}

}  // namespace gfl

using namespace gfl;

extern "C" {

int gfl_profile_enable(unsigned stage_mask) {
    g_mask = stage_mask;
    return GFL_OK;
}

int gfl_profile_read(double* total_ms, int* counts, int n_stages) {
    if (!total_ms || !counts || n_stages < ST_COUNT) return GFL_ERR_INVALID;
    for (int i = 0; i < n_stages; ++i) { total_ms[i] = 0.0; counts[i] = 0; }
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            total_ms[r.stage] += ms;
            counts[r.stage] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    return GFL_OK;
}

}  // extern "C"
