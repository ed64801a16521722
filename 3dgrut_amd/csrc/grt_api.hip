// grt_api.hip — 3DGRT entry points (placeholder until the LBVH path lands; every call fails loudly).
#include "common.hpp"

struct GrtHandle {
    GrtConfig cfg;
};

extern "C" {

static int grt_unsupported(const char* what) {
    grut::set_last_error("%s: the 3DGRT software-BVH path is not built into this library yet", what);
    return GRUT_ERR_UNSUPPORTED;
}

int grt_create(const GrtConfig*, GrtHandle**) { return grt_unsupported("grt_create"); }
void grt_destroy(GrtHandle* h) { delete h; }
int grt_build_bvh(GrtHandle*, void*, uint32_t, const float*, const float*, const float*, const float*, int, int) {
    return grt_unsupported("grt_build_bvh");
}
int grt_forward(GrtHandle*, void*, const GrtFrame*, const float*, const float*, const float*, const float*, float*, float*, float*,
                float*, float*, int32_t*) {
    return grt_unsupported("grt_forward");
}
int grt_backward(GrtHandle*, void*, const GrtFrame*, const float*, const float*, const float*, const float*, const float*, const float*,
                 const float*, const float*, const float*, const float*, const float*, const float*, float*, float*) {
    return grt_unsupported("grt_backward");
}
int grt_timings(GrtHandle*, float*, float*, float*) { return grt_unsupported("grt_timings"); }
int grt_stats(GrtHandle*, GrtStats*) { return grt_unsupported("grt_stats"); }
}
