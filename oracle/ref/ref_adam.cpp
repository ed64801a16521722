// ref_adam.cpp — runs the reference's OWN selective_adam_update_kernel (threedgrut/optimizers/optimizers.cu:49-77) on the
// host.  The Makefile cuts the kernel section (everything before the torch launch wrapper, which needs nvcc's <<< >>>)
// out of the reference file where it lies into oracle/_ref/sel_adam_kernel.inc (git-ignored, never committed), and this
// file drives it thread by thread.  Test infrastructure only: pins oracle/adam_oracle.py and tests/golden/adam.npz.
#include <math.h>  // brings the float overload of sqrt into the global namespace, as CUDA device code sees it (a bare
                   // <cmath> would resolve `sqrt(float)` to the C double function and compute the step in double)
#include "shim/cuda_shim.h"
#define __global__
#include "shim/cooperative_groups.h"
namespace cooperative_groups { thread_local uint64_t shim_thread_rank = 0; }
#include "../_ref/sel_adam_kernel.inc"
}  // namespace threedgrut  (the cut ends inside the namespace)

extern "C" void ref_selective_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const bool* visibility,
                                   float lr, float b1, float b2, float eps, uint32_t N, uint32_t M) {
    const uint64_t threads = ((uint64_t)N * M + 255) / 256 * 256;  // the launch grid of optimizers.cu:95-96
    for (uint64_t t = 0; t < threads; ++t) {
        cooperative_groups::shim_thread_rank = t;
        threedgrut::selective_adam_update_kernel<float>(param, grad, exp_avg, exp_avg_sq, visibility, lr, b1, b2, eps, N, M);
    }
}
