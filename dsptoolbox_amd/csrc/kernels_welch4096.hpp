// Specialised Welch H1/H2/H3 path for the headline shape (nfft 4096, 50 % overlap,
// one input channel).  Placeholder: the generic kernels serve it until the tuned
// kernel lands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace welch4096 {
inline bool enabled() { return false; }
inline int run(hipStream_t, const float*, int64_t, const float*, int, int64_t, int64_t, int,
               const float*, int, int, int, double, double, int, float2*, float*) {
    return -2;
}
}  // namespace welch4096
