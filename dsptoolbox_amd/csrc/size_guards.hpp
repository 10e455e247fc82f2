// Size predicates shared by device code and the host-side sanitizer run (tests/host_san): plain integer
// arithmetic, no HIP types.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define DS_HD __host__ __device__
#else
#define DS_HD
#endif

// k_welch_finish (kernels_finish.hpp) reads the partial slabs of one chunk -- sx floats of input auto spectra,
// sy complex values of cross spectra (and sy floats of output auto spectra) -- through raw-buffer descriptors whose
// size and per-lane offsets are 32-bit byte counts.  A slab of 4 GiB or more (windows of 2^23 / 2^24 samples with
// 128 / 64 output channels on the four-step path) would wrap; such slabs take the kernel's plain 64-bit loads.
DS_HD inline bool welch_finish_wide_slab(int64_t sx, int64_t sy) {
    const int64_t limit = (int64_t)0xfffffff0;  // the hardware range check compares against a 32-bit size
    return sx * 4 >= limit || sy * 8 >= limit;
}
