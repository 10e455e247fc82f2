// FIR transfer functions at arbitrary frequencies, float64: H_k(f) = sum_n b_k[n] exp(-2 pi i f n / fs)
// -- what scipy.signal.freqz evaluates for Filter.get_transfer_function (classes/filter.py:862-900) and,
// filter by filter, for FilterBank.get_transfer_function (classes/filterbank.py:615-655).  gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace freqz {

struct Args {
    const double2* taps;  // [n_filt][n_taps] complex128 (real filters: imaginary parts 0)
    int n_filt, n_taps;
    const double* freqs;  // [n_freq] Hz
    int n_freq;
    double fs;
    double2* out;  // [n_filt][n_freq]
};

// grid = (ceil(n_freq / 256), n_filt): one frequency per thread, Horner in z^-1 from the last tap
// (|z| = 1: the rounding grows like n_taps eps); the taps of a filter are read by all threads together
__global__ __launch_bounds__(256) void k_freqz(Args p) {
    const int i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (i >= p.n_freq) return;
    double s, c;
    sincospi(-2.0 * p.freqs[i] / p.fs, &s, &c);  // z^-1 = exp(-2 pi i f / fs)
    const double2* b = p.taps + (size_t)k * p.n_taps;
    double hr = 0.0, hi = 0.0;
    for (int n = p.n_taps - 1; n >= 0; --n) {
        const double2 t = b[n];
        const double r = hr * c - hi * s + t.x;
        hi = hr * s + hi * c + t.y;
        hr = r;
    }
    p.out[(size_t)k * p.n_freq + i] = make_double2(hr, hi);
}

}  // namespace freqz
