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

// Direct causal convolution in float64 for SMALL problems: y[f][c][n] = sum_k taps[f][k] x[c][n - k] (_lfilter_fir,
// classes/filter_helpers.py:454-503).  An FFT convolution rounds at 1e-7 of (peak of the block) x (size of the taps); when
// the output is far below that -- a signal much shorter than the filter whose leading taps are near zero -- its RELATIVE
// error is 1e-5 and more (limit (x)); the direct sum has no such floor.  grid = (ceil(n / 256), n_ch, n_filt)
struct DirectArgs {
    const float* x;     // [n_ch][ldx]
    const float* taps;  // [n_filt][n_taps]
    int64_t n_samples, ldx, ld_y;
    int n_ch, n_taps;
    float* y;  // [(f n_ch + c) ld_y + n]
};
__global__ __launch_bounds__(256) void k_fir_direct(DirectArgs p) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= p.n_samples) return;
    const int c = blockIdx.y, f = blockIdx.z;
    const float* __restrict__ x = p.x + (int64_t)c * p.ldx;
    const float* __restrict__ h = p.taps + (int64_t)f * p.n_taps;
    const int kmax = (int)(n < (int64_t)(p.n_taps - 1) ? n : (int64_t)(p.n_taps - 1));
    double acc = 0.0;
    for (int k = 0; k <= kmax; ++k) acc += (double)h[k] * (double)x[n - k];
    p.y[((int64_t)f * p.n_ch + c) * p.ld_y + n] = (float)acc;
}

}  // namespace freqz
