// Block-streaming FIR classes with their state on the device (reference:
// dsptoolbox/classes/fir_filter_realtime.py:75-335).  The reference's algorithms are executed
// literally -- time-domain input buffers, the frequency-domain delay line of partition spectra
// with its ONE running index, an inverse transform without an explicit length -- so that the
// blocks equal the reference's also where those are not the causal convolution.  The transforms
// are the library's own (ds_rfft_dev / ds_deconv_dev: LDS, four-step or Bluestein by length);
// this file holds the small kernels between them.  gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace firstream {

// partitioned classes (:214-221, :310-315): inbuf[c][0:bs] = inbuf[c][bs:2bs]; inbuf[c][bs:2bs] = block
// inbuf rows of the channels [ch0, ch0 + n_call); block [n_call][bs]
__global__ __launch_bounds__(256) void k_shift_in(float* inbuf, const float* block, int bs, int ch0, int n_call) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n_call * bs) return;
    const int c = (int)(i / bs), n = (int)(i - (int64_t)c * bs);
    float* row = inbuf + (int64_t)(ch0 + c) * 2 * bs;
    row[n] = row[bs + n];  // element-wise: slot n is read before any thread writes it
    row[bs + n] = block[i];
}

// :224-233 / :318-323: S[:, ind, ch] = X;  Y = sum_p H[:, p, (ch)] * S[:, ind - p (mod P), ch]
// X: [B][n_call] (the forward transform's (bins, channels) layout); Y: [n_call][B] (what the
// inverse transform takes per channel); S: [B][P][C]; H: [B][P][Cf]
struct AccArgs {
    const float2* X;
    float2* S;
    const float2* H;
    float2* Y;
    int B, P, C, Cf, ch0, n_call, ind;
};
__global__ __launch_bounds__(256) void k_part_acc(AccArgs p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)p.B * p.n_call) return;
    const int b = (int)(i / p.n_call), cc = (int)(i - (int64_t)b * p.n_call), ch = p.ch0 + cc;
    const int hf = p.Cf == 1 ? 0 : ch;
    float2* Sb = p.S + (int64_t)b * p.P * p.C;
    const float2* Hb = p.H + (int64_t)b * p.P * p.Cf;
    const float2 x = p.X[i];
    Sb[(int64_t)p.ind * p.C + ch] = x;
    double yr = 0.0, yi = 0.0;
    for (int q = 0; q < p.P; ++q) {
        int slot = p.ind - q;
        if (slot < 0) slot += p.P;
        const float2 s = q == 0 ? x : Sb[(int64_t)slot * p.C + ch];
        const float2 h = Hb[(int64_t)q * p.Cf + hf];
        yr += (double)h.x * s.x - (double)h.y * s.y;
        yi += (double)h.x * s.y + (double)h.y * s.x;
    }
    p.Y[(int64_t)cc * p.B + b] = make_float2((float)yr, (float)yi);  // [channel][bin]: ds_deconv_dev's per-channel layout
}

// last bs samples of each row of full [n_call][len] -> out [n_call][bs]
__global__ __launch_bounds__(256) void k_tail(const float* full, int64_t len, int bs, int n_call, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n_call * bs) return;
    const int c = (int)(i / bs), n = (int)(i - (int64_t)c * bs);
    out[i] = full[(int64_t)c * len + (len - bs) + n];
}

// overlap-save class (:135-142): buffer[-bs:] = block before the transform ...
__global__ __launch_bounds__(256) void k_ols_put(float* row, const float* block, int64_t L, int bs) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < bs) row[L - bs + n] = block[n];
}
// ... and buffer[:-bs] = buffer[bs:] after it (through a scratch row: the ranges overlap)
__global__ __launch_bounds__(256) void k_ols_roll(const float* src, float* dst, int64_t L, int bs) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n < L - bs) dst[n] = src[bs + n];
    else if (n < L) dst[n] = src[n];
}

// Y[b] = X[b] * H[b] for b < nb (the product the inverse transform sees)
__global__ __launch_bounds__(256) void k_cmul(const float2* X, const float2* H, float2* Y, int nb) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    const float2 x = X[b], h = H[b];
    Y[b] = make_float2(x.x * h.x - x.y * h.y, x.x * h.y + x.y * h.x);
}

}  // namespace firstream
