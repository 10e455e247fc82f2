// STFT with any fft_length_samples (reference: _stft, standard/_spectral_methods.py:247-281:
// frames of W samples are windowed, detrended over ALL W samples, then
// `np.fft.rfft(framed, axis=0, n=fft_length_samples)` -- numpy crops the frame to n samples when
// n < W and zero-pads it otherwise; n need not be a power of two).  The power-of-two n >= W
// cases have their own fused kernels; this is the general route for n < W and for every other n:
// windowed (detrended, cropped) frames are materialised as rows, transformed by the whole-signal
// real transform (ds_rfft_dev: LDS / four-step / Bluestein by length), and a small pass applies
// the edge-bin and power scalings on the way into the (bins, frames, channels) result.  gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stftany {

struct PrepArgs {
    const float* x;  // planar (n_ch, ld)
    int64_t n_samples, ld, pad_front;
    int n_ch, W, hop, detrend;
    const float* window;
    int keep;    // samples of a frame the transform sees: min(W, nfft)
    int row0;    // first row of this group; row = frame * n_ch + channel
    float* rows;  // [rows of the group][keep]
};

// one workgroup per row
__global__ __launch_bounds__(256) void k_prepare(PrepArgs p) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int row = p.row0 + blockIdx.x, f = row / p.n_ch, c = row - f * p.n_ch;
    const float* ch = p.x + (int64_t)c * p.ld;
    const int64_t start = (int64_t)f * p.hop - p.pad_front;
    double mean = 0.0;
    if (p.detrend) {  // mean of the whole windowed frame, also when only `keep` samples are kept
        double part = 0.0;
        for (int n = tid; n < p.W; n += 256) {
            const int64_t s = start + n;
            if (s >= 0 && s < p.n_samples) part += (double)(ch[s] * p.window[n]);
        }
        red[tid] = part;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        mean = red[0] / (double)p.W;
    }
    float* out = p.rows + (int64_t)blockIdx.x * p.keep;
    for (int n = tid; n < p.keep; n += 256) {
        const int64_t s = start + n;
        const float v = (s >= 0 && s < p.n_samples) ? ch[s] * p.window[n] : 0.f;
        out[n] = (float)((double)v - mean);
    }
}

struct PostArgs {
    const float2* tmp;  // [B][n_rows] (ds_rfft_dev layout of the group)
    float2* out;        // [B][total_rows]
    int B, n_rows, row0, total_rows;
    float scale, edge;  // amplitude: Z scale (edge bins x edge); power: |Z|^2 scale (edge bins x edge^2)
    int nyq_is_edge, power;
};

__global__ __launch_bounds__(256) void k_post(PostArgs p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)p.B * p.n_rows) return;
    const int b = (int)(i / p.n_rows), r = (int)(i - (int64_t)b * p.n_rows);
    const float2 z = p.tmp[i];
    const bool edge = b == 0 || (p.nyq_is_edge && b == p.B - 1);
    float2 o;
    if (p.power) {
        const float e = edge ? p.scale * p.edge * p.edge : p.scale;
        o = make_float2((z.x * z.x + z.y * z.y) * e, 0.f);
    } else {
        const float e = edge ? p.scale * p.edge : p.scale;
        o = make_float2(z.x * e, z.y * e);
    }
    p.out[(int64_t)b * p.total_rows + p.row0 + r] = o;
}

}  // namespace stftany
