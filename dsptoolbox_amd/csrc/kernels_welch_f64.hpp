// Welch transfer functions in float64 end to end: the precise route for small or ill-conditioned
// problems (reference: _welch, standard/_spectral_methods.py:10-173, and compute_transfer_function,
// transfer_functions/transfer_functions.py:476-534, which are float64 throughout).  gfx950.
//
// Why it exists: with fp32 transforms every frame's rounding floor (1e-7 of the frame's PEAK)
// lands on all of its bins, so bins 80 dB below the peak -- the high end of a fast pink sweep,
// BASELINE config 1 -- are only good to 1e-4 (single-precision pocketfft gives the same).  MI355X
// has a 78 TFLOP/s fp64 vector pipe and such problems are a few hundred transforms: one
// workgroup per (frame, channel) runs a plain radix-2 transform in LDS on double2 values
// (W <= 8192: 128 KB; W = 16384 as the 8192-point complex transform of the even / odd samples plus the
// real-input split), the frame spectra go to HBM as complex128, and a second kernel sums them
// per (bin, channel) in fp64 and applies the same finish() as the fp32 path.
//   inputs: float64 (samples, channels) C-order arrays exactly as the reference holds them
//   (the stride between samples is n_ch), float64 window, mean averaging.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_finish.hpp"

namespace w64 {

using dsk::cd;

struct FrameArgs {
    const double* sig;  // (n_samples, n_ch) C order
    int64_t n_samples;
    int n_ch, W, lgW, hop, n_frames, detrend;
    const double* window;  // [W]
    const double2* tw;     // [W / 2]: exp(-2 pi i k / W)
    double2* spec;         // [n_ch][n_frames][W / 2 + 1]
};

__global__ __launch_bounds__(256) void k_twiddles(double2* tw, int half) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= half) return;
    double s, c;
    sincospi(-(double)k / (double)half, &s, &c);  // exp(-2 pi i k / W), W = 2 half
    tw[k] = make_double2(c, s);
}

// grid = (n_frames, n_ch); dynamic LDS = W * 16 bytes (+ 256 * 8 for the mean)
// PACKED (W = 16384: 256 KB of double2 would not fit): the real frame travels as the W/2-point complex
// sequence z[n] = v[2n] + i v[2n+1] (128 KB), one W/2-point transform, and the split
// X[k] = (Z[k] + conj Z[M-k]) / 2 + W^k (Z[k] - conj Z[M-k]) / (2i), M = W/2, k = 0..M.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_frames(FrameArgs p) {
    extern __shared__ __align__(16) double2 buf[];
    const int tid = threadIdx.x, f = blockIdx.x, c = blockIdx.y;
    const int W = p.W;
    const int M = PACKED ? W / 2 : W, lg = PACKED ? p.lgW - 1 : p.lgW;  // the transform that runs in LDS
    double* red = reinterpret_cast<double*>(buf + M);
    const int64_t start = (int64_t)f * p.hop;
    // windowed frame, zero past the end of the signal (helpers/other.py:207-209)
    double part = 0.0;
    for (int n = tid; n < M; n += 256) {
        double2 z;
        if (PACKED) {
            const int64_t s = start + 2 * n;
            z.x = s < p.n_samples ? p.sig[s * p.n_ch + c] * p.window[2 * n] : 0.0;
            z.y = s + 1 < p.n_samples ? p.sig[(s + 1) * p.n_ch + c] * p.window[2 * n + 1] : 0.0;
        } else {
            const int64_t s = start + n;
            z = make_double2(s < p.n_samples ? p.sig[s * p.n_ch + c] * p.window[n] : 0.0, 0.0);
        }
        part += z.x + z.y;
        buf[__brev((unsigned)n) >> (32 - lg)] = z;  // bit-reversed order in
    }
    if (p.detrend) {  // mean of the WINDOWED frame (_spectral_methods.py:136-139)
        red[tid] = part;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        const double mean = red[0] / (double)W;
        __syncthreads();
        for (int n = tid; n < M; n += 256) {  // every slot holds one sample (PACKED: two)
            buf[n].x -= mean;
            if (PACKED) buf[n].y -= mean;
        }
    }
    __syncthreads();
    // radix-2 decimation in time, natural order out; tw[k] = exp(-2 pi i k / W): the M-point transform of
    // the packed form uses every second entry
    for (int s = 0; s < lg; ++s) {
        const int half = 1 << s;
        for (int i = tid; i < M / 2; i += 256) {
            const int j = i & (half - 1), a = ((i >> s) << (s + 1)) + j, b = a + half;
            const double2 w = p.tw[((size_t)j << (lg - 1 - s)) << (PACKED ? 1 : 0)];
            const double2 u = buf[a], v = buf[b];
            const double2 t = make_double2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            buf[a] = make_double2(u.x + t.x, u.y + t.y);
            buf[b] = make_double2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
    double2* out = p.spec + ((size_t)c * p.n_frames + f) * (W / 2 + 1);
    if (!PACKED) {
        for (int k = tid; k <= W / 2; k += 256) out[k] = buf[k];
        return;
    }
    for (int k = tid; k <= M; k += 256) {
        const double2 zk = buf[k & (M - 1)], zm = buf[(M - k) & (M - 1)];
        const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));   // (Z[k] + conj Z[M-k]) / 2
        const double2 o = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));  // (Z[k] - conj Z[M-k]) / (2i)
        const double2 w = k < M ? p.tw[k] : make_double2(-1.0, 0.0);
        out[k] = make_double2(e.x + o.x * w.x - o.y * w.y, e.y + o.x * w.y + o.y * w.x);
    }
}

struct TfArgs {
    const double2* xs;  // [n_cx][F][nb]
    const double2* ys;  // [n_cy][F][nb]
    int n_cx, n_cy, n_frames, mode;
    dsk::FinishPar fin;
    double2* tf;  // [nb][n_cy]
    double* coh;  // [nb][n_cy]
};

// grid = (ceil(nb / 256), n_cy)
__global__ __launch_bounds__(256) void k_tf(TfArgs p) {
    const int b = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, nb = p.fin.nb;
    if (b >= nb) return;
    const int cx = p.n_cx == 1 ? 0 : c;
    const double2* X = p.xs + (size_t)cx * p.n_frames * nb + b;
    const double2* Y = p.ys + (size_t)c * p.n_frames * nb + b;
    double sxx = 0.0, syy = 0.0;
    cd sxy{0.0, 0.0};
    for (int f = 0; f < p.n_frames; ++f) {
        const double2 x = X[(size_t)f * nb], y = Y[(size_t)f * nb];
        sxx += x.x * x.x + x.y * x.y;
        syy += y.x * y.x + y.y * y.y;
        sxy.x += x.x * y.x + x.y * y.y;  // conj(x) y
        sxy.y += x.x * y.y - x.y * y.x;
    }
    const cd gxy = dsk::finish_cplx(sxy, b, p.fin);
    const double gxx = dsk::finish_real(sxx, b, p.fin), gyy = dsk::finish_real(syy, b, p.fin);
    const double axy2 = gxy.x * gxy.x + gxy.y * gxy.y;
    cd h;
    if (p.mode == 1) {
        h = cd{gxy.x / gxx, gxy.y / gxx};
    } else if (p.mode == 2) {  // see tf_from_sums (kernels_finish.hpp) for the real-negative case
        const cd gyx = (sxy.y == 0.0) ? gxy : cd{gxy.x, -gxy.y};
        h = cd{gyy * gyx.x / axy2, -gyy * gyx.y / axy2};
    } else {
        const double s = sqrt(gyy / gxx) / sqrt(axy2);
        h = cd{gxy.x * s, gxy.y * s};
    }
    p.tf[(size_t)b * p.n_cy + c] = make_double2(h.x, h.y);
    p.coh[(size_t)b * p.n_cy + c] = axy2 / gxx / gyy;
}

// average = "median" (_spectral_methods.py:153-162): per bin the median over the frames of |X|^2,
// |Y|^2 and of the real and imaginary parts of conj(X) Y; the host folds the bias n (F or F - 1)
// into fin.inv.  One workgroup per (bin, channel); the four series sit in LDS (4 F doubles) and
// every element is ranked against all others (ties by index).  grid = (nb, n_cy).
__device__ __forceinline__ double median_rank(const double* s, int F, int tid, double* out2) {
    const int r0 = (F - 1) / 2, r1 = F / 2;
    for (int i = tid; i < F; i += 256) {
        const double v = s[i];
        int rank = 0;
        for (int j = 0; j < F; ++j) {
            const double u = s[j];
            rank += (u < v || (u == v && j < i)) ? 1 : 0;
        }
        if (rank == r0) out2[0] = v;
        if (rank == r1) out2[1] = v;
    }
    return 0.0;
}
__global__ __launch_bounds__(256) void k_tf_median(TfArgs p) {
    extern __shared__ __align__(16) double ser[];  // [4][F] + 8 results
    const int b = blockIdx.x, c = blockIdx.y, nb = p.fin.nb, F = p.n_frames, tid = threadIdx.x;
    double* res = ser + 4 * (size_t)F;
    const int cx = p.n_cx == 1 ? 0 : c;
    const double2* X = p.xs + (size_t)cx * F * nb + b;
    const double2* Y = p.ys + (size_t)c * F * nb + b;
    for (int f = tid; f < F; f += 256) {
        const double2 x = X[(size_t)f * nb], y = Y[(size_t)f * nb];
        ser[f] = x.x * x.x + x.y * x.y;
        ser[F + f] = y.x * y.x + y.y * y.y;
        ser[2 * F + f] = x.x * y.x + x.y * y.y;
        ser[3 * F + f] = x.x * y.y - x.y * y.x;
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q) median_rank(ser + (size_t)q * F, F, tid, res + 2 * q);
    __syncthreads();
    if (tid != 0) return;
    const double sxx = 0.5 * (res[0] + res[1]), syy = 0.5 * (res[2] + res[3]);
    // + 0.0: the reference forms `median(real) + 1j * median(imag)`, so a -0 imaginary median becomes +0
    // (it decides the branch of the principal square root at the purely real bins)
    const cd sxy{0.5 * (res[4] + res[5]), 0.5 * (res[6] + res[7]) + 0.0};
    const cd gxy = dsk::finish_cplx(sxy, b, p.fin);
    const double gxx = dsk::finish_real(sxx, b, p.fin), gyy = dsk::finish_real(syy, b, p.fin);
    const double axy2 = gxy.x * gxy.x + gxy.y * gxy.y;
    cd h;
    if (p.mode == 1) {
        h = cd{gxy.x / gxx, gxy.y / gxx};
    } else if (p.mode == 2) {
        const cd gyx = (sxy.y == 0.0) ? gxy : cd{gxy.x, -gxy.y};
        h = cd{gyy * gyx.x / axy2, -gyy * gyx.y / axy2};
    } else {
        const double s = sqrt(gyy / gxx) / sqrt(axy2);
        h = cd{gxy.x * s, gxy.y * s};
    }
    p.tf[(size_t)b * p.n_cy + c] = make_double2(h.x, h.y);
    p.coh[(size_t)b * p.n_cy + c] = axy2 / gxx / gyy;
}

}  // namespace w64
