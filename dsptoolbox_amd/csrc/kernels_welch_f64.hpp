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
// real-input split; W = 2^15 ... 2^18: one decimation-in-frequency stage in front of it, k_frames_cls), the frame spectra go to HBM as complex128, and a second kernel sums them
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
    // sample s of channel c at sig[s * s_stride + c * c_stride]: (n_ch, 1) for the reference's layout, (1, n_samples) for
    // the planar copy k_planar makes when there are enough channels for the strided reads to hurt
    int64_t s_stride = 0, c_stride = 1;
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
    const double* __restrict__ chan = p.sig + (int64_t)c * p.c_stride;
    // windowed frame, zero past the end of the signal (helpers/other.py:207-209)
    double part = 0.0;
    for (int n = tid; n < M; n += 256) {
        double2 z;
        if (PACKED) {
            const int64_t s = start + 2 * n;
            z.x = s < p.n_samples ? chan[s * p.s_stride] * p.window[2 * n] : 0.0;
            z.y = s + 1 < p.n_samples ? chan[(s + 1) * p.s_stride] * p.window[2 * n + 1] : 0.0;
        } else {
            const int64_t s = start + n;
            z = make_double2(s < p.n_samples ? chan[s * p.s_stride] * p.window[n] : 0.0, 0.0);
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

// ---- windows of 2^15 ... 2^18 samples --------------------------------------------------------------------------
// The packed sequence z[n] = v[2n] + i v[2n+1] has Wh = W/2 = RC x 8192 points (RC = 2 ... 16): one radix-RC
// decimation-in-frequency stage in front of the 8192-point LDS transform,
//     b_r[m] = ( sum_{s < RC} z[m + 8192 s] W_RC^(r s) ) W_Wh^(r m) ,   Z[RC k' + r] = FFT8192(b_r)[k'] ,
// one workgroup per (frame, class r, channel) -- every class reads the whole frame (L2; this is the route of SHORT
// estimates) -- into zc[channel][frame][r][k'], then k_split forms X[k] = (Z[k] + conj Z[Wh-k]) / 2 + W^k (...) / (2i)
// as k_frames<true> does.  Removing the mean of the windowed frame only changes class 0 (sum_s W_RC^(r s) = 0 otherwise):
// b_0[m] -= RC mean (1 + i).
constexpr int LONG_M = 8192, LONG_LG = 13;

// exp(-2 pi i k / W) from the half table tw[0 .. W/2)
__device__ __forceinline__ double2 tw_full(const double2* __restrict__ tw, unsigned k, int W) {
    k &= (unsigned)(W - 1);
    if (k < (unsigned)(W / 2)) return tw[k];
    const double2 t = tw[k - (unsigned)(W / 2)];
    return make_double2(-t.x, -t.y);
}

struct LongArgs {
    FrameArgs f;
    int rc, lg_rc;  // classes: W / 2 = rc * 8192
    double2* zc;    // [n_ch][n_frames][rc][8192]
    const double* planar;  // [n_ch][n_samples]: the signal channel by channel (k_planar)
};

// (samples, channels) -> [channel][samples] through a 32 x 32 tile: every class workgroup of a long window reads its
// channel's frame rc times, and from the reference's channel-fastest layout each of those reads is one 8-byte value per
// 128-byte line (64 + 1 channels x 2^20 samples: k_frames_cls 3.7 / 6.1 / 19.4 ms at 2^15 / 2^16 / 2^18-sample windows from the
// host layout, 2.7 / 3.8 / 9.5 ms from the planar copy).
// grid = (ceil(n_samples / 32), ceil(n_ch / 32)), 256 threads
__global__ __launch_bounds__(256) void k_planar(const double* __restrict__ sig, int64_t n_samples, int n_ch, double* __restrict__ out) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t s0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t sI = s0 + ty + 8 * j;
        const int c = c0 + tx;
        tile[ty + 8 * j][tx] = (sI < n_samples && c < n_ch) ? sig[sI * n_ch + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j;
        const int64_t sI = s0 + tx;
        if (c < n_ch && sI < n_samples) out[(int64_t)c * n_samples + sI] = tile[tx][ty + 8 * j];
    }
}

// grid = (n_frames * rc, n_ch); dynamic LDS = 8192 * 16 + 256 * 8 bytes
__global__ __launch_bounds__(256) void k_frames_cls(LongArgs q) {
    extern __shared__ __align__(16) double2 buf[];
    const FrameArgs& p = q.f;
    constexpr int M = LONG_M, lg = LONG_LG;
    const int tid = threadIdx.x, f = (int)blockIdx.x >> q.lg_rc, r = (int)blockIdx.x & (q.rc - 1), c = blockIdx.y;
    const int W = p.W, Wh = W / 2;
    double* red = reinterpret_cast<double*>(buf + M);
    const int64_t start = (int64_t)f * p.hop;
    const unsigned step_rs = (unsigned)(W >> q.lg_rc);  // W_RC^1 = exp(-2 pi i (W / RC) / W)
    const double* __restrict__ chan = q.planar + (int64_t)c * p.n_samples;
    double part = 0.0;
    for (int m = tid; m < M; m += 256) {
        double2 acc = make_double2(0.0, 0.0);
        for (int s = 0; s < q.rc; ++s) {
            const int n = m + M * s;
            const int64_t i = start + 2 * n;
            const double a = i < p.n_samples ? chan[i] * p.window[2 * n] : 0.0;
            const double b = i + 1 < p.n_samples ? chan[i + 1] * p.window[2 * n + 1] : 0.0;
            part += a + b;
            const double2 w = tw_full(p.tw, (unsigned)(r * s) * step_rs, W);
            acc.x += a * w.x - b * w.y;
            acc.y += a * w.y + b * w.x;
        }
        buf[m] = acc;
    }
    if (p.detrend && r == 0) {  // (block-uniform branch)
        red[tid] = part;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        const double sub = red[0] / (double)W * (double)q.rc;
        for (int m = tid; m < M; m += 256) {  // (a thread's own slots)
            buf[m].x -= sub;
            buf[m].y -= sub;
        }
    }
    __syncthreads();
    // twiddle W_Wh^(r m) = exp(-2 pi i (2 r m) / W), then into bit-reversed order (through registers: 32 per thread)
    double2 v[M / 256];
#pragma unroll
    for (int j = 0; j < M / 256; ++j) {
        const int m = tid + 256 * j;
        const double2 b = buf[m], w = tw_full(p.tw, 2u * (unsigned)r * (unsigned)m, W);
        v[j] = make_double2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < M / 256; ++j) buf[__brev((unsigned)(tid + 256 * j)) >> (32 - lg)] = v[j];
    __syncthreads();
    const int tstep = W >> lg;  // exp(-2 pi i k / 8192) = tw[k * W / 8192]
    for (int s = 0; s < lg; ++s) {
        const int half = 1 << s;
        for (int i = tid; i < M / 2; i += 256) {
            const int j = i & (half - 1), a = ((i >> s) << (s + 1)) + j, b = a + half;
            const double2 w = p.tw[(size_t)(j << (lg - 1 - s)) * tstep];
            const double2 u = buf[a], t0 = buf[b];
            const double2 t = make_double2(t0.x * w.x - t0.y * w.y, t0.x * w.y + t0.y * w.x);
            buf[a] = make_double2(u.x + t.x, u.y + t.y);
            buf[b] = make_double2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
    double2* out = q.zc + ((((size_t)c * p.n_frames + f) << q.lg_rc) + r) * M;
    for (int k = tid; k < M; k += 256) out[k] = buf[k];
}

// grid = (ceil((W/2 + 1) / 256), n_frames, n_ch)
__global__ __launch_bounds__(256) void k_split(LongArgs q) {
    const FrameArgs& p = q.f;
    const int Wh = p.W / 2, k = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y, c = blockIdx.z;
    if (k > Wh) return;
    const double2* z = q.zc + (((size_t)c * p.n_frames + f) << q.lg_rc) * LONG_M;
    auto Z = [&](int n) {
        n &= Wh - 1;
        return z[(size_t)(n & (q.rc - 1)) * LONG_M + (n >> q.lg_rc)];
    };
    const double2 zk = Z(k), zm = Z(Wh - k);
    const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
    const double2 o = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
    const double2 w = k < Wh ? p.tw[k] : make_double2(-1.0, 0.0);
    p.spec[((size_t)c * p.n_frames + f) * (Wh + 1) + k] = make_double2(e.x + o.x * w.x - o.y * w.y, e.y + o.x * w.y + o.y * w.x);
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

// ---- auto / cross spectra and the cross-spectral matrix from the same float64 frame spectra ----------
// (_welch itself, _spectral_methods.py:141-171, and _csm_welch, :351-369: short estimates -- a handful of
// frames -- have no averaging to bring the fp32 transform rounding down, see backend._x64_short)
struct SpecArgs {
    const double2* xs;  // [n_ch][F][nb]
    const double2* ys;  // [n_ch][F][nb] or nullptr: auto spectra of xs
    int n_ch, n_frames;
    dsk::FinishPar fin;
    double2* out;  // [nb][n_ch]: auto spectra in .x (imaginary part 0), cross spectra conj(X) Y
};

// grid = (ceil(nb / 256), n_ch)
__global__ __launch_bounds__(256) void k_spec(SpecArgs p) {
    const int b = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, nb = p.fin.nb;
    if (b >= nb) return;
    const double2* X = p.xs + (size_t)c * p.n_frames * nb + b;
    if (!p.ys) {
        double s = 0.0;
        for (int f = 0; f < p.n_frames; ++f) {
            const double2 x = X[(size_t)f * nb];
            s += x.x * x.x + x.y * x.y;
        }
        p.out[(size_t)b * p.n_ch + c] = make_double2(dsk::finish_real(s, b, p.fin), 0.0);
        return;
    }
    const double2* Y = p.ys + (size_t)c * p.n_frames * nb + b;
    cd s{0.0, 0.0};
    for (int f = 0; f < p.n_frames; ++f) {
        const double2 x = X[(size_t)f * nb], y = Y[(size_t)f * nb];
        s.x += x.x * y.x + x.y * y.y;  // conj(x) y
        s.y += x.x * y.y - x.y * y.x;
    }
    s.y += 0.0;  // a sum of -0 terms becomes +0 like the reference's mean
    const cd g = dsk::finish_cplx(s, b, p.fin);
    p.out[(size_t)b * p.n_ch + c] = make_double2(g.x, g.y);
}

// average = "median": grid = (nb, n_ch), LDS = (2 F + 4) doubles; the host folds the bias into fin.inv
__global__ __launch_bounds__(256) void k_spec_median(SpecArgs p) {
    extern __shared__ __align__(16) double ser[];
    const int b = blockIdx.x, c = blockIdx.y, nb = p.fin.nb, F = p.n_frames, tid = threadIdx.x;
    double* res = ser + 2 * (size_t)F;
    const double2* X = p.xs + (size_t)c * F * nb + b;
    const double2* Y = p.ys ? p.ys + (size_t)c * F * nb + b : nullptr;
    for (int f = tid; f < F; f += 256) {
        const double2 x = X[(size_t)f * nb];
        if (Y) {
            const double2 y = Y[(size_t)f * nb];
            ser[f] = x.x * y.x + x.y * y.y;
            ser[F + f] = x.x * y.y - x.y * y.x;
        } else {
            ser[f] = x.x * x.x + x.y * x.y;
        }
    }
    __syncthreads();
    median_rank(ser, F, tid, res);
    if (Y) median_rank(ser + F, F, tid, res + 2);
    __syncthreads();
    if (tid != 0) return;
    // the reference forms `median(real) + 1j * median(imag)` and finishes that complex number, auto spectra included
    const cd s{0.5 * (res[0] + res[1]), Y ? 0.5 * (res[2] + res[3]) + 0.0 : 0.0};
    const cd g = dsk::finish_cplx(s, b, p.fin);
    p.out[(size_t)b * p.n_ch + c] = make_double2(g.x, g.y);
}

struct CsmArgs {
    const double2* xs;  // [n_ch][F][nb]
    int n_ch, n_frames;
    dsk::FinishPar fin;
    double2* csm;  // [nb][n_ch][n_ch]
};
// a workgroup takes 9 * 256 = 2304 channel pairs (64 * 65 / 2 = 2080: one workgroup per bin up to 64 channels); more
// channels: further workgroups of the same bin (grid.y) take the next 2304 pairs each and stage the same frame values
constexpr int CSM_MAX_CH = 1024, CSM_PAIRS_PER_THREAD = 9, CSM_PAIRS_PER_WG = CSM_PAIRS_PER_THREAD * 256;
inline int csm_pair_groups(int n_ch) { return (n_ch * (n_ch + 1) / 2 + CSM_PAIRS_PER_WG - 1) / CSM_PAIRS_PER_WG; }

// One workgroup per bin: the bin's frame values X[c][f] go through LDS in tiles of frames, every thread sums up to nine
// (i2 >= i1) pairs conj(X_i1) X_i2 over the frames (fp64), finishes them and stores the element and its conjugate
// mirror (the reference's lower triangle + swapaxes, :351-369).  grid = (nb, csm_pair_groups(n_ch)), dynamic LDS =
// n_ch * tile * 16 bytes.
__global__ __launch_bounds__(256) void k_csm(CsmArgs p, int tile) {
    extern __shared__ __align__(16) double2 xt[];  // [tile][n_ch]
    const int b = blockIdx.x, nb = p.fin.nb, C = p.n_ch, F = p.n_frames, tid = threadIdx.x;
    const int pairs = C * (C + 1) / 2, q0 = (int)blockIdx.y * CSM_PAIRS_PER_WG;
    int i1s[CSM_PAIRS_PER_THREAD], i2s[CSM_PAIRS_PER_THREAD];
    cd acc[CSM_PAIRS_PER_THREAD];
#pragma unroll
    for (int s = 0; s < CSM_PAIRS_PER_THREAD; ++s) {
        const int q = q0 + tid + 256 * s;
        // pair q -> (i2, i1), i2 >= i1, rows of the lower triangle one after the other
        int i2 = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
        while ((i2 + 1) * (i2 + 2) / 2 <= q) ++i2;
        while (i2 * (i2 + 1) / 2 > q) --i2;
        i2s[s] = i2;
        i1s[s] = q - i2 * (i2 + 1) / 2;
        acc[s] = cd{0.0, 0.0};
    }
    for (int f0 = 0; f0 < F; f0 += tile) {
        const int nf = min(tile, F - f0);
        __syncthreads();
        for (int i = tid; i < nf * C; i += 256) {
            const int c = i / nf, f = i - c * nf;  // frames fastest: neighbouring threads read neighbouring frames
            xt[f * C + c] = p.xs[((size_t)c * F + f0 + f) * nb + b];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < CSM_PAIRS_PER_THREAD; ++s) {
            if (q0 + tid + 256 * s >= pairs) continue;
            const int i1 = i1s[s], i2 = i2s[s];
            cd a = acc[s];
            for (int f = 0; f < nf; ++f) {
                const double2 u = xt[f * C + i1], v = xt[f * C + i2];
                a.x += u.x * v.x + u.y * v.y;  // conj(u) v
                a.y += u.x * v.y - u.y * v.x;
            }
            acc[s] = a;
        }
    }
#pragma unroll
    for (int s = 0; s < CSM_PAIRS_PER_THREAD; ++s) {
        if (q0 + tid + 256 * s >= pairs) continue;
        const int i1 = i1s[s], i2 = i2s[s];
        double2* m = p.csm + (size_t)b * C * C;
        if (i1 == i2) {
            m[(size_t)i1 * C + i1] = make_double2(dsk::finish_real(acc[s].x, b, p.fin), 0.0);
        } else {
            const cd g = dsk::finish_cplx(cd{acc[s].x, acc[s].y + 0.0}, b, p.fin);
            m[(size_t)i2 * C + i1] = make_double2(g.x, g.y);
            m[(size_t)i1 * C + i2] = make_double2(g.x, 0.0 - g.y);
        }
    }
}

// average = "median" of the matrix (_csm_welch calls _welch per channel pair, :351-369, so every element is the median over
// the frames of the real and of the imaginary part of conj(X_i1) X_i2; the diagonal the median of |X_i|^2).  Short
// estimates only: F <= 128 frames, two per lane.  A workgroup takes one bin and one pair of channel tiles (32 + 32
// channels, their frame values in LDS: 2 * 32 * F * 16 bytes <= 128 KB), a wave one channel pair at a time: every lane
// ranks its two frame values against all F (ties by frame index, values read across the wave with v_readlane), the
// two middle ranks are averaged.  grid = (nb, csm_median_tile_pairs(n_ch)), 256 threads; the host folds the bias into fin.inv.
constexpr int CSM_MEDIAN_TILE = 32, CSM_MEDIAN_MAX_FRAMES = 128;
inline int csm_median_tiles(int n_ch) { return (n_ch + CSM_MEDIAN_TILE - 1) / CSM_MEDIAN_TILE; }
inline int csm_median_tile_pairs(int n_ch) { return csm_median_tiles(n_ch) * (csm_median_tiles(n_ch) + 1) / 2; }
inline size_t csm_median_lds(int n_ch, int n_frames) {
    return (size_t)(csm_median_tiles(n_ch) > 1 ? 2 : 1) * CSM_MEDIAN_TILE * n_frames * sizeof(double2);
}

__device__ __forceinline__ double wave_lane_value(double v, int lane) {  // lane is wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// median over the F values held as (v0: frame lane, v1: frame lane + 64) across one wave
__device__ __forceinline__ double wave_median(double v0, double v1, int F, int lane) {
    int rank0 = 0, rank1 = 0;
    const int n0 = min(F, 64);
    for (int j = 0; j < n0; ++j) {
        const double u = wave_lane_value(v0, j);
        rank0 += (u < v0 || (u == v0 && j < lane)) ? 1 : 0;
        rank1 += (u <= v1) ? 1 : 0;  // frame j < 64 <= frame of v1
    }
    for (int j = 64; j < F; ++j) {
        const double u = wave_lane_value(v1, j - 64);
        rank0 += (u < v0) ? 1 : 0;
        rank1 += (u < v1 || (u == v1 && j - 64 < lane)) ? 1 : 0;
    }
    // exactly one frame holds each rank for finite values (ties by frame index); NaN samples are outside the contract --
    // they compare false everywhere, so the ranks of the other frames close up and a finite order statistic comes out
    // where numpy's median would say NaN
    auto value_of_rank = [&](int r) {
        const unsigned long long hit0 = __ballot(lane < F && rank0 == r);
        const unsigned long long hit1 = __ballot(lane + 64 < F && rank1 == r);
        const double a = wave_lane_value(v0, hit0 ? __ffsll((long long)hit0) - 1 : 0);
        const double b = wave_lane_value(v1, hit1 ? __ffsll((long long)hit1) - 1 : 0);
        return hit0 ? a : (hit1 ? b : __longlong_as_double(0x7ff8000000000000ll));
    };
    return 0.5 * (value_of_rank((F - 1) / 2) + value_of_rank(F / 2));
}

__global__ __launch_bounds__(256) void k_csm_median(CsmArgs p) {
    extern __shared__ __align__(16) double2 xt[];  // tile A [32][F], then tile B [32][F] when it is another tile
    constexpr int T = CSM_MEDIAN_TILE;
    const int b = blockIdx.x, nb = p.fin.nb, C = p.n_ch, F = p.n_frames, tid = threadIdx.x;
    // tile pair blockIdx.y -> (tb >= ta), rows of the lower triangle one after the other
    int tb = 0;
    while ((tb + 1) * (tb + 2) / 2 <= (int)blockIdx.y) ++tb;
    const int ta = (int)blockIdx.y - tb * (tb + 1) / 2;
    double2* A = xt;
    double2* B = ta == tb ? xt : xt + (size_t)T * F;
    for (int i = tid; i < T * F; i += 256) {
        const int c = i / F, f = i - c * F;
        double2 va = make_double2(0.0, 0.0), vb = va;  // (channels past the end are never paired)
        if (ta * T + c < C) va = p.xs[((size_t)(ta * T + c) * F + f) * nb + b];
        if (ta != tb && tb * T + c < C) vb = p.xs[((size_t)(tb * T + c) * F + f) * nb + b];
        A[i] = va;
        if (ta != tb) B[i] = vb;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    double2* m = p.csm + (size_t)b * C * C;
    for (int q = wave; q < T * T; q += 4) {
        const int a = q / T, bb = q - a * T, i1 = ta * T + a, i2 = tb * T + bb;
        if (i1 >= C || i2 >= C || i2 < i1) continue;
        double re0 = inf, re1 = inf, im0 = inf, im1 = inf;  // frames past the end rank last
        if (lane < F) {
            const double2 u = A[a * F + lane], v = B[bb * F + lane];
            re0 = u.x * v.x + u.y * v.y;  // conj(u) v
            im0 = u.x * v.y - u.y * v.x;
        }
        if (lane + 64 < F) {
            const double2 u = A[a * F + lane + 64], v = B[bb * F + lane + 64];
            re1 = u.x * v.x + u.y * v.y;
            im1 = u.x * v.y - u.y * v.x;
        }
        const double mre = wave_median(re0, re1, F, lane);
        // the diagonal is _welch(x_i, None): the median of |X|^2, imaginary part 0
        const double mim = i1 == i2 ? 0.0 : wave_median(im0, im1, F, lane) + 0.0;
        if (lane != 0) continue;
        const cd g = dsk::finish_cplx(cd{mre, mim}, b, p.fin);
        if (i1 == i2) {
            m[(size_t)i1 * C + i1] = make_double2(g.x, 0.0);  // 0.5 g + conj(0.5 g)
        } else {
            m[(size_t)i2 * C + i1] = make_double2(g.x, g.y);
            m[(size_t)i1 * C + i2] = make_double2(g.x, 0.0 - g.y);
        }
    }
}

}  // namespace w64
