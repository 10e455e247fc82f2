// Generic (any power-of-two length 8..16384) kernels of the spectral hot path.
// One workgroup transforms one length-N complex FFT in LDS (fft_lds.hpp); real
// frames travel in pairs as real/imaginary parts and are separated with the
// Hermitian identity  A[k] = (Z[k] + conj Z[N-k])/2,  B[k] = (Z[k] - conj Z[N-k])/(2i).
//
// Reference semantics (dsptoolbox 0.8):
//   framing / zero padding   helpers/other.py:181-213, _framed_signal_representation.py:9-67
//   window, detrend-after-window, rfft   standard/_spectral_methods.py:126-148, 260-268
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_lds.hpp"

namespace dsk {
using namespace dsfft;

struct FrameSrc {
    const float* base;  // channel base pointer (may be nullptr: all zeros)
    int64_t start;      // sample index of frame sample 0 (may be negative / beyond the end)
};

// workgroup sum of a float2 (all threads get the result). red: >= 16 float2 of LDS.
template <int NT>
__device__ __forceinline__ float2 block_sum2(float2 s, float2* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s.x += __shfl_xor(s.x, o);
        s.y += __shfl_xor(s.y, o);
    }
    constexpr int NW = NT / 64;
    if constexpr (NW > 1) {
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        float2 r = red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            r.x += red[w].x;
            r.y += red[w].y;
        }
        __syncthreads();  // red reusable afterwards
        s = r;
    }
    return s;
}

// Raw samples of two real frames in the first-pass register layout (radix Cfg<N>::R1:
// slot i*R1 + t holds sample n = tid + i*NT + t*N/R1), zero for n >= W or outside
// [0, n_samples).  Kept apart from the windowing so a kernel can issue the loads of the next
// frame pair before it transforms the current one.
template <int N>
struct RawPair {
    float a[Cfg<N>::VMAX], b[Cfg<N>::VMAX];
};

template <int N>
__device__ __forceinline__ void load_raw_pair(RawPair<N>& r, FrameSrc a, FrameSrc b,
                                              int64_t n_samples, int W, int tid) {
    using C = Cfg<N>;
    const int span = W < N ? W : N;
    // interior frames: unconditional loads behind ONE team-uniform test (a per-load range
    // test makes hipcc branch around every load and drain vmcnt each time)
    const bool ia = a.base && a.start >= 0 && a.start + span <= n_samples;
    const bool ib = b.base && b.start >= 0 && b.start + span <= n_samples;
    if (ia && ib) {
        const float* __restrict__ pa = a.base + a.start;
        const float* __restrict__ pb = b.base + b.start;
        for_each_reg<N, C::R1>(tid, [&](int idx, int n) {
            r.a[idx] = n < W ? pa[n] : 0.f;
            r.b[idx] = n < W ? pb[n] : 0.f;
        });
    } else {
        // edges / padding / missing partner: clamp the address, select the value
        const float* __restrict__ pa = a.base ? a.base : b.base;
        const float* __restrict__ pb = b.base ? b.base : a.base;
        const int64_t last = n_samples - 1;
        for_each_reg<N, C::R1>(tid, [&](int idx, int n) {
            float xa = 0.f, xb = 0.f;
            if (n < W && pa) {
                int64_t ga = a.start + n, gb = b.start + n;
                int64_t ca = ga < 0 ? 0 : (ga > last ? last : ga);
                int64_t cb = gb < 0 ? 0 : (gb > last ? last : gb);
                float ta = pa[ca], tb = pb[cb];
                xa = (a.base && ga == ca) ? ta : 0.f;
                xb = (b.base && gb == cb) ? tb : 0.f;
            }
            r.a[idx] = xa;
            r.b[idx] = xb;
        });
    }
}

// z[n] = a[n] w[n] + i b[n] w[n]; optional detrend = subtract the mean over the W windowed
// samples (also those cropped away when W > N: they are read here).
template <int N>
__device__ __forceinline__ void window_pair(float2 (&v)[Cfg<N>::VMAX], const RawPair<N>& r, FrameSrc a,
                                            FrameSrc b, int64_t n_samples, int W,
                                            const float* __restrict__ window, bool detrend,
                                            float2* red, int tid) {
    using C = Cfg<N>;
    float2 sum = make_float2(0.f, 0.f);
    for_each_reg<N, C::R1>(tid, [&](int idx, int n) {
        const float w = (window && n < W) ? window[n] : 1.0f;
        float2 z = make_float2(r.a[idx] * w, r.b[idx] * w);
        v[idx] = z;
        sum.x += z.x;
        sum.y += z.y;
    });
    if (detrend) {
        for (int n = N + tid; n < W; n += C::NT) {  // samples cropped by nfft < W still count
            float w = window ? window[n] : 1.0f;
            int64_t ga = a.start + n, gb = b.start + n;
            if (a.base && ga >= 0 && ga < n_samples) sum.x += a.base[ga] * w;
            if (b.base && gb >= 0 && gb < n_samples) sum.y += b.base[gb] * w;
        }
        sum = block_sum2<C::NT>(sum, red, tid);
        float inv = 1.0f / (float)W;
        float2 m = make_float2(sum.x * inv, sum.y * inv);
        for_each_reg<N, C::R1>(tid, [&](int idx, int n) {
            if (n < W) {
                v[idx].x -= m.x;
                v[idx].y -= m.y;
            }
        });
    }
}

template <int N>
__device__ __forceinline__ void load_pair(float2 (&v)[Cfg<N>::VMAX], FrameSrc a, FrameSrc b,
                                          int64_t n_samples, int W,
                                          const float* __restrict__ window, bool detrend,
                                          float2* red, int tid) {
    RawPair<N> r;
    load_raw_pair<N>(r, a, b, n_samples, W, tid);
    window_pair<N>(v, r, a, b, n_samples, W, window, detrend, red, tid);
}

// spectra of the two packed real sequences at bin k (0 <= k <= N/2), from the
// natural-order complex spectrum in LDS
template <int N>
__device__ __forceinline__ void unpack_bin(const float2* __restrict__ buf, int k, float2& A,
                                           float2& B) {
    float2 P = buf[lidx(k)];
    float2 Qc = buf[lidx((N - k) & (N - 1))];  // Q = conj(Qc)
    A = make_float2(0.5f * (P.x + Qc.x), 0.5f * (P.y - Qc.y));
    // B = -i (P - Q)/2,  P - Q = (P.x - Qc.x, P.y + Qc.y)
    B = make_float2(0.5f * (P.y + Qc.y), -0.5f * (P.x - Qc.x));
}

template <int N>
struct Bins {
    static constexpr int NBINS = N / 2 + 1;
    static constexpr int BPB = (NBINS + Cfg<N>::NT - 1) / Cfg<N>::NT;  // bins per thread
};

// ---------------------------------------------------------------- STFT
// grid = (ceil(ceil(n_frames/2)/fpw), ceil(n_ch/ct)); block = ct teams of Cfg<N>::NT threads;
// out[(b*F + f)*C + c].
// Team j transforms the frame pair (f0, f0+1) of channel c0 + j in its own LDS buffer and
// separates the two half spectra IN PLACE (frame f0 bins at [k], frame f0+1 bins at [N-k],
// its bins 0 and N/2 in two spare slots).  Then the whole workgroup streams the images
// out with the channel index fastest: the reference's (bins, frames, channels) order is
// written in ct*8-byte contiguous runs.  Channel stride N + 33 complex keeps the
// channel-fastest LDS reads bank-conflict free.
struct StftArgs {
    const float* x;
    int64_t n_samples, ld, pad_front;
    int n_ch, W, hop, n_frames, detrend, power, ct, fpw;  // fpw: frame pairs per workgroup
    const float* window;
    const float2* tw;
    float scale, edge_scale;
    float2* out;
    int decim = 1;  // stft1k::k_stft_wave only: keep every decim-th bin (frames shorter than the 256-point transform)
};

// per-channel image: padded transform buffer (+2 spare slots), rounded up to 1 (mod 32) complex
// so the channel-fastest read-out hits distinct banks
template <int N>
__host__ __device__ constexpr int stft_ch_stride() { return ((lds_len<N>() + 2 + 30) / 32) * 32 + 1; }

// channel teams per workgroup the host may ask for: <= 16, <= 1024 threads, <= 74 KB of LDS (two
// workgroups per CU), a power of two -- and the thread count the kernel is compiled for (a blanket
// __launch_bounds__(1024) capped every length at 128 registers: 4096 points spilled 268 bytes)
template <int N>
__host__ __device__ constexpr int stft_max_teams() {
    int ct = 16;
    if (1024 / Cfg<N>::NT < ct) ct = 1024 / Cfg<N>::NT;
    const int by_lds = (int)((74 * 1024) / ((size_t)stft_ch_stride<N>() * sizeof(float2)));
    if (by_lds < ct) ct = by_lds < 1 ? 1 : by_lds;
    int p2 = 1;
    while (2 * p2 <= ct) p2 *= 2;
    return p2;
}
template <int N>
__global__ __launch_bounds__(stft_max_teams<N>() * Cfg<N>::NT) void k_stft(StftArgs p) {
    using C = Cfg<N>;
    constexpr int NB = N / 2 + 1, CHS = stft_ch_stride<N>();
    extern __shared__ __align__(16) float2 lds[];
    __shared__ float2 red_all[16][16];
    const int team = threadIdx.x / C::NT, tid = threadIdx.x % C::NT;
    float2* buf = lds + (int64_t)team * CHS;
    const int c0 = blockIdx.y * p.ct;
    const int ctv = min(p.ct, p.n_ch - c0);  // valid channels in this tile
    const int c = c0 + team;
    const float* xc = c < p.n_ch ? p.x + (int64_t)c * p.ld : nullptr;  // idle teams transform zeros
    const int64_t F = p.n_frames, Cn = p.n_ch;
    const int lct = __ffs(p.ct) - 1;  // ct is a power of two
    const int cl = threadIdx.x & (p.ct - 1);
    // this workgroup's frame pairs: fp0, fp0 + 1, ... (fpw of them); the raw samples of the
    // next pair are in flight while the current one is transformed and written out
    const int n_fp = (p.n_frames + 1) >> 1;
    const int fp0 = blockIdx.x * p.fpw, fp1 = min(fp0 + p.fpw, n_fp);
    auto src = [&](int fp, FrameSrc& a, FrameSrc& b) {
        const int f0 = 2 * fp, f1 = f0 + 1;
        a = FrameSrc{xc, (int64_t)f0 * p.hop - p.pad_front};
        b = FrameSrc{f1 < p.n_frames ? xc : nullptr, (int64_t)f1 * p.hop - p.pad_front};
    };
    RawPair<N> raw;
    FrameSrc a, b;
    if (fp0 < fp1) {
        src(fp0, a, b);
        load_raw_pair<N>(raw, a, b, p.n_samples, p.W, tid);
    }
    for (int fp = fp0; fp < fp1; ++fp) {
        const int f0 = 2 * fp;
        const bool v1 = f0 + 1 < p.n_frames;
        float2 v[C::VMAX];
        src(fp, a, b);
        __syncthreads();  // the previous pair has been streamed out of the LDS images
        window_pair<N>(v, raw, a, b, p.n_samples, p.W, p.window, p.detrend != 0, red_all[team], tid);
        if (fp + 1 < fp1) {
            FrameSrc na, nb;
            src(fp + 1, na, nb);
            load_raw_pair<N>(raw, na, nb, p.n_samples, p.W, tid);
        }
        fft<N, false, true, false>(v, buf, p.tw, tid);
        for (int k = tid; k <= N / 2; k += C::NT) {
            float2 A, B;
            unpack_bin<N>(buf, k, A, B);
            const bool edge = (k == 0 || k == N / 2);
            if (p.power) {
                // reference: |x/sqrt2|^2 * factor at the edges -> scale applied after squaring
                float e = (edge ? p.edge_scale * p.edge_scale : 1.0f) * p.scale;
                A = make_float2((A.x * A.x + A.y * A.y) * e, 0.f);
                B = make_float2((B.x * B.x + B.y * B.y) * e, 0.f);
            } else {
                float sc = p.scale * (edge ? p.edge_scale : 1.0f);
                A = make_float2(A.x * sc, A.y * sc);
                B = make_float2(B.x * sc, B.y * sc);
            }
            buf[lidx(k)] = A;  // this thread owns the pair (k, N-k): in place
            buf[k == 0 ? lidx(N) : (k == N / 2 ? lidx(N) + 1 : lidx(N - k))] = B;
        }
        __syncthreads();
        // (bin, frame, channel) from shifts, channel fastest; row = 2*k + frame
        if (cl < ctv) {
            for (int r = threadIdx.x >> lct; r < 2 * NB; r += blockDim.x >> lct) {
                const int fl = r & 1, k = r >> 1;
                if (fl && !v1) continue;
                const int sidx = fl == 0 ? lidx(k) : (k == 0 ? lidx(N) : (k == N / 2 ? lidx(N) + 1 : lidx(N - k)));
                p.out[((int64_t)k * F + f0 + fl) * Cn + c0 + cl] = lds[(int64_t)cl * CHS + sidx];
            }
        }
    }
}

// ---------------------------------------------------------------- Welch: input spectra
// grid = (n_chunks, n_cx).  Frames [q*fpc, min((q+1)*fpc, F)) of channel cx, two per
// FFT.  Stores Xs[(cx*F + f)*NB + b] (if xs) and the chunk's sum |X|^2 in
// pxx[(q*n_cx + cx)*NB + b].
struct XspecArgs {
    const float* x;
    int64_t n_samples, ld;
    int n_cx, W, hop, n_frames, detrend, fpc;  // fpc even
    const float* window;
    const float2* tw;
    float2* xs;  // may be nullptr
    float* pxx;
};

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_xspec(XspecArgs p) {
    using C = Cfg<N>;
    using BN = Bins<N>;
    extern __shared__ __align__(16) float2 buf[];
    __shared__ float2 red[16];
    const int tid = threadIdx.x;
    const int q = blockIdx.x, cx = blockIdx.y;
    const float* xc = p.x + (int64_t)cx * p.ld;
    const int fb = q * p.fpc, fe = min(fb + p.fpc, p.n_frames);
    float acc[BN::BPB];
#pragma unroll
    for (int s = 0; s < BN::BPB; ++s) acc[s] = 0.f;
    for (int f0 = fb; f0 < fe; f0 += 2) {
        const int f1 = f0 + 1;
        const bool v1 = f1 < fe;
        FrameSrc a{xc, (int64_t)f0 * p.hop};
        FrameSrc b{v1 ? xc : nullptr, (int64_t)f1 * p.hop};
        float2 v[C::VMAX];
        __syncthreads();  // previous iteration's unpack reads are done
        load_pair<N>(v, a, b, p.n_samples, p.W, p.window, p.detrend != 0, red, tid);
        fft<N, false, true, false>(v, buf, p.tw, tid);
#pragma unroll
        for (int s = 0; s < BN::BPB; ++s) {
            int k = tid + s * C::NT;
            if (k <= N / 2) {
                float2 A, B;
                unpack_bin<N>(buf, k, A, B);
                acc[s] += A.x * A.x + A.y * A.y;
                if (v1) acc[s] += B.x * B.x + B.y * B.y;
                if (p.xs) {
                    p.xs[((int64_t)cx * p.n_frames + f0) * BN::NBINS + k] = A;
                    if (v1) p.xs[((int64_t)cx * p.n_frames + f1) * BN::NBINS + k] = B;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < BN::BPB; ++s) {
        int k = tid + s * C::NT;
        if (k <= N / 2) p.pxx[((int64_t)q * p.n_cx + cx) * BN::NBINS + k] = acc[s];
    }
}

// ---------------------------------------------------------------- Welch: output channels
// grid = (n_chunks, ceil(n_cy/2)).  Channels (2p, 2p+1) ride one complex FFT per
// frame; cross power against the stored input spectra.
//   pxy[(q*n_cy + c)*NB + b] = sum_f conj(X_f[b]) Y_cf[b],  pyy[...] = sum_f |Y_cf[b]|^2
struct YaccArgs {
    const float* y;
    int64_t n_samples, ld;
    int n_cy, n_cx, W, hop, n_frames, detrend, fpc;
    const float* window;
    const float2* tw;
    const float2* xs;
    float2* pxy;
    float* pyy;
};

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_yacc(YaccArgs p) {
    using C = Cfg<N>;
    using BN = Bins<N>;
    extern __shared__ __align__(16) float2 buf[];
    __shared__ float2 red[16];
    const int tid = threadIdx.x;
    const int q = blockIdx.x;
    const int ca = 2 * blockIdx.y, cb = ca + 1;
    const bool vb = cb < p.n_cy;
    const float* ya = p.y + (int64_t)ca * p.ld;
    const float* yb = vb ? p.y + (int64_t)cb * p.ld : nullptr;
    const int xa = p.n_cx == 1 ? 0 : ca, xb = p.n_cx == 1 ? 0 : (vb ? cb : ca);
    const int fb = q * p.fpc, fe = min(fb + p.fpc, p.n_frames);
    float2 sxa[BN::BPB], sxb[BN::BPB];
    float sya[BN::BPB], syb[BN::BPB];
#pragma unroll
    for (int s = 0; s < BN::BPB; ++s) {
        sxa[s] = sxb[s] = make_float2(0.f, 0.f);
        sya[s] = syb[s] = 0.f;
    }
    for (int f = fb; f < fe; ++f) {
        FrameSrc a{ya, (int64_t)f * p.hop};
        FrameSrc b{yb, (int64_t)f * p.hop};
        float2 v[C::VMAX];
        __syncthreads();
        load_pair<N>(v, a, b, p.n_samples, p.W, p.window, p.detrend != 0, red, tid);
        fft<N, false, true, false>(v, buf, p.tw, tid);
        const float2* Xa = p.xs + ((int64_t)xa * p.n_frames + f) * BN::NBINS;
        const float2* Xb = p.xs + ((int64_t)xb * p.n_frames + f) * BN::NBINS;
#pragma unroll
        for (int s = 0; s < BN::BPB; ++s) {
            int k = tid + s * C::NT;
            if (k <= N / 2) {
                float2 A, B;
                unpack_bin<N>(buf, k, A, B);
                float2 X = Xa[k];
                float2 t = cmul_conj(A, X);  // A * conj(X) = conj(X) * A
                sxa[s].x += t.x;
                sxa[s].y += t.y;
                sya[s] += A.x * A.x + A.y * A.y;
                if (vb) {
                    float2 X2 = (xb == xa) ? X : Xb[k];
                    float2 u = cmul_conj(B, X2);
                    sxb[s].x += u.x;
                    sxb[s].y += u.y;
                    syb[s] += B.x * B.x + B.y * B.y;
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < BN::BPB; ++s) {
        int k = tid + s * C::NT;
        if (k <= N / 2) {
            int64_t ia = ((int64_t)q * p.n_cy + ca) * BN::NBINS + k;
            p.pxy[ia] = sxa[s];
            p.pyy[ia] = sya[s];
            if (vb) {
                int64_t ib = ((int64_t)q * p.n_cy + cb) * BN::NBINS + k;
                p.pxy[ib] = sxb[s];
                p.pyy[ib] = syb[s];
            }
        }
    }
}

// ---------------------------------------------------------------- whole-signal rFFT
// grid.x = ceil(n_ch/2); spec[b*n_ch + c] = rfft(x_c, N)[b] * scale
struct RfftArgs {
    const float* x;
    int64_t n_samples, ld;
    int n_ch;
    const float2* tw;
    float scale;
    float2* spec;
};

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_rfft(RfftArgs p) {
    using C = Cfg<N>;
    extern __shared__ __align__(16) float2 buf[];
    __shared__ float2 red[16];
    const int tid = threadIdx.x;
    const int ca = 2 * blockIdx.x, cb = ca + 1;
    FrameSrc a{p.x + (int64_t)ca * p.ld, 0};
    FrameSrc b{cb < p.n_ch ? p.x + (int64_t)cb * p.ld : nullptr, 0};
    float2 v[C::VMAX];
    load_pair<N>(v, a, b, p.n_samples, N, nullptr, false, red, tid);
    fft<N, false, true, false>(v, buf, p.tw, tid);
    for (int k = tid; k <= N / 2; k += C::NT) {
        float2 A, B;
        unpack_bin<N>(buf, k, A, B);
        p.spec[(int64_t)k * p.n_ch + ca] = make_float2(A.x * p.scale, A.y * p.scale);
        if (cb < p.n_ch) p.spec[(int64_t)k * p.n_ch + cb] = make_float2(B.x * p.scale, B.y * p.scale);
    }
}

// ---------------------------------------------------------------- spectral division
// grid = (ceil(n_ch/2), n_items): ir = irfft(rfft(y, N) * r, N)[:n_out]
struct DeconvArgs {
    const float* y;
    int64_t n_samples, ld, n_out, ld_out;
    int n_ch, r_per_channel;
    const float2* tw;
    const float2* r;  // [n_ch or 1][N/2+1]
    float* ir;
};

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_deconv(DeconvArgs p) {
    using C = Cfg<N>;
    constexpr int NB = N / 2 + 1;
    extern __shared__ __align__(16) float2 buf[];
    __shared__ float2 red[16];
    const int tid = threadIdx.x;
    const int ca = 2 * blockIdx.x, cb = ca + 1;
    const bool vb = cb < p.n_ch;
    const int64_t item = blockIdx.y;
    const float* ya = p.y + (item * p.n_ch + ca) * p.ld;
    FrameSrc a{ya, 0};
    FrameSrc b{vb ? ya + p.ld : nullptr, 0};
    float2 v[C::VMAX];
    load_pair<N>(v, a, b, p.n_samples, N, nullptr, false, red, tid);
    fft<N, false, true, false>(v, buf, p.tw, tid);
    const float2* Ra = p.r + (p.r_per_channel ? (int64_t)ca * NB : 0);
    const float2* Rb = p.r + (p.r_per_channel ? (int64_t)(vb ? cb : ca) * NB : 0);
    for (int k = tid; k <= N / 2; k += C::NT) {
        float2 A, B;
        unpack_bin<N>(buf, k, A, B);
        float2 VA = cmul(A, Ra[k]), VB = cmul(B, Rb[k]);
        if (k == 0 || k == N / 2) {  // irfft ignores the imaginary part there
            buf[lidx(k)] = make_float2(VA.x, VB.x);
        } else {
            buf[lidx(k)] = make_float2(VA.x - VB.y, VA.y + VB.x);      // VA + i VB
            buf[lidx(N - k)] = make_float2(VA.x + VB.y, VB.x - VA.y);  // conj(VA) + i conj(VB)
        }
    }
    __syncthreads();
    fft<N, true, false, false>(v, buf, p.tw, tid);
    const float inv = 1.0f / (float)N;
    float* oa = p.ir + (item * p.n_ch + ca) * p.ld_out;
    float* ob = oa + p.ld_out;
    for (int n = tid; n < N && n < p.n_out; n += C::NT) {
        float2 z = buf[lidx(n)];
        oa[n] = z.x * inv;
        if (vb) ob[n] = z.y * inv;
    }
}

// ---------------------------------------------------------------- inverse STFT
// (reference: transforms.istft, transforms/transforms.py:444-586)
// k_istft: grid = (ceil(F/2), C).  Frames (f0, f0+1) of channel c: Z = A + i B from the one-sided
// spectra stft[(k*F + f)*C + c] (bins beyond n_bins are zero, imaginary parts of bins 0 and N/2
// ignored, like numpy.fft.irfft), inverse transform, * scale * window, first W samples ->
// frames[(c*F + f)*W + n].
struct IstftArgs {
    const float2* stft;
    int n_bins, n_frames, n_ch, W;
    const float* window;
    const float2* tw;
    float scale;
    float* frames;
    int ct = 1, fpw = 1;  // k_istft_ct: channels (teams) per workgroup, frame pairs per workgroup
};

// The same with ct channels per workgroup (the mirror image of k_stft's tile): the spectrogram is
// channel-fastest, so ONE channel per workgroup reads 8 bytes out of every 128-byte line it touches
// (64 channels x 512 000 samples: 0.79 ms at every window length, 0.5 TB/s).  Here the workgroup reads the
// bins of ct neighbouring channels together -- runs of 8 ct bytes -- straight into the ct LDS images
// (Z = A + i B and its mirror half), every team transforms its own image back and writes its two frames,
// which are contiguous per channel.  grid = (ceil(ceil(F/2) / fpw), ceil(C / ct)), block = ct * NT.
template <int N>
__global__ __launch_bounds__(stft_max_teams<N>() * Cfg<N>::NT) void k_istft_ct(IstftArgs p) {
    using C = Cfg<N>;
    constexpr int CHS = stft_ch_stride<N>();
    extern __shared__ __align__(16) float2 lds[];
    const int team = threadIdx.x / C::NT, tid = threadIdx.x % C::NT;
    float2* buf = lds + (int64_t)team * CHS;
    const int c0 = blockIdx.y * p.ct;
    const int ctv = min(p.ct, p.n_ch - c0);
    const int c = c0 + team;
    const int64_t F = p.n_frames, Cn = p.n_ch;
    const int lct = __ffs(p.ct) - 1;  // ct is a power of two
    const int cl = threadIdx.x & (p.ct - 1);
    float2* img = lds + (int64_t)cl * CHS;
    const int n_fp = (p.n_frames + 1) >> 1;
    const int fp0 = blockIdx.x * p.fpw, fp1 = min(fp0 + p.fpw, n_fp);
    for (int fp = fp0; fp < fp1; ++fp) {
        const int f0 = 2 * fp;
        const bool v1 = f0 + 1 < p.n_frames;
        __syncthreads();  // the frames of the previous pair have been read out of the images
        for (int k = threadIdx.x >> lct; k <= N / 2; k += blockDim.x >> lct) {
            float2 A = make_float2(0.f, 0.f), B = make_float2(0.f, 0.f);
            if (cl < ctv && k < p.n_bins) {
                const float2* s = p.stft + ((int64_t)k * F + f0) * Cn + c0 + cl;
                A = s[0];
                if (v1) B = s[Cn];
            }
            if (k == 0 || k == N / 2) {
                img[lidx(k)] = make_float2(A.x, B.x);
            } else {
                img[lidx(k)] = make_float2(A.x - B.y, A.y + B.x);      // A + i B
                img[lidx(N - k)] = make_float2(A.x + B.y, B.x - A.y);  // conj(A) + i conj(B)
            }
        }
        __syncthreads();
        float2 v[C::VMAX];
        fft<N, true, false, false>(v, buf, p.tw, tid);
        if (c < p.n_ch) {
            float* oa = p.frames + ((int64_t)c * F + f0) * p.W;
            float* ob = oa + p.W;
            for (int n = tid; n < p.W && n < N; n += C::NT) {
                const float2 z = buf[lidx(n)];
                const float w = p.window[n] * p.scale;
                oa[n] = z.x * w;
                if (v1) ob[n] = z.y * w;
            }
        }
    }
}

// k_istft_ct with the overlap-add folded in, for THE common case W == N, step == N / 2 (50 % overlap): a frame's first
// half completes the second half of the frame before it, so a team thread carries N / (2 NT) sums in registers from
// one frame pair to the next and writes finished output samples -- the frames never go to memory (263 MB written and
// read back for 64 channels x 512 000 samples) and the separate overlap-add launch disappears.  A workgroup owns the
// output from the start of its first frame to the start of the next workgroup's; it first transforms the frame
// pair in front of its range to obtain the carry (the last one also writes what follows the last frame).
// out[c * ld + n] = sum / max(sum of w^2 over the frame slots [0, n_total) that cover n, 1e-4)  (k_istft_ola).
struct IstftFusedArgs {
    IstftArgs a;
    int off, n_total;
    int64_t total_length, ld;
    float* out;
};
template <int N>
__global__ __launch_bounds__(stft_max_teams<N>() * Cfg<N>::NT) void k_istft_fused(IstftFusedArgs q) {
    using C = Cfg<N>;
    const IstftArgs& p = q.a;
    // (the host launches it only where half a frame is a whole number of values per thread: N % (2 NT) == 0)
    constexpr int CHS = stft_ch_stride<N>(), NT = C::NT, H = N / (2 * NT) > 0 ? N / (2 * NT) : 1, STEP = N / 2;
    extern __shared__ __align__(16) float2 lds[];
    const int team = threadIdx.x / NT, tid = threadIdx.x % NT;
    float2* buf = lds + (int64_t)team * CHS;
    const int c0 = blockIdx.y * p.ct;
    const int ctv = min(p.ct, p.n_ch - c0);
    const int c = c0 + team;
    const int64_t F = p.n_frames, Cn = p.n_ch;
    const int lct = __ffs(p.ct) - 1;
    const int cl = threadIdx.x & (p.ct - 1);
    float2* img = lds + (int64_t)cl * CHS;
    const int n_fp = (p.n_frames + 1) >> 1;
    const int fp0 = blockIdx.x * p.fpw, fp1 = min(fp0 + p.fpw, n_fp);
    if (fp0 >= fp1) return;
    float* oc = q.out + (int64_t)(c < p.n_ch ? c : 0) * q.ld;
    // 1 / envelope wherever two frame slots cover a position (everywhere but at the ends of the signal): it depends
    // on the position within the step only (a per-sample float64 division tripled the vector work of a team)
    float* inv_env = reinterpret_cast<float*>(lds + (int64_t)p.ct * CHS);
    for (int m = threadIdx.x; m < STEP; m += blockDim.x) {
        const double w0 = (double)p.window[m], w1 = (double)p.window[m + STEP];
        const double e = w0 * w0 + w1 * w1;
        inv_env[m] = (float)(1.0 / (e < 1e-4 ? 1e-4 : e));
    }
    // envelope of the window at output position pos: the frame slots fs = pos / STEP and fs - 1 (both within [0, n_total))
    auto emit = [&](int64_t pos, float sum) {
        if (pos < 0 || pos >= q.total_length) return;
        const int64_t fs = pos / STEP;
        const int m = (int)(pos - fs * STEP);
        double env = 0.0;
        if (fs < q.n_total) {
            const float w = p.window[m];
            env += (double)w * (double)w;
        }
        if (fs >= 1 && fs - 1 < q.n_total) {
            const float w = p.window[m + STEP];
            env += (double)w * (double)w;
        }
        oc[pos] = (float)((double)sum / (env < 1e-4 ? 1e-4 : env));
    };
    float carry[H];
#pragma unroll
    for (int j = 0; j < H; ++j) carry[j] = 0.f;
    // positions in front of the first frame slot (off = 1: the reference's empty frame in front) are written by the
    // first workgroup
    if (fp0 == 0 && c < p.n_ch)
        for (int64_t n = tid; n < (int64_t)q.off * STEP; n += NT) emit(n, 0.f);
    for (int fp = fp0 > 0 ? fp0 - 1 : 0; fp < fp1; ++fp) {
        const int f0 = 2 * fp;
        const bool v1 = f0 + 1 < p.n_frames;
        const bool owned = fp >= fp0;  // the pair in front of the range only yields the carry
        __syncthreads();  // the previous pair has been read out of the images
        for (int k = threadIdx.x >> lct; k <= N / 2; k += blockDim.x >> lct) {
            float2 A = make_float2(0.f, 0.f), B = make_float2(0.f, 0.f);
            if (cl < ctv && k < p.n_bins) {
                const float2* s = p.stft + ((int64_t)k * F + f0) * Cn + c0 + cl;
                A = s[0];
                if (v1) B = s[Cn];
            }
            if (k == 0 || k == N / 2) {
                img[lidx(k)] = make_float2(A.x, B.x);
            } else {
                img[lidx(k)] = make_float2(A.x - B.y, A.y + B.x);      // A + i B
                img[lidx(N - k)] = make_float2(A.x + B.y, B.x - A.y);  // conj(A) + i conj(B)
            }
        }
        __syncthreads();
        float2 v[C::VMAX];
        fft<N, true, false, false>(v, buf, p.tw, tid);
        if (c < p.n_ch) {
            const int64_t P0 = (int64_t)(f0 + q.off) * STEP;
            // both covering frame slots exist: fs - 1 >= 0 and fs < n_total (uniform per segment)
            const bool fast0 = f0 + q.off >= 1 && f0 + q.off < q.n_total;
            const bool fast1 = f0 + q.off + 1 < q.n_total;
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int n = tid + NT * j;
                const float2 lo = buf[lidx(n)], hi = buf[lidx(n + STEP)];
                const float wl = p.window[n] * p.scale, wh = p.window[n + STEP] * p.scale;
                if (owned) {
                    const float s0 = carry[j] + lo.x * wl;       // second half of the frame before + first half of f0
                    const float s1 = hi.x * wh + lo.y * wl;      // second half of f0 + first half of f0 + 1
                    if (fast0 && P0 + n < q.total_length)
                        oc[P0 + n] = s0 * inv_env[n];
                    else
                        emit(P0 + n, s0);
                    if (fast1 && P0 + STEP + n < q.total_length)
                        oc[P0 + STEP + n] = s1 * inv_env[n];
                    else
                        emit(P0 + STEP + n, s1);
                }
                carry[j] = hi.y * wh;                                // second half of f0 + 1 (zero if it does not exist)
            }
        }
    }
    // behind the last frame: its second half, then nothing but the envelope's floor
    if (fp1 == n_fp && c < p.n_ch) {
        const int64_t P = (int64_t)(2 * n_fp + q.off) * STEP;
#pragma unroll
        for (int j = 0; j < H; ++j) emit(P + tid + NT * j, carry[j]);
        for (int64_t n = P + STEP + tid; n < q.total_length; n += NT) emit(n, 0.f);
    }
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_istft(IstftArgs p) {
    using C = Cfg<N>;
    extern __shared__ __align__(16) float2 buf[];
    const int tid = threadIdx.x;
    const int f0 = 2 * blockIdx.x, c = blockIdx.y;
    const bool v1 = f0 + 1 < p.n_frames;
    const int64_t F = p.n_frames, Cn = p.n_ch;
    for (int k = tid; k <= N / 2; k += C::NT) {
        float2 A = make_float2(0.f, 0.f), B = make_float2(0.f, 0.f);
        if (k < p.n_bins) {
            A = p.stft[((int64_t)k * F + f0) * Cn + c];
            if (v1) B = p.stft[((int64_t)k * F + f0 + 1) * Cn + c];
        }
        if (k == 0 || k == N / 2) {
            buf[lidx(k)] = make_float2(A.x, B.x);
        } else {
            buf[lidx(k)] = make_float2(A.x - B.y, A.y + B.x);      // A + i B
            buf[lidx(N - k)] = make_float2(A.x + B.y, B.x - A.y);  // conj(A) + i conj(B)
        }
    }
    __syncthreads();
    float2 v[C::VMAX];
    fft<N, true, false, false>(v, buf, p.tw, tid);
    float* oa = p.frames + ((int64_t)c * F + f0) * p.W;
    float* ob = oa + p.W;
    for (int n = tid; n < p.W && n < N; n += C::NT) {
        const float2 z = buf[lidx(n)];
        const float w = p.window[n] * p.scale;
        oa[n] = z.x * w;
        if (v1) ob[n] = z.y * w;
    }
}

// k_istft_ola: out[c*ld + n] = sum_f frames[c][f][n - (f + off)*step] / max(sum_f' w^2[n - f'*step], 1e-4)
// over the frames that cover sample n (f' = f + off runs over n_total frame slots; the reference
// adds a zero frame before and after the data when the signal was not padded, off = 1).
struct IstftOlaArgs {
    const float* frames;
    int n_frames, n_ch, W, step, off, n_total;
    const float* window;
    int64_t total_length, ld;
    float* out;
};
__global__ void k_istft_ola(IstftOlaArgs p) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (n >= p.total_length) return;
    // (32-bit quotients wherever the signal is shorter than 2^31 samples: the two 64-bit integer divisions were
    // most of this kernel's time)
    int64_t lo, hi;
    if (p.total_length < ((int64_t)1 << 31)) {
        const unsigned nn = (unsigned)n, st = (unsigned)p.step;
        lo = n - p.W + 1 <= 0 ? 0 : (int64_t)((nn - (unsigned)p.W + st) / st);  // ceil((n - W + 1) / step)
        hi = (int64_t)(nn / st);
    } else {
        lo = n - p.W + 1 <= 0 ? 0 : (n - p.W + p.step) / p.step;
        hi = n / p.step;
    }
    if (hi > p.n_total - 1) hi = p.n_total - 1;
    double acc = 0.0, env = 0.0;
    for (int64_t fs = lo; fs <= hi; ++fs) {
        const int m = (int)(n - fs * p.step);
        const float w = p.window[m];
        env += (double)w * (double)w;
        const int64_t f = fs - p.off;
        if (f >= 0 && f < p.n_frames) acc += (double)p.frames[((int64_t)c * p.n_frames + f) * p.W + m];
    }
    if (env < 1e-4) env = 1e-4;
    p.out[(int64_t)c * p.ld + n] = (float)(acc / env);
}

// Four neighbouring samples per thread, 16-byte loads and stores: when the window length, the step, the row pitch
// and the total length are multiples of 4 the four samples are covered by the same frames (frame boundaries fall
// on multiples of the step).  grid = (ceil(total_length / 4 / 256), n_ch).
__global__ void k_istft_ola4(IstftOlaArgs p) {
    const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int c = blockIdx.y;
    if (n >= p.total_length) return;
    int64_t lo, hi;
    if (p.total_length < ((int64_t)1 << 31)) {
        const unsigned nn = (unsigned)n, st = (unsigned)p.step;
        lo = n + 3 - p.W + 1 <= 0 ? 0 : (int64_t)((nn + 3u - (unsigned)p.W + st) / st);
        hi = (int64_t)(nn / st);
    } else {
        lo = n + 3 - p.W + 1 <= 0 ? 0 : (n + 3 - p.W + p.step) / p.step;
        hi = n / p.step;
    }
    if (hi > p.n_total - 1) hi = p.n_total - 1;
    double acc[4] = {0.0, 0.0, 0.0, 0.0}, env[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t fs = lo; fs <= hi; ++fs) {
        const int m = (int)(n - fs * p.step);  // multiple of 4, 0 <= m <= W - 4
        const float4 w = *reinterpret_cast<const float4*>(p.window + m);
        env[0] += (double)w.x * (double)w.x;
        env[1] += (double)w.y * (double)w.y;
        env[2] += (double)w.z * (double)w.z;
        env[3] += (double)w.w * (double)w.w;
        const int64_t f = fs - p.off;
        if (f >= 0 && f < p.n_frames) {
            const float4 v = *reinterpret_cast<const float4*>(p.frames + ((int64_t)c * p.n_frames + f) * p.W + m);
            acc[0] += (double)v.x;
            acc[1] += (double)v.y;
            acc[2] += (double)v.z;
            acc[3] += (double)v.w;
        }
    }
    float4 o;
    o.x = (float)(acc[0] / (env[0] < 1e-4 ? 1e-4 : env[0]));
    o.y = (float)(acc[1] / (env[1] < 1e-4 ? 1e-4 : env[1]));
    o.z = (float)(acc[2] / (env[2] < 1e-4 ? 1e-4 : env[2]));
    o.w = (float)(acc[3] / (env[3] < 1e-4 ? 1e-4 : env[3]));
    *reinterpret_cast<float4*>(p.out + (int64_t)c * p.ld + n) = o;
}

// ---------------------------------------------------------------- band powers of a spectrogram
// (reference: np.tensordot(mel_filters, np.abs(stft)**2, axes=(-1, 0)) followed by to_db(., False),
// transforms/transforms.py:181-184 and :421-424).  grid = (ceil(F*C / 256), n_bands).
//   out[(band*F + f)*C + c] = sum_b w[band][b] |X[b][f][c]|^2  over b in [b0[band], b1[band])
// (the filters are banded: only their non-zero bin range is read), optionally
// 10 log10(max(., DBL_MIN)).  fp64 accumulation.
struct BandPowerArgs {
    const float2* X;  // [n_bins][F][C]
    const float* w;   // [n_bands][n_bins]
    const int* b0;
    const int* b1;
    int n_bins, n_bands;
    int64_t n_fc;  // F * C
    int to_db;
    float* out;
};
__global__ void k_band_power(BandPowerArgs p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int band = blockIdx.y;
    if (i >= p.n_fc) return;
    const float* w = p.w + (int64_t)band * p.n_bins;
    double acc = 0.0;
    for (int b = p.b0[band]; b < p.b1[band]; ++b) {
        const float2 x = p.X[(int64_t)b * p.n_fc + i];
        acc += (double)w[b] * ((double)x.x * (double)x.x + (double)x.y * (double)x.y);
    }
    if (p.to_db) {
        const double a = fabs(acc);
        acc = 10.0 * log10(a < 2.2250738585072014e-308 ? 2.2250738585072014e-308 : a);
    }
    p.out[(int64_t)band * p.n_fc + i] = (float)acc;
}

// |DCT-II| along the band axis (scipy.fft.dct(type=2, axis=0), unnormalised), NaN -> 0:
//   out[k][i] = | 2 sum_n x[n][i] cos(pi k (2n + 1) / (2 N)) |      (mfcc, transforms.py:426-429)
struct DctArgs {
    const float* x;  // [N][n_fc]
    int n;
    int64_t n_fc;
    float* out;
};
__global__ void k_dct2_abs(DctArgs p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= p.n_fc) return;
    double acc = 0.0;
    for (int n = 0; n < p.n; ++n) {
        double s, c;
        sincospi((double)k * (double)(2 * n + 1) / (double)(2 * p.n), &s, &c);
        acc += (double)p.x[(int64_t)n * p.n_fc + i] * c;
    }
    acc = fabs(2.0 * acc);
    p.out[(int64_t)k * p.n_fc + i] = (acc != acc) ? 0.f : (float)acc;
}

// ---------------------------------------------------------------- FIR block convolution
// tap spectra: grid.x = ceil(n_filt/2); hs[k*N + m] = fft(taps_k zero padded)[m] / N
struct FirTapsArgs {
    const float* taps;
    int n_filt, n_taps;
    const float2* tw;
    float2* hs;
};

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_fir_taps(FirTapsArgs p) {
    using C = Cfg<N>;
    extern __shared__ __align__(16) float2 buf[];
    __shared__ float2 red[16];
    const int tid = threadIdx.x;
    const int ka = 2 * blockIdx.x, kb = ka + 1;
    FrameSrc a{p.taps + (int64_t)ka * p.n_taps, 0};
    FrameSrc b{kb < p.n_filt ? p.taps + (int64_t)kb * p.n_taps : nullptr, 0};
    float2 v[C::VMAX];
    load_pair<N>(v, a, b, p.n_taps, N, nullptr, false, red, tid);
    fft<N, false, true, false>(v, buf, p.tw, tid);
    const float inv = 1.0f / (float)N;
    for (int k = tid; k <= N / 2; k += C::NT) {
        float2 A, B;
        unpack_bin<N>(buf, k, A, B);
        A = make_float2(A.x * inv, A.y * inv);
        B = make_float2(B.x * inv, B.y * inv);
        float2* ha = p.hs + (int64_t)ka * N;
        ha[k] = A;
        if (k != 0 && k != N / 2) ha[N - k] = make_float2(A.x, -A.y);
        if (kb < p.n_filt) {
            float2* hb = ha + N;
            hb[k] = B;
            if (k != 0 && k != N / 2) hb[N - k] = make_float2(B.x, -B.y);
        }
    }
}

// grid = (n_blocks, ceil(n_ch/2)).  Overlap-save form of the overlap-add
// convolution: block j transforms input samples [j*L - (T-1), j*L - (T-1) + N),
// L = N - T + 1, multiplies by every filter spectrum and keeps the last L
// outputs of each inverse transform = y[j*L .. j*L+L).  Every output sample is
// produced once and stored once (no read-modify-write add-back pass); two
// channels ride one complex transform, so no Hermitian separation is needed:
// ifft(fft(xa + i xb) H) = ya + i yb.
struct FirArgs {
    const float* x;
    int64_t n_samples, ldx, ld_y;
    int n_ch, n_filt, n_taps;
    const float2* tw;
    const float2* hs;  // [n_filt][N], 1/N folded in
    float* y;          // [(k*n_ch + c)*ld_y + n]
};

template <int N>
__global__ __launch_bounds__(Cfg<N>::NT) void k_fir(FirArgs p) {
    using C = Cfg<N>;
    constexpr int RZ = last_radix<N, false>();  // layout of the spectrum kept in registers
    constexpr int RO = last_radix<N, true>();   // layout of the time-domain block
    static_assert(RZ == first_radix<N, true>(), "forward output layout must feed the inverse");
    extern __shared__ __align__(16) float2 buf[];
    __shared__ float2 red[16];
    const int tid = threadIdx.x;
    const int T1 = p.n_taps - 1;
    const int L = N - T1;
    const int64_t out0 = (int64_t)blockIdx.x * L;
    const int ca = 2 * blockIdx.y, cb = ca + 1;
    const bool vb = cb < p.n_ch;
    FrameSrc a{p.x + (int64_t)ca * p.ldx, out0 - T1};
    FrameSrc b{vb ? p.x + (int64_t)cb * p.ldx : nullptr, out0 - T1};
    float2 z[C::VMAX], v[C::VMAX];
    load_pair<N>(z, a, b, p.n_samples, N, nullptr, false, red, tid);
    fft<N, false, true, true>(z, buf, p.tw, tid);  // spectrum stays in registers
    for (int k = 0; k < p.n_filt; ++k) {
        const float2* H = p.hs + (int64_t)k * N;
        for_each_reg<N, RZ>(tid, [&](int idx, int n) { v[idx] = cmul(z[idx], H[n]); });
        __syncthreads();  // LDS free: the previous transform's last pass has read it
        fft<N, true, true, true, true>(v, buf, p.tw, tid);  // reversed radix order
        float* oa = p.y + ((int64_t)k * p.n_ch + ca) * p.ld_y;
        float* ob = oa + p.ld_y;
        for_each_reg<N, RO>(tid, [&](int idx, int n) {
            int64_t g = out0 + (n - T1);
            if (n >= T1 && g < p.n_samples) {
                oa[g] = v[idx].x;
                if (vb) ob[g] = v[idx].y;
            }
        });
    }
}

}  // namespace dsk
