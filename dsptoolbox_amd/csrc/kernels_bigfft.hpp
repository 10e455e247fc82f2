// Four-step FFT for power-of-two lengths beyond one workgroup's LDS (N = N1 * N2,
// 2^15 .. 2^24), used by the whole-signal paths (ds_rfft, ds_deconv).
//   view z[n1*N2 + j2]:
//   columns  (k_big_cols): for each column j2, FFT over n1 (length N1, stride N2), times
//            W_N^(j2 k1), in place -> t[k1*N2 + j2]
//   rows     (k_big_rows): for each row k1, FFT over j2 (length N2, contiguous)
//            -> X[k1 + N1 k2] stored at out[k2*N1 + k1] (natural order)
// Both kernels work on CT columns / rows per workgroup (CT teams of Cfg<N>::NT threads,
// each with its own LDS buffer, channel stride N+33 complex) so that every global access
// is a CT*8-byte contiguous run.  Twiddles W_N^m come from sincospi in fp64 (exact to
// fp32 rounding for any N).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_lds.hpp"

namespace dsbig {
using namespace dsfft;

template <int N>
__host__ __device__ constexpr int ch_stride() { return ((lds_len<N>() + 30) / 32) * 32 + 1; }

struct ColsArgs {
    // source: complex buffer `zin` (in place allowed) or, if xa != nullptr, two real
    // signals xa + i xb (xb may be nullptr) zero padded beyond n_samples
    const float2* zin;
    const float* xa;
    const float* xb;
    int64_t n_samples;
    int64_t batch_stride_real;  // between consecutive batch entries of xa / xb (floats)
    float2* zout;               // [batch][N]
    int64_t n_total;            // N
    int n2, ct, pairs;          // pairs: real-source batch = channel pairs (xb = xa + ld)
    int64_t ld_real;
    int n_ch;
    const float2* tw;  // length-N1 table
    // frame mode (win != nullptr): batch entry bt0 + blockIdx.y = (channel, frame pair) of the
    // framed signal xa[channel]; frames 2p and 2p+1 travel as real / imaginary part, windowed,
    // minus means[channel*n_frames + frame] if means != nullptr (detrend), zero beyond
    // min(W, N) and outside [0, n_samples)
    const float* win;
    const float* means;
    int W, hop, n_frames;
    int64_t pad_front, bt0;
    int64_t x_off;  // real channel-pair source: sample index of n = 0 (may be negative: zeros)
};

template <int N1>
__global__ __launch_bounds__(1024) void k_big_cols(ColsArgs p) {
    using C = Cfg<N1>;
    constexpr int CHS = ch_stride<N1>();
    extern __shared__ __align__(16) float2 lds[];
    const int team = threadIdx.x / C::NT, tid = threadIdx.x % C::NT;
    const int j20 = blockIdx.x * p.ct;
    const int64_t bt = blockIdx.y;
    const int64_t N = p.n_total;
    float2* zo = p.zout + bt * N;
    // cooperative coalesced load: rows n1, ct consecutive columns
    const int total = N1 * p.ct;
    if (p.win) {
        const int64_t g = p.bt0 + bt;
        const int nfp = (p.n_frames + 1) / 2;
        const int ch = (int)(g / nfp), f0 = 2 * (int)(g % nfp);
        const bool v1 = f0 + 1 < p.n_frames;
        const float* xc = p.xa + (int64_t)ch * p.ld_real;
        const int64_t sa = (int64_t)f0 * p.hop - p.pad_front, sb = sa + p.hop;
        float ma = 0.f, mb = 0.f;
        if (p.means) {
            ma = p.means[(int64_t)ch * p.n_frames + f0];
            if (v1) mb = p.means[(int64_t)ch * p.n_frames + f0 + 1];
        }
        const int64_t span = p.W < N ? p.W : N;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            int j = i % p.ct, n1 = i / p.ct;
            int64_t n = (int64_t)n1 * p.n2 + j20 + j;
            float2 z = make_float2(0.f, 0.f);
            if (n < span) {
                const float w = p.win[n];
                const int64_t ga = sa + n, gb = sb + n;
                const float xa = (ga >= 0 && ga < p.n_samples) ? xc[ga] : 0.f;
                const float xb = (v1 && gb >= 0 && gb < p.n_samples) ? xc[gb] : 0.f;
                z.x = xa * w - ma;
                z.y = v1 ? xb * w - mb : 0.f;
            }
            lds[j * CHS + lidx(n1)] = z;
        }
    } else if (p.xa) {
        const int ca = 2 * (int)(bt % ((p.n_ch + 1) / 2)), item = (int)(bt / ((p.n_ch + 1) / 2));
        const float* a = p.xa + ((int64_t)item * p.n_ch + ca) * p.ld_real;
        const float* b = (ca + 1 < p.n_ch) ? a + p.ld_real : nullptr;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            int j = i % p.ct, n1 = i / p.ct;
            int64_t n = (int64_t)n1 * p.n2 + j20 + j;
            float2 z = make_float2(0.f, 0.f);
            const int64_t g = n + p.x_off;
            if (g >= 0 && g < p.n_samples) {
                z.x = a[g];
                if (b) z.y = b[g];
            }
            lds[j * CHS + lidx(n1)] = z;
        }
    } else {
        const float2* zi = p.zin + bt * N;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            int j = i % p.ct, n1 = i / p.ct;
            lds[j * CHS + lidx(n1)] = zi[(int64_t)n1 * p.n2 + j20 + j];
        }
    }
    __syncthreads();
    float2* buf = lds + team * CHS;
    float2 v[C::VMAX];
    fft<N1, false, false, false>(v, buf, p.tw, tid);
    team_barrier<N1>();
    // twiddle W_N^(j2 k1)
    const int j2 = j20 + team;
    for (int k1 = tid; k1 < N1; k1 += C::NT) {
        double s, c;
        sincospi(-2.0 * (double)((int64_t)j2 * k1) / (double)N, &s, &c);
        buf[lidx(k1)] = cmul(buf[lidx(k1)], make_float2((float)c, (float)s));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int j = i % p.ct, k1 = i / p.ct;
        zo[(int64_t)k1 * p.n2 + j20 + j] = lds[j * CHS + lidx(k1)];
    }
}

struct RowsArgs {
    const float2* zin;  // [batch][N1][N2]
    float2* zout;       // [batch][N] natural order
    int64_t n_total;
    int n1, ct;
    const float2* tw;  // length-N2 table
};

template <int N2>
__global__ __launch_bounds__(1024) void k_big_rows(RowsArgs p) {
    using C = Cfg<N2>;
    constexpr int CHS = ch_stride<N2>();
    extern __shared__ __align__(16) float2 lds[];
    const int team = threadIdx.x / C::NT, tid = threadIdx.x % C::NT;
    const int k10 = blockIdx.x * p.ct;
    const int64_t bt = blockIdx.y;
    const float2* zi = p.zin + bt * p.n_total + (int64_t)(k10 + team) * N2;
    float2* zo = p.zout + bt * p.n_total;
    float2* buf = lds + team * CHS;
    float2 v[C::VMAX];
    for_each_reg<N2, C::R1>(tid, [&](int idx, int n) { v[idx] = zi[n]; });
    fft<N2, false, true, false>(v, buf, p.tw, tid);
    __syncthreads();  // all teams' spectra complete before the cooperative transposed store
    const int total = N2 * p.ct;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int j = i % p.ct, k2 = i / p.ct;
        zo[(int64_t)k2 * p.n1 + k10 + j] = lds[j * CHS + lidx(k2)];
    }
}

// spec[k*n_ch + c] = scale * (spectrum of channel c), k <= N/2, from packed pair spectra
struct UnpackArgs {
    const float2* z;  // [batch = items * pairs][N]
    int64_t n_total;
    int n_ch;
    float scale;
    float2* spec;  // spec[k*bin_stride + c*ch_stride]; [N/2+1][n_ch] = (n_ch, 1)
    int64_t bin_stride, ch_stride;
};
__global__ void k_big_unpack(UnpackArgs p) {
    const int64_t N = p.n_total, nb = N / 2 + 1;
    const int pair = blockIdx.y;
    const float2* z = p.z + (int64_t)pair * N;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += (int64_t)gridDim.x * blockDim.x) {
        float2 P = z[k], Qc = z[(N - k) & (N - 1)];
        float2 A = make_float2(0.5f * (P.x + Qc.x) * p.scale, 0.5f * (P.y - Qc.y) * p.scale);
        float2 B = make_float2(0.5f * (P.y + Qc.y) * p.scale, -0.5f * (P.x - Qc.x) * p.scale);
        const int ca = 2 * pair;
        p.spec[k * p.bin_stride + ca * p.ch_stride] = A;
        if (ca + 1 < p.n_ch) p.spec[k * p.bin_stride + (ca + 1) * p.ch_stride] = B;
    }
}

// in place: z <- conj( A Ra + i B Rb ) with Hermitian completion, ready for a FORWARD
// transform that realises the inverse one (ifft(V) = conj(fft(conj V)) / N)
struct MulArgs {
    float2* z;  // [batch][N]
    int64_t n_total;
    int n_ch, r_per_channel;
    const float2* r;  // [n_ch or 1][N/2+1]
};
__global__ void k_big_mul(MulArgs p) {
    const int64_t N = p.n_total, nb = N / 2 + 1;
    const int64_t bt = blockIdx.y;
    const int npair = (p.n_ch + 1) / 2;
    const int ca = 2 * (int)(bt % npair), cb = (ca + 1 < p.n_ch) ? ca + 1 : ca;
    float2* z = p.z + bt * N;
    const float2* Ra = p.r + (p.r_per_channel ? (int64_t)ca * nb : 0);
    const float2* Rb = p.r + (p.r_per_channel ? (int64_t)cb * nb : 0);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += (int64_t)gridDim.x * blockDim.x) {
        float2 P = z[k], Qc = z[(N - k) & (N - 1)];
        float2 A = make_float2(0.5f * (P.x + Qc.x), 0.5f * (P.y - Qc.y));
        float2 B = make_float2(0.5f * (P.y + Qc.y), -0.5f * (P.x - Qc.x));
        float2 VA = cmul(A, Ra[k]), VB = cmul(B, Rb[k]);
        if (k == 0 || k == N / 2) {
            z[k] = make_float2(VA.x, -VB.x);  // conj(VA.x + i VB.x)
        } else {
            // V[k] = VA + i VB, V[N-k] = conj(VA) + i conj(VB); store the conjugates
            z[k] = make_float2(VA.x - VB.y, -(VA.y + VB.x));
            z[N - k] = make_float2(VA.x + VB.y, -(VB.x - VA.y));
        }
    }
}

// ir_a[n] = Re F[n] / N, ir_b[n] = -Im F[n] / N
struct StoreArgs {
    const float2* z;
    int64_t n_total, n_out, ld_out;
    int n_ch;
    float* ir;  // [(item*n_ch + c)*ld_out + n]
    int64_t z_off;  // first transform sample that is stored (overlap-save: n_taps - 1)
};
__global__ void k_big_store(StoreArgs p) {
    const int64_t bt = blockIdx.y;
    const int npair = (p.n_ch + 1) / 2;
    const int ca = 2 * (int)(bt % npair);
    const int64_t item = bt / npair;
    const float2* z = p.z + bt * p.n_total;
    float* oa = p.ir + (item * p.n_ch + ca) * p.ld_out;
    const bool vb = ca + 1 < p.n_ch;
    const float inv = 1.0f / (float)p.n_total;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < p.n_out; n += (int64_t)gridDim.x * blockDim.x) {
        float2 f = z[n + p.z_off];
        oa[n] = f.x * inv;
        if (vb) oa[p.ld_out + n] = -f.y * inv;
    }
}

// ---- framed transforms (STFT / Welch with window or FFT lengths beyond the LDS-resident FFT) ----
// mean of the windowed frame (the reference detrends AFTER windowing, over all W samples)
struct FrameMeansArgs {
    const float* x;
    int64_t n_samples, ld, pad_front;
    int n_ch, W, hop, n_frames;
    const float* window;
    float* means;  // [n_ch][n_frames]
};
__global__ __launch_bounds__(256) void k_frame_means(FrameMeansArgs p) {
    __shared__ double red[4];
    const int f = blockIdx.x, c = blockIdx.y;
    const float* xc = p.x + (int64_t)c * p.ld;
    const int64_t s0 = (int64_t)f * p.hop - p.pad_front;
    double acc = 0.0;
    for (int n = threadIdx.x; n < p.W; n += 256) {
        const int64_t s = s0 + n;
        if (s >= 0 && s < p.n_samples) acc += (double)(xc[s] * p.window[n]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        p.means[(int64_t)c * p.n_frames + f] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)p.W);
}

// packed frame-pair spectra z[bt][N] -> layout 0: spec[(c*F + f)*nb + k] (Welch accumulation /
// median), layout 1: out[(k*F + f)*C + c] (the reference's STFT order) with the STFT scalings
struct UnpackFramesArgs {
    const float2* z;
    int64_t n_total, bt0;
    int n_ch, n_frames, layout, power;
    float scale, edge_scale;
    float2* out;
};
__global__ void k_big_unpack_frames(UnpackFramesArgs p) {
    const int64_t N = p.n_total, nb = N / 2 + 1;
    const int64_t g = p.bt0 + blockIdx.y;
    const int nfp = (p.n_frames + 1) / 2;
    const int ch = (int)(g / nfp), f0 = 2 * (int)(g % nfp);
    const bool v1 = f0 + 1 < p.n_frames;
    const float2* z = p.z + (int64_t)blockIdx.y * N;
    const int64_t F = p.n_frames, C = p.n_ch;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += (int64_t)gridDim.x * blockDim.x) {
        float2 P = z[k], Qc = z[(N - k) & (N - 1)];
        float2 A = make_float2(0.5f * (P.x + Qc.x), 0.5f * (P.y - Qc.y));
        float2 B = make_float2(0.5f * (P.y + Qc.y), -0.5f * (P.x - Qc.x));
        const bool edge = (k == 0 || k == N / 2);
        if (p.power) {
            float e = (edge ? p.edge_scale * p.edge_scale : 1.0f) * p.scale;
            A = make_float2((A.x * A.x + A.y * A.y) * e, 0.f);
            B = make_float2((B.x * B.x + B.y * B.y) * e, 0.f);
        } else {
            float sc = p.scale * (edge ? p.edge_scale : 1.0f);
            A = make_float2(A.x * sc, A.y * sc);
            B = make_float2(B.x * sc, B.y * sc);
        }
        if (p.layout == 0) {
            p.out[((int64_t)ch * F + f0) * nb + k] = A;
            if (v1) p.out[((int64_t)ch * F + f0 + 1) * nb + k] = B;
        } else {
            p.out[(k * F + f0) * C + ch] = A;
            if (v1) p.out[(k * F + f0 + 1) * C + ch] = B;
        }
    }
}

// frame sums of |X|^2, conj(X) Y, |Y|^2 from stored spectra [c][F][nb] (fp64 accumulation);
// outputs in the one-chunk slab layout k_welch_finish reads.  kind as in WelchFinArgs.
struct SpecSumArgs {
    const float2* xs;
    const float2* ys;
    int n_cx, n_cy, n_frames, nb, kind;
    float* pxx;   // [n_cx][nb]
    float2* pxy;  // [n_cy][nb]
    float* pyy;   // [n_cy][nb]
};
__global__ void k_spec_sum(SpecSumArgs p) {
    const int nb = p.nb, F = p.n_frames;
    const int c = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int cx = (p.kind == 1 || p.n_cx != 1) ? c : 0;
    const float2* X = p.xs + (size_t)cx * F * nb + b;
    double sxx = 0.0;
    if (p.kind == 1) {
        for (int f = 0; f < F; ++f) {
            float2 x = X[(size_t)f * nb];
            sxx += (double)x.x * x.x + (double)x.y * x.y;
        }
        p.pxx[(size_t)c * nb + b] = (float)sxx;
        return;
    }
    const float2* Y = p.ys + (size_t)c * F * nb + b;
    double syy = 0.0, sr = 0.0, si = 0.0;
    for (int f = 0; f < F; ++f) {
        float2 x = X[(size_t)f * nb], y = Y[(size_t)f * nb];
        sxx += (double)x.x * x.x + (double)x.y * x.y;
        syy += (double)y.x * y.x + (double)y.y * y.y;
        sr += (double)x.x * y.x + (double)x.y * y.y;
        si += (double)x.x * y.y - (double)x.y * y.x;
    }
    if (p.kind == 0) {
        if (p.n_cx != 1 || c == 0) p.pxx[(size_t)cx * nb + b] = (float)sxx;
        p.pyy[(size_t)c * nb + b] = (float)syy;
    }
    p.pxy[(size_t)c * nb + b] = make_float2((float)sr, (float)si);
}

// FIR bank on the four-step FFT: z[(kf*npair + pair)] <- conj( A R_kf + i B R_kf ) with A, B the
// two channel spectra packed in spec[pair] and R_kf the (real-signal) tap spectrum of filter kf
struct MulBankArgs {
    const float2* spec;  // [npair][N]
    const float2* r;     // [filters][N/2+1]
    float2* z;           // [filters*npair][N]
    int64_t n_total;
    int npair;
};
__global__ void k_big_mul_bank(MulBankArgs p) {
    const int64_t N = p.n_total, nb = N / 2 + 1;
    const int64_t bt = blockIdx.y;
    const float2* zi = p.spec + (bt % p.npair) * N;
    const float2* R = p.r + (bt / p.npair) * nb;
    float2* z = p.z + bt * N;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += (int64_t)gridDim.x * blockDim.x) {
        float2 P = zi[k], Qc = zi[(N - k) & (N - 1)];
        float2 A = make_float2(0.5f * (P.x + Qc.x), 0.5f * (P.y - Qc.y));
        float2 B = make_float2(0.5f * (P.y + Qc.y), -0.5f * (P.x - Qc.x));
        float2 VA = cmul(A, R[k]), VB = cmul(B, R[k]);
        if (k == 0 || k == N / 2) {
            z[k] = make_float2(VA.x, -VB.x);
        } else {
            z[k] = make_float2(VA.x - VB.y, -(VA.y + VB.x));
            z[N - k] = make_float2(VA.x + VB.y, -(VB.x - VA.y));
        }
    }
}

}  // namespace dsbig
