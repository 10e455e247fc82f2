// Bluestein (chirp-z) DFT of arbitrary length L on top of the power-of-two four-step FFT:
//   X[k] = w[k] * sum_n (z[n] w[n]) conj(w[k-n]),  w[n] = exp(-i pi n^2 / L)
// = a circular convolution of length M >= 2L-1 (M = 2^m):  X = w . ifft_M( fft_M(z w) . fft_M(b) ),
// b[n] = conj(w[n]) wrapped to [0, M).  n^2 mod 2L is taken in 64-bit integers and the phase in
// fp64, so the chirps are exact to fp32 rounding for any L.  Used for the whole-signal spectra
// and deconvolution at the reference's next_fast_len lengths (e.g. 192 000 = 2^9 3 5^3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsblue {

__device__ __forceinline__ float2 chirp(int64_t n, int64_t L) {
    int64_t q = (n * n) % (2 * L);
    double s, c;
    sincospi(-(double)q / (double)L, &s, &c);
    return make_float2((float)c, (float)s);
}
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// bf[m] = conj(w[n]) at m = n and m = M - n (n < L), 0 elsewhere  (time domain; FFT it once)
__global__ void k_filter(float2* bf, int64_t L, int64_t M) {
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = m < L ? m : ((M - m) < L ? (M - m) : -1);
        float2 v = make_float2(0.f, 0.f);
        if (n >= 0) {
            float2 w = chirp(n, L);
            v = make_float2(w.x, -w.y);
        }
        bf[m] = v;
    }
}

// a[n] = z[n] w[n] (n < L), 0 up to M.  Source: real channel pair (xa + i xb, zero beyond
// n_samples) or a complex buffer of length L.   grid.y = batch
struct PreArgs {
    const float* xreal;  // planar, or nullptr
    int64_t ld_real, n_samples;
    int n_ch;
    const float2* zin;  // [batch][L] when xreal == nullptr
    float2* a;          // [batch][M]
    int64_t L, M;
};
__global__ void k_pre(PreArgs p) {
    const int64_t bt = blockIdx.y;
    const float* xa = nullptr;
    const float* xb = nullptr;
    if (p.xreal) {
        const int npair = (p.n_ch + 1) / 2;
        const int ca = 2 * (int)(bt % npair);
        const int64_t item = bt / npair;
        xa = p.xreal + (item * p.n_ch + ca) * p.ld_real;
        xb = (ca + 1 < p.n_ch) ? xa + p.ld_real : nullptr;
    }
    float2* a = p.a + bt * p.M;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < p.M; n += (int64_t)gridDim.x * blockDim.x) {
        float2 v = make_float2(0.f, 0.f);
        if (n < p.L) {
            float2 z;
            if (p.xreal) {
                z.x = n < p.n_samples ? xa[n] : 0.f;
                z.y = (xb && n < p.n_samples) ? xb[n] : 0.f;
            } else {
                z = p.zin[bt * p.L + n];
            }
            v = cmulf(z, chirp(n, p.L));
        }
        a[n] = v;
    }
}

// q <- conj(q * bf): the following FORWARD transform then realises the inverse one
__global__ void k_mul_filter(float2* q, const float2* bf, int64_t M) {
    float2* qq = q + (int64_t)blockIdx.y * M;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < M; k += (int64_t)gridDim.x * blockDim.x) {
        float2 v = cmulf(qq[k], bf[k]);
        qq[k] = make_float2(v.x, -v.y);
    }
}

// x[k] = w[k] conj(c[k]) / M, k < L
__global__ void k_post(const float2* c, float2* x, int64_t L, int64_t M) {
    const float2* cc = c + (int64_t)blockIdx.y * M;
    float2* xx = x + (int64_t)blockIdx.y * L;
    const float inv = 1.0f / (float)M;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < L; k += (int64_t)gridDim.x * blockDim.x) {
        float2 v = cc[k];
        float2 r = cmulf(make_float2(v.x * inv, -v.y * inv), chirp(k, L));
        xx[k] = r;
    }
}

// ---- real-signal helpers for an arbitrary length L (spectrum x[batch][L] of packed pairs) ----
// spec[k*n_ch + c] = scale * X_c[k], k <= L/2
__global__ void k_unpack(const float2* x, int64_t L, int n_ch, float scale, float2* spec) {
    const int pair = blockIdx.y;
    const float2* z = x + (int64_t)pair * L;
    const int64_t nb = L / 2 + 1;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += (int64_t)gridDim.x * blockDim.x) {
        float2 P = z[k], Qc = z[(L - k) % L];
        float2 A = make_float2(0.5f * (P.x + Qc.x) * scale, 0.5f * (P.y - Qc.y) * scale);
        float2 B = make_float2(0.5f * (P.y + Qc.y) * scale, -0.5f * (P.x - Qc.x) * scale);
        const int ca = 2 * pair;
        spec[k * n_ch + ca] = A;
        if (ca + 1 < n_ch) spec[k * n_ch + ca + 1] = B;
    }
}

// in place on x[batch][L]: x <- conj( A Ra + i B Rb ) with Hermitian completion
__global__ void k_mul_r(float2* x, int64_t L, int n_ch, int r_per_channel, const float2* r) {
    const int64_t bt = blockIdx.y;
    const int npair = (n_ch + 1) / 2;
    const int ca = 2 * (int)(bt % npair), cb = (ca + 1 < n_ch) ? ca + 1 : ca;
    float2* z = x + bt * L;
    const int64_t nb = L / 2 + 1;
    const float2* Ra = r + (r_per_channel ? (int64_t)ca * nb : 0);
    const float2* Rb = r + (r_per_channel ? (int64_t)cb * nb : 0);
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t km = (L - k) % L;
        float2 P = z[k], Qc = z[km];
        float2 A = make_float2(0.5f * (P.x + Qc.x), 0.5f * (P.y - Qc.y));
        float2 B = make_float2(0.5f * (P.y + Qc.y), -0.5f * (P.x - Qc.x));
        float2 VA = cmulf(A, Ra[k]), VB = cmulf(B, Rb[k]);
        if (km == k) {  // DC, and Nyquist for even L: irfft ignores the imaginary part
            z[k] = make_float2(VA.x, -VB.x);
        } else {
            z[k] = make_float2(VA.x - VB.y, -(VA.y + VB.x));
            z[km] = make_float2(VA.x + VB.y, -(VB.x - VA.y));
        }
    }
}

// ir_a[n] = Re X[n] / L, ir_b[n] = -Im X[n] / L   (X = DFT_L of the conjugated product)
__global__ void k_store(const float2* x, int64_t L, int64_t n_out, int64_t ld_out, int n_ch, float* ir) {
    const int64_t bt = blockIdx.y;
    const int npair = (n_ch + 1) / 2;
    const int ca = 2 * (int)(bt % npair);
    const int64_t item = bt / npair;
    const float2* z = x + bt * L;
    float* oa = ir + (item * n_ch + ca) * ld_out;
    const bool vb = ca + 1 < n_ch;
    const float inv = 1.0f / (float)L;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < n_out; n += (int64_t)gridDim.x * blockDim.x) {
        float2 f = z[n];
        oa[n] = f.x * inv;
        if (vb) oa[ld_out + n] = -f.y * inv;
    }
}

}  // namespace dsblue
