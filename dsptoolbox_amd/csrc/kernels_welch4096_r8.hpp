// Welch H1/H2/H3 for the headline shape (nfft 4096, one input channel): radix-8 variant of
// kernels_welch4096.hpp.  Same data flow and algebra (k_x -> xs / px, k_y -> T, P per chunk,
// fold k <-> N-k once per chunk); the transform is laid out for FOUR waves per SIMD:
//
//   4096 = 8 x 8 x 8 x 8, 512 threads, 8 complex values per thread (<= 128 VGPRs, two
//   workgroups = 16 waves per CU).  A wave issues at most one VALU instruction every 4 cycles
//   while the SIMD can take one every 2: with the radix-16 kernel's two waves per SIMD the
//   vector pipe idles whenever one of them waits on LDS, a barrier or memory (measured: 38 %
//   VALU utilisation); four waves per SIMD keep it fed.
//
//   n = 512 n1 + 64 n2 + 8 n3 + n4 ,  k = k1 + 8 k2 + 64 k3 + 512 k4
//   pass 1  thread t = 64 n2 + 8 n3 + n4 : DFT8 over n1 of z[t + 512 n1] (straight from HBM),
//           times W4096^(t k1) (7 per-thread constants in registers)  -> LDS [k1][t]
//   pass 2  thread u = 64 k1 + r, r = 8 n3 + n4 : DFT8 over n2, times W512^(r k2) (LDS table)
//           -> LDS [k1][k2][r], k2 stride 72
//   pass 3  thread w = 64 k1 + 8 k2 + n4 : DFT8 over n3, times W64^(n4 k3) (LDS table)
//           -> LDS rows v = k1 + 8 k2 + 64 k3 of 8 complex (n4), row stride 9
//   pass 4  thread v : DFT8 over n4 -> Z[v + 512 k4] in registers
//   Every LDS access is a conflict-free ds_{read,write}_b64 (checked exhaustively for the
//   strides above).  Two exchange buffers used alternately (A B A | B A B | ...): three
//   barriers per transform, no write-after-read hazard across iterations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <vector>

#include "kernels_welch4096.hpp"

namespace welch4096r8 {

using welch4096::Args;

constexpr int N = 4096, NT = 512, NB = N / 2 + 1;
constexpr int S1 = 512, R2S = 72, R1S = 8 * R2S, RS = 9;
constexpr int BUF_C = 4608;  // >= 8*512, 8*576, 512*9 complex
constexpr int TW2_LEN = 7 * 64, TW3_LEN = 7 * 8;
constexpr int LDS_BYTES = (2 * BUF_C + TW2_LEN + TW3_LEN) * 8;  // 77 760 B: two workgroups per CU
constexpr int TWT_LEN = 7 * 512 + TW2_LEN + TW3_LEN;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 sub_i(float2 a, float2 b) { return make_float2(a.x + b.y, a.y - b.x); }  // a - i b
__device__ __forceinline__ float2 add_i(float2 a, float2 b) { return make_float2(a.x - b.y, a.y + b.x); }  // a + i b

// 8-point DFT in registers, natural order in and out: X[k] = E[k mod 4] + W8^k O[k mod 4]
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    constexpr float R2 = 0.70710678118654752440f;
    const float2 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]), a2 = cadd(v[2], v[6]), a3 = csub(v[2], v[6]);
    const float2 a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]), a6 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    const float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = sub_i(a1, a3), b3 = add_i(a1, a3);
    const float2 b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = sub_i(a5, a7), b7 = add_i(a5, a7);
    v[0] = cadd(b0, b4);
    v[4] = csub(b0, b4);
    v[2] = sub_i(b2, b6);
    v[6] = add_i(b2, b6);
    {  // W8 b5 = ((x + y) + i (y - x)) / sqrt 2
        const float p = b5.x + b5.y, q = b5.y - b5.x;
        v[1] = make_float2(fmaf(p, R2, b1.x), fmaf(q, R2, b1.y));
        v[5] = make_float2(fmaf(-p, R2, b1.x), fmaf(-q, R2, b1.y));
    }
    {  // W8^3 b7 = ((y - x) - i (x + y)) / sqrt 2
        const float p = b7.y - b7.x, q = b7.x + b7.y;
        v[3] = make_float2(fmaf(p, R2, b3.x), fmaf(-q, R2, b3.y));
        v[7] = make_float2(fmaf(-p, R2, b3.x), fmaf(q, R2, b3.y));
    }
}

// v[n1] = z[tid + 512 n1] on entry, v[k4] = Z[tid + 512 k4] on return.  bufA / bufB: the two
// exchange buffers in this iteration's order (the caller swaps them every transform).
__device__ __forceinline__ void fft4096(float2 (&v)[8], const float2 (&tw1)[7], float2* __restrict__ bufA,
                                        float2* __restrict__ bufB, const float2* __restrict__ tw2,
                                        const float2* __restrict__ tw3, int tid) {
    dft8(v);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) v[k1] = cmul(v[k1], tw1[k1 - 1]);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) bufA[k1 * S1 + tid] = v[k1];
    const int k1u = tid >> 6, r = tid & 63;
    // the next pass' twiddles do not depend on the exchange: fetch them (into the registers the
    // stored values just left) before the barrier, so their latency hides behind it
    float2 w[7];
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) w[k2 - 1] = tw2[(k2 - 1) * 64 + r];
    __syncthreads();
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) v[n2] = bufA[k1u * S1 + 64 * n2 + r];
    dft8(v);
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) v[k2] = cmul(v[k2], w[k2 - 1]);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) bufB[k1u * R1S + k2 * R2S + r] = v[k2];
    const int k2w = (tid >> 3) & 7, n4 = tid & 7;
#pragma unroll
    for (int k3 = 1; k3 < 8; ++k3) w[k3 - 1] = tw3[(k3 - 1) * 8 + n4];
    __syncthreads();
    const int base3 = k1u * R1S + k2w * R2S + n4;
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) v[n3] = bufB[base3 + 8 * n3];
    dft8(v);
#pragma unroll
    for (int k3 = 1; k3 < 8; ++k3) v[k3] = cmul(v[k3], w[k3 - 1]);
    const int row0 = k1u + 8 * k2w;
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) bufA[(row0 + 64 * k3) * RS + n4] = v[k3];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bufA[tid * RS + j];
    dft8(v);
}

// twt: fp64-computed tables: [7][512] W4096^(t k1), [7][64] W512^(r k2), [7][8] W64^(n4 k3), k = 1..7
inline void host_tables(std::vector<float2>& t) {
    t.resize(TWT_LEN);
    auto w = [](int n, int m) {
        double a = -2.0 * M_PI * (double)m / (double)n;
        return make_float2((float)std::cos(a), (float)std::sin(a));
    };
    for (int k = 1; k < 8; ++k) {
        for (int tt = 0; tt < 512; ++tt) t[(k - 1) * 512 + tt] = w(4096, tt * k);
        for (int r = 0; r < 64; ++r) t[7 * 512 + (k - 1) * 64 + r] = w(512, r * k);
        for (int n4 = 0; n4 < 8; ++n4) t[7 * 512 + TW2_LEN + (k - 1) * 8 + n4] = w(64, n4 * k);
    }
}

__device__ __forceinline__ void init_tables(float2 (&tw1)[7], float (&win)[8], float2* tw2, float2* tw3,
                                            const float* __restrict__ window,
                                            const float2* __restrict__ twt, int tid) {
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) tw1[k1 - 1] = twt[(k1 - 1) * 512 + tid];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) win[n1] = window[tid + 512 * n1];
    if (tid < TW2_LEN) tw2[tid] = twt[7 * 512 + tid];
    if (tid < TW3_LEN) tw3[tid] = twt[7 * 512 + TW2_LEN + tid];
    __syncthreads();  // the tables are read before the first exchange barrier of the transform
}

// Raw samples of the frame pair (2p, 2p+1) of one channel; HALF_HOP (hop == 2048): the frames
// share half their samples -> 12 loads s[m] = ch[start + tid + 512 m], otherwise 16.  Interior
// pairs: unconditional loads behind one wave-uniform test; ragged tail: clamp + select.
template <bool HALF_HOP>
struct Raw {
    float s[HALF_HOP ? 12 : 16];
};

template <bool HALF_HOP>
__device__ __forceinline__ void load_raw(Raw<HALF_HOP>& r, const float* __restrict__ ch,
                                         int64_t n_samples, int64_t start0, int hop, int tid) {
    const float* __restrict__ src = ch + start0;
    const int64_t remain = n_samples - start0;
    const int span = HALF_HOP ? 3 * 2048 : hop + 4096;
    if (remain >= span) {
        if (HALF_HOP) {
#pragma unroll
            for (int m = 0; m < 12; ++m) r.s[m] = src[tid + 512 * m];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                r.s[n1] = src[tid + 512 * n1];
                r.s[8 + n1] = src[hop + tid + 512 * n1];
            }
        }
    } else {
        const int last = (int)(remain > (int64_t)(1 << 30) ? (1 << 30) : remain) - 1;
        constexpr int CNT = HALF_HOP ? 12 : 16;
#pragma unroll
        for (int m = 0; m < CNT; ++m) {
            int i = HALF_HOP ? tid + 512 * m : (m < 8 ? tid + 512 * m : hop + tid + 512 * (m - 8));
            float a = src[min(i, last)];
            r.s[m] = i <= last ? a : 0.f;
        }
    }
}

template <bool HALF_HOP>
__device__ __forceinline__ void window_pair(float2 (&v)[8], const Raw<HALF_HOP>& r, bool second,
                                            const float (&win)[8]) {
    const float m2 = second ? 1.f : 0.f;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        float b = HALF_HOP ? r.s[n1 + 4] : r.s[8 + n1];
        v[n1] = make_float2(r.s[n1] * win[n1], b * (win[n1] * m2));
    }
}

// ---- input spectra: xs[pair][4][512] float4 = bins (tid + 512*2g, tid + 512*(2g+1)) of thread tid
template <bool HALF_HOP>
__global__ __launch_bounds__(NT) void k_x(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* bufA = lds;
    float2* bufB = lds + BUF_C;
    float2* tw2 = lds + 2 * BUF_C;
    float2* tw3 = tw2 + TW2_LEN;
    const int tid = threadIdx.x, pr = blockIdx.x;
    float2 tw1[7];
    float win[8];
    init_tables(tw1, win, tw2, tw3, p.window, p.twt, tid);
    float2 v[8];
    {
        Raw<HALF_HOP> raw;
        load_raw<HALF_HOP>(raw, p.sig, p.n_samples, (int64_t)(2 * pr) * p.hop, p.hop, tid);
        window_pair<HALF_HOP>(v, raw, 2 * pr + 1 < p.n_frames, win);
    }
    fft4096(v, tw1, bufA, bufB, tw2, tw3, tid);
    if (p.detrend && tid == 0) v[0] = make_float2(0.f, 0.f);
    float4* xo = reinterpret_cast<float4*>(p.xs + (int64_t)pr * N) + tid;
#pragma unroll
    for (int g = 0; g < 4; ++g) xo[512 * g] = make_float4(v[2 * g].x, v[2 * g].y, v[2 * g + 1].x, v[2 * g + 1].y);
    float* pw = reinterpret_cast<float*>(bufB);  // bufA may still be read by slow waves
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) pw[tid + 512 * k4] = v[k4].x * v[k4].x + v[k4].y * v[k4].y;
    __syncthreads();
    float* po = p.px + (int64_t)pr * NB;
    for (int k = tid; k < NB; k += NT) po[k] = 0.5f * (pw[k] + pw[(N - k) & (N - 1)]);
}

// ---- output channels: workgroup = (channel c, chunk q of frame pairs)
// (the second __launch_bounds__ argument is waves per SIMD: 4 -> <= 128 VGPRs, two workgroups per CU)
template <bool HALF_HOP>
__global__ __launch_bounds__(NT, 4) void k_y(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* tw2 = lds + 2 * BUF_C;
    float2* tw3 = tw2 + TW2_LEN;
    const int tid = threadIdx.x;
    int q, c;  // XCD-aware decode, as in welch4096::k_y
    {
        const int b = blockIdx.x;
        if ((p.n_chunks & 7) == 0) {
            const int per = p.n_chunks >> 3;
            q = (b & 7) + 8 * ((b >> 3) % per);
            c = (b >> 3) / per;
        } else {
            q = b % p.n_chunks;
            c = b / p.n_chunks;
        }
    }
    float2 tw1[7];
    float win[8];
    init_tables(tw1, win, tw2, tw3, p.window, p.twt, tid);
    const float* ch = p.sig + (int64_t)c * p.ld;
    const int p0 = q * p.ppc, p1 = min(p0 + p.ppc, p.n_pairs);
    float2 T[8];
    float P[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        T[j] = make_float2(0.f, 0.f);
        P[j] = 0.f;
    }
    Raw<HALF_HOP> raw;
    if (p0 < p1) load_raw<HALF_HOP>(raw, ch, p.n_samples, (int64_t)(2 * p0) * p.hop, p.hop, tid);
    int flip = 0;
    for (int pr = p0; pr < p1; ++pr) {
        float2 v[8];
        window_pair<HALF_HOP>(v, raw, 2 * pr + 1 < p.n_frames, win);
        if (pr + 1 < p1) load_raw<HALF_HOP>(raw, ch, p.n_samples, (int64_t)(2 * pr + 2) * p.hop, p.hop, tid);
        float2 xw[8];
        {
            const float4* __restrict__ xp = reinterpret_cast<const float4*>(p.xs + (int64_t)pr * N) + tid;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t = xp[512 * g];
                xw[2 * g] = make_float2(t.x, t.y);
                xw[2 * g + 1] = make_float2(t.z, t.w);
            }
        }
        fft4096(v, tw1, lds + flip, lds + (BUF_C - flip), tw2, tw3, tid);
        flip = BUF_C - flip;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const float2 w = xw[k4], z = v[k4];
            T[k4].x = fmaf(w.x, z.x, fmaf(w.y, z.y, T[k4].x));  // conj(w) z
            T[k4].y = fmaf(w.x, z.y, fmaf(-w.y, z.x, T[k4].y));
            P[k4] = fmaf(z.x, z.x, fmaf(z.y, z.y, P[k4]));
        }
    }
    if (p.detrend && tid == 0) P[0] = 0.f;  // xs bin 0 is already 0 -> T[0] = 0
    // fold k <-> N-k once per chunk, through LDS
    float2* buf = lds;
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) buf[tid + 512 * k4] = T[k4];
    __syncthreads();
    const int64_t so = ((int64_t)q * p.n_ch + c) * NB;
    for (int k = tid; k < NB; k += NT) {
        float2 a = buf[k], b = buf[(N - k) & (N - 1)];
        p.pxy[so + k] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    }
    __syncthreads();
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) pw[tid + 512 * k4] = P[k4];
    __syncthreads();
    for (int k = tid; k < NB; k += NT) p.pyy[so + k] = 0.5f * (pw[k] + pw[(N - k) & (N - 1)]);
}

}  // namespace welch4096r8
