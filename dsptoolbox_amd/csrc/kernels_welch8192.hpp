// Welch H1/H2/H3 with an 8192-sample window and ONE input channel: the algebra and pipeline of
// kernels_welch4096.hpp on two of its 4096-point register transforms per frame pair.
//
//   8192 = 2 x 4096, 512 threads = 2 groups of 256; group q owns the sub-spectrum Z[2k' + q]
//   (radix-2 decimation in frequency in front of the transform, as kernels_deconv8k.hpp):
//       b_q[n'] = (z[n'] + (-1)^q z[n' + 4096]) W8192^(n' q) ,   Z[2k' + q] = FFT4096(b_q)[k']
//   z = frame_2p w + i frame_2p+1 w.  T[k] += conj(W[k]) Z[k], P[k] += |Z[k]|^2 over the 4096 bins
//   of the class, 16 per thread; the fold k <-> N-k stays inside a class:
//       N - 2k'     = 2 (4096 - k')         (class 0)
//       N - (2k'+1) = 2 (4095 - k') + 1     (class 1)
//   With hop = 4096 the second half of frame 2p IS the first half of frame 2p+1: 48 sample loads
//   per thread and pair (three 4096-sample segments) instead of 64.  150 KB of LDS (two exchange
//   buffers per group): one workgroup per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_deconv8k.hpp"
#include "kernels_welch4096.hpp"

namespace welch8k {

namespace w4 = welch4096;
using w4::cmul;
using w4::pos16;
constexpr int N = 8192, M = 4096, NB = N / 2 + 1, NTB = 512;
constexpr int LDS_BYTES = (4 * w4::BUF_C + 256) * 8;  // 149 504 B
constexpr int LDS_BYTES_WINLDS = (2 * w4::BUF_C + 256) * 8 + N * 4;  // 108 544 B

struct Args {
    const float* sig;  // x (k_x) or y (k_y), planar
    int64_t n_samples, ld;
    int n_ch, hop, n_frames, n_pairs, detrend;
    int n_chunks;
    const float* window;
    const float2* twt;  // welch4096::host_tables()
    const float2* twn;  // deconv8k::host_tables(): [256] W8192^t, [16] W32^n1
    float4* xs;         // [n_pairs][2][8][256]: class q, thread t holds bins (2(t + 256*2g) + q, 2(t + 256*(2g+1)) + q)
    float* px;          // [n_pairs][NB]
    float2* pxy;        // [n_chunks][n_ch][NB]
    float* pyy;         // [n_chunks][n_ch][NB]
    float* psx;         // [n_chunks][n_cx][NB]
    int n_cx;           // input channels: 1 (shared) or n_ch (one per output channel; then xs is
                        // [n_cx][n_pairs][2][8][256], px [n_cx][n_pairs][NB] and k_px_sum fills psx)
};

// Raw samples of the frame pair (2p, 2p+1): segment j, slot n1 = ch[start0 + off_j + t + 256 n1].
// HALF_HOP (hop == 4096): offsets 0, 4096, 8192 (frame a = segments 0,1; frame b = segments 1,2);
// otherwise 0, 4096, hop, hop + 4096 (a lo, a hi, b lo, b hi).
template <bool HALF_HOP>
struct Raw {
    float s[HALF_HOP ? 48 : 64];
};
template <bool HALF_HOP>
__device__ __forceinline__ void load_raw(Raw<HALF_HOP>& r, const float* __restrict__ ch, int64_t n_samples,
                                         int64_t start0, int hop, int t) {
    const float* __restrict__ src = ch + start0;
    const int64_t remain = n_samples - start0;  // >= 1 for every valid pair
    const int span = HALF_HOP ? 3 * M : hop + N;
    constexpr int SEG = HALF_HOP ? 3 : 4;
    if (remain >= span) {
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
            const int off = HALF_HOP ? M * j : (j < 2 ? M * j : hop + M * (j - 2));
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) r.s[16 * j + n1] = src[off + t + 256 * n1];
        }
    } else {
        const int last = (int)(remain > (int64_t)(1 << 30) ? (1 << 30) : remain) - 1;
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
            const int off = HALF_HOP ? M * j : (j < 2 ? M * j : hop + M * (j - 2));
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int i = off + t + 256 * n1;
                const float a = src[min(i, last)];
                r.s[16 * j + n1] = i <= last ? a : 0.f;
            }
        }
    }
}

// The same samples through a raw-buffer descriptor of the channel: 32-bit offsets, and samples at
// or past n_samples come back as 0 from the hardware range check -- no ragged-tail path, no 64-bit
// address registers (the kernel sits at the 256-register limit; they were spilled).
template <bool HALF_HOP>
__device__ __forceinline__ void load_raw_buf(Raw<HALF_HOP>& r, __amdgpu_buffer_rsrc_t rs, int64_t start0, int hop, int t) {
    constexpr int SEG = HALF_HOP ? 3 : 4;
    const uint32_t o = ((uint32_t)start0 + (uint32_t)t) * 4u;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        const uint32_t off = HALF_HOP ? (uint32_t)(M * j) : (uint32_t)(j < 2 ? M * j : hop + M * (j - 2));
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1)
            r.s[16 * j + n1] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o + 4u * (off + 256u * n1)), 0, 0));
    }
}
inline bool buf_fits(int64_t n_samples, int n_frames, int hop) {
    return n_samples < ((int64_t)1 << 30) - 4 * N && (int64_t)(n_frames + 2) * hop < ((int64_t)1 << 30) - 4 * N;
}

// window, pack the two frames and run the radix-2 front end of class q:
// v[n1] = (z[n'] + (-1)^q z[n' + 4096]) W8192^(n' q), n' = t + 256 n1
template <bool HALF_HOP>
__device__ __forceinline__ void front(float2 (&v)[16], const Raw<HALF_HOP>& r, const float* __restrict__ window,
                                      int q, float2 wt, bool drop, int t) {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const float w0 = window[t + 256 * n1], w1 = window[M + t + 256 * n1];
        const float alo = r.s[n1], ahi = r.s[16 + n1];
        const float blo = HALF_HOP ? r.s[16 + n1] : r.s[32 + n1];
        const float bhi = HALF_HOP ? r.s[32 + n1] : r.s[48 + n1];
        float2 lo = make_float2(alo * w0, blo * w0), hi = make_float2(ahi * w1, bhi * w1);
        if (drop) {
            lo.y = 0.f;
            hi.y = 0.f;
        }
        if (q == 0)
            v[n1] = make_float2(lo.x + hi.x, lo.y + hi.y);
        else
            v[n1] = cmul(make_float2(lo.x - hi.x, lo.y - hi.y), cmul(wt, deconv8k::w32(n1)));
    }
}

// the last pair of an odd frame count when frame F would still overlap the signal
__device__ __forceinline__ bool needs_drop(const Args& p, int pr) {
    return pr == p.n_pairs - 1 && (p.n_frames & 1) && (int64_t)p.n_frames * p.hop < p.n_samples;
}

// fold k <-> N - k of one class from a natural-order LDS image img[k'] (k' < 4096):
// class 0 bin 2k' (k' <= 2048) pairs with (4096 - k') & 4095, class 1 bin 2k'+1 (k' <= 2047) with 4095 - k'
__device__ __forceinline__ int fold_partner(int q, int kp) { return q == 0 ? ((M - kp) & (M - 1)) : (M - 1 - kp); }
__device__ __forceinline__ int fold_count(int q) { return q == 0 ? M / 2 + 1 : M / 2; }

// ---- input spectra: grid = n_pairs ------------------------------------------------
template <bool HALF_HOP>
__global__ __launch_bounds__(NTB) void k_x(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    const int tid = threadIdx.x, t = tid & 255;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 8);
    float2* buf = lds + q * 2 * w4::BUF_C;
    float2* tw2 = lds + 4 * w4::BUF_C;
    const int pr = blockIdx.x, cx = blockIdx.y;
    Raw<HALF_HOP> raw;
    load_raw<HALF_HOP>(raw, p.sig + (int64_t)cx * p.ld, p.n_samples, (int64_t)(2 * pr) * p.hop, p.hop, t);
    w4::Tw tw;
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw.w[k1 - 1] = p.twt[(k1 - 1) * 256 + t];
    if (tid < 256) tw2[tid] = p.twt[15 * 256 + tid];
    const float2 wt = p.twn[t];
    __syncthreads();
    float2 v[16];
    front<HALF_HOP>(v, raw, p.window, q, wt, needs_drop(p, pr), t);
    w4::fft4096_plain<true>(v, tw, buf, tw2, t);
    if (p.detrend && tid == 0) v[pos16(0)] = make_float2(0.f, 0.f);  // bin 0 = class 0, k' = 0
    float4* xo = p.xs + (((int64_t)cx * p.n_pairs + pr) * 2 + q) * (M / 2) + t;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float2 z0 = v[pos16(2 * g)], z1 = v[pos16(2 * g + 1)];
        xo[256 * g] = make_float4(z0.x, z0.y, z1.x, z1.y);
    }
    float* pw = reinterpret_cast<float*>(buf);
    __syncthreads();
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
        const float2 z = v[pos16(k3)];
        pw[t + 256 * k3] = z.x * z.x + z.y * z.y;
    }
    __syncthreads();
    float* po = p.px + ((int64_t)cx * p.n_pairs + pr) * NB;
    for (int kp = t; kp < fold_count(q); kp += 256) po[2 * kp + q] = 0.5f * (pw[kp] + pw[fold_partner(q, kp)]);
}

// ---- output channels: grid = n_chunks * n_ch ----------------------------------------
// WINLDS: the window lives in LDS (32 KB) and each group has ONE exchange buffer (two more barriers
// per transform) -- 108 KB instead of 150 KB; otherwise the window is read from global memory at
// every pair and each group has two buffers.
// AUTO: auto spectra only (ds_welch_psd): no input spectra, no cross sums, no psx.
template <bool HALF_HOP, bool WINLDS = false, bool AUTO = false>
__global__ __launch_bounds__(NTB, 1) void k_y(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    const int tid = threadIdx.x, t = tid & 255;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 8);
    constexpr int NBUF = WINLDS ? 1 : 2;
    float2* buf = lds + q * NBUF * w4::BUF_C;
    float2* tw2 = lds + 2 * NBUF * w4::BUF_C;
    float* winl = reinterpret_cast<float*>(tw2 + 256);
    if (WINLDS)
        for (int i = tid; i < N; i += NTB) winl[i] = p.window[i];
    const float* wsrc = WINLDS ? winl : p.window;
    // XCD-aware decode: whole chunks per XCD (the input spectra a chunk re-reads stay in its L2)
    int cq, c;
    {
        const int b = blockIdx.x;
        if ((p.n_chunks & 7) == 0) {
            const int per = p.n_chunks >> 3;
            cq = (b & 7) + 8 * ((b >> 3) % per);
            c = (b >> 3) / per;
        } else {
            cq = b % p.n_chunks;
            c = b / p.n_chunks;
        }
    }
    w4::Tw tw;
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw.w[k1 - 1] = p.twt[(k1 - 1) * 256 + t];
    if (tid < 256) tw2[tid] = p.twt[15 * 256 + tid];
    const float2 wt = p.twn[t];
    const float* ch = p.sig + (int64_t)c * p.ld;
    const int p0 = (int)((int64_t)cq * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(cq + 1) * p.n_pairs / p.n_chunks);
    if (!AUTO && p.n_cx <= 1) {
        // input auto-spectrum of this chunk: this workgroup's slice of the bins, px rows summed in fp64
        const int bpc = (NB + p.n_ch - 1) / p.n_ch;
        const int b0 = c * bpc, b1 = min(b0 + bpc, NB);
        for (int k = b0 + tid; k < b1; k += NTB) {
            double sum = 0.0;
            for (int pr = p0; pr < p1; ++pr) sum += (double)p.px[(int64_t)pr * NB + k];
            p.psx[(int64_t)cq * NB + k] = (float)sum;
        }
    }
    __syncthreads();  // W256 table written
    float2 T[16];
    float P[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        T[j] = make_float2(0.f, 0.f);
        P[j] = 0.f;
    }
    Raw<HALF_HOP> raw;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ch), 0, (int)(uint32_t)(p.n_samples * 4), 0x00020000);
    if (p0 < p1) load_raw_buf<HALF_HOP>(raw, rs, (int64_t)(2 * p0) * p.hop, p.hop, t);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): keep the pre-loop loads out of the loop's wait counts
    for (int pr = p0; pr < p1; ++pr) {
        float2 v[16];
        front<HALF_HOP>(v, raw, wsrc, q, wt, needs_drop(p, pr), t);
        float2 xw[16];
        auto issue_loads = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            if (pr + 1 < p1) load_raw_buf<HALF_HOP>(raw, rs, (int64_t)(2 * pr + 2) * p.hop, p.hop, t);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto issue_xs = [&]() {
            if (AUTO) return;
            __builtin_amdgcn_sched_barrier(0);
            const float4* __restrict__ xp = p.xs + (((int64_t)(p.n_cx > 1 ? c : 0) * p.n_pairs + pr) * 2 + q) * (M / 2) + t;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 r = xp[256 * g];
                xw[2 * g] = make_float2(r.x, r.y);
                xw[2 * g + 1] = make_float2(r.z, r.w);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#if W4_TIMING
        unsigned long long ph[12] = {}, prev = 0;
        w4::fft4096<!WINLDS>(v, tw, buf, tw2, t, ph, prev, issue_loads, issue_xs);
#else
        w4::fft4096<!WINLDS>(v, tw, buf, tw2, t, issue_loads, issue_xs);
#endif
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) {
            const float2 z = v[pos16(k3)];
            if (!AUTO) {
                const float2 w = xw[k3];
                T[k3].x = fmaf(w.x, z.x, fmaf(w.y, z.y, T[k3].x));  // conj(w) z
                T[k3].y = fmaf(w.x, z.y, fmaf(-w.y, z.x, T[k3].y));
            }
            P[k3] = fmaf(z.x, z.x, fmaf(z.y, z.y, P[k3]));
        }
    }
    if (p.detrend && tid == 0) P[0] = 0.f;  // class 0, k' = 0: xs bin 0 is already 0 -> T[0] = 0
    // fold k <-> N-k once per chunk, inside each class, through the group's LDS buffer
    __syncthreads();
    const int64_t so = ((int64_t)cq * p.n_ch + c) * NB;
    if (!AUTO) {
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) buf[t + 256 * k3] = T[k3];
        __syncthreads();
        for (int kp = t; kp < fold_count(q); kp += 256) {
            const float2 a = buf[kp], b = buf[fold_partner(q, kp)];
            p.pxy[so + 2 * kp + q] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
        }
        __syncthreads();
    }
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) pw[t + 256 * k3] = P[k3];
    __syncthreads();
    for (int kp = t; kp < fold_count(q); kp += 256) p.pyy[so + 2 * kp + q] = 0.5f * (pw[kp] + pw[fold_partner(q, kp)]);
}

// ---- host side ------------------------------------------------------------------------
// input auto spectra per chunk when every output channel has its own input channel:
// psx[q][cx][k] = sum over the chunk's pairs of px[cx][pair][k] (fp64).  grid = (n_chunks, n_cx)
__global__ __launch_bounds__(256) void k_px_sum(Args p) {
    const int q = blockIdx.x, cx = blockIdx.y;
    const int p0 = (int)((int64_t)q * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(q + 1) * p.n_pairs / p.n_chunks);
    const float* __restrict__ px = p.px + (int64_t)cx * p.n_pairs * NB;
    for (int k = threadIdx.x; k < NB; k += 256) {
        double sum = 0.0;
        for (int pr = p0; pr < p1; ++pr) sum += (double)px[(int64_t)pr * NB + k];
        p.psx[((int64_t)q * p.n_cx + cx) * NB + k] = (float)sum;
    }
}

struct Plan {
    int n_pairs, n_chunks;
    size_t bytes;
};
inline Plan plan(int n_frames, int n_cy, int n_cx = 1) {
    Plan pl;
    pl.n_pairs = (n_frames + 1) / 2;
    // one workgroup per CU (256) resident at once when there is enough work; fp32 chains <= 64 pairs
    int want = (256 + n_cy - 1) / n_cy;
    want = (want + 7) & ~7;
    const int by_len = (pl.n_pairs + 63) / 64;
    if (want < by_len) want = (by_len + 7) & ~7;
    want = std::max(1, std::min(want, pl.n_pairs));
    if (want >= 8) want &= ~7;
    pl.n_chunks = want;
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    pl.bytes = pad(sizeof(float2) * (size_t)n_cx * pl.n_pairs * N) + pad(sizeof(float) * (size_t)n_cx * pl.n_pairs * NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * NB) + pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cy * NB);
    return pl;
}

}  // namespace welch8k
