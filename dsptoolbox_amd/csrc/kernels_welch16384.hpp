// Welch H1/H2/H3, auto and cross spectra with a 16384-sample window on the 4096-point register
// transform of kernels_welch4096.hpp (reference: _welch, standard/_spectral_methods.py:10-173;
// compute_transfer_function, transfer_functions/transfer_functions.py:476-534).  gfx950.
//
//   16384 = 4 x 4096.  Class q holds the sub-spectrum Z[4k' + q] (radix-4 decimation in frequency
//   in front of the transform):
//       b_q[n'] = ( sum_j z[n' + 4096 j] (-i)^(j q) ) W16384^(n' q) ,   Z[4k' + q] = FFT4096(b_q)[k']
//   z = frame_2p w + i frame_2p+1 w (two real frames per complex transform).
//   One workgroup of 256 threads per class (blockIdx.z = q), two workgroups per CU (77 KB of LDS,
//   <= 256 registers), each with its own barriers; the four classes of a (chunk, channel) unit are
//   four workgroups -- one class per workgroup keeps the register budget of the 8192-sample
//   kernel (16 + 8 accumulators, 32 values, 32 input-spectrum values, 30 twiddles, a batch of loads);
//   both classes of a slot in one workgroup spilled 300 registers.
//   T[k] += conj(W[k]) Z[k], P[k] += |Z[k]|^2 over the 4096 bins of a class, 16 per thread, written
//   UNFOLDED per class; the fold k <-> N - k crosses the classes (N - (4k' + q) = 4 (4096 - k') - q:
//   class 0 and class 2 fold inside themselves, classes 1 and 3 into each other) and is applied
//   once per chunk by k_fold, and to the input auto spectra by k_px_sum.
//   Samples come through a range-checked buffer descriptor (zeros past the end of the signal), the
//   window from global memory (64 KB, L2-resident; the LDS holds two exchange buffers per slot).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kernels_fir16k.hpp"
#include "kernels_welch4096.hpp"

namespace welch16k {

namespace w4 = welch4096;
using w4::cmul;
using w4::pos16;
constexpr int N = 16384, M = 4096, NB = N / 2 + 1, NTB = 256;
constexpr int LDS_BYTES = (2 * w4::BUF_C + 256) * 8;  // two exchange buffers + W256 table: 76 800 B, two workgroups per CU

struct Args {
    const float* sig;  // x (k_x) or y (k_y), planar
    int64_t n_samples, ld;
    int n_ch, hop, n_frames, n_pairs, detrend;
    int n_chunks;
    const float* window;
    const float2* twt;  // welch4096::host_tables()
    const float2* twn;  // fir16k::host_tables(): [4][256] W16384^(t q), [4][16] W64^(n1 q)
    float4* xs;         // [n_cx][n_pairs][4][8][256]: class q, thread t holds bins 4 (t + 256 (2g, 2g+1)) + q
    float* pxu;         // [n_cx][n_pairs][4][4096]: |W|^2 per class, unfolded
    float2* pxy;        // [n_chunks][n_ch][NB]
    float* pyy;         // [n_chunks][n_ch][NB]
    float* psx;         // [n_chunks][n_cx][NB]
    int n_cx;           // input channels: 1 (shared) or n_ch (one per output channel)
    float2* tu;         // [n_chunks][n_ch][4][4096]: cross sums per class, unfolded
    float* pu;          // [n_chunks][n_ch][4][4096]: output power sums per class, unfolded
};

// samples must be addressable with 32-bit byte offsets through the channel's buffer descriptor
inline bool buf_fits(int64_t n_samples, int n_frames, int hop) {
    return n_samples < ((int64_t)1 << 29) && (int64_t)(n_frames + 2) * hop + N < ((int64_t)1 << 29);
}

// the last pair of an odd frame count when frame F would still overlap the signal
__device__ __forceinline__ bool needs_drop(const Args& p, int pr) {
    return pr == p.n_pairs - 1 && (p.n_frames & 1) && (int64_t)p.n_frames * p.hop < p.n_samples;
}

// Window, pack the two frames (a at sample a0, b at a0 + hop) and run the radix-4 front end of class
// q = s + 2 hi:  v[n1] = b_q[n'], n' = t + 256 n1.  (-i)^(jq): q = 0: 1,1,1,1; 1: 1,-i,-1,i;
// 2: 1,-1,1,-1; 3: 1,i,-1,-i.
__device__ __forceinline__ void front(float2 (&v)[16], __amdgpu_buffer_rsrc_t rs, uint32_t a0, uint32_t hop,
                                      const float* __restrict__ window, int s, int hi, float2 wt,
                                      const float2* __restrict__ c64, bool drop, int t) {
    // four n1 at a time: 32 sample and 16 window loads in flight, then their arithmetic (left to
    // itself hipcc hoists all 192 loads of a class to the front and spills 300 registers)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float a[4][4], b[4][4], w[4][4];
        // the lane index is laundered per batch: otherwise the 192 lane-dependent byte offsets are loop
        // invariants, hoisted out of the pair loop and spilled
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t n = (uint32_t)(tt + 256 * (4 * g + i) + M * j);
                w[i][j] = window[n];
                a[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((a0 + n) * 4u), 0, 0));
                b[i][j] =
                    __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((a0 + hop + n) * 4u), 0, 0));
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n1 = 4 * g + i;
            float2 z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = make_float2(a[i][j] * w[i][j], drop ? 0.f : b[i][j] * w[i][j]);
            float2 r;
            if (s == 0) {  // classes 0 and 2: (z0 + z2) +- (z1 + z3)
                const float2 e = make_float2(z[0].x + z[2].x, z[0].y + z[2].y), o = make_float2(z[1].x + z[3].x, z[1].y + z[3].y);
                r = hi ? make_float2(e.x - o.x, e.y - o.y) : make_float2(e.x + o.x, e.y + o.y);
            } else {  // classes 1 and 3: (z0 - z2) -+ i (z1 - z3)
                const float2 e = make_float2(z[0].x - z[2].x, z[0].y - z[2].y), o = make_float2(z[1].x - z[3].x, z[1].y - z[3].y);
                r = hi ? make_float2(e.x - o.y, e.y + o.x) : make_float2(e.x + o.y, e.y - o.x);  // e + i o : e - i o
            }
            v[n1] = (s == 0 && !hi) ? r : cmul(r, cmul(wt, c64[n1]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// fold partner of bin 4 k' + q: class of the partner and its index
__device__ __forceinline__ int fold_index(int q, int kp) { return q == 0 ? ((M - kp) & (M - 1)) : (M - 1 - kp); }

// ---- input spectra: grid = (n_pairs, n_cx, 4) ---------------------------------------
__global__ __launch_bounds__(NTB, 2) void k_x(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    const int tid = threadIdx.x, t = tid;
    float2* buf = lds;
    float2* tw2 = lds + 2 * w4::BUF_C;
    const int pr = blockIdx.x, cx = blockIdx.y, q = blockIdx.z, s = q & 1, hi = q >> 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.sig + (int64_t)cx * p.ld), 0, (int)(uint32_t)(p.n_samples * 4), 0x00020000);
    w4::Tw tw;
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw.w[k1 - 1] = p.twt[(k1 - 1) * 256 + t];
    if (tid < 256) tw2[tid] = p.twt[15 * 256 + tid];
    __syncthreads();
    const float2 wt = p.twn[q * 256 + t];
    float2 v[16];
    front(v, rs, (uint32_t)((int64_t)(2 * pr) * p.hop), (uint32_t)p.hop, p.window, s, hi, wt, p.twn + 4 * 256 + q * 16,
          needs_drop(p, pr), t);
    w4::fft4096_plain<true>(v, tw, buf, tw2, t);
    if (p.detrend && tid == 0 && q == 0) v[pos16(0)] = make_float2(0.f, 0.f);  // bin 0 = class 0, k' = 0
    const int64_t unit = ((int64_t)cx * p.n_pairs + pr) * 4 + q;
    float4* xo = p.xs + unit * (M / 2) + t;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float2 z0 = v[pos16(2 * g)], z1 = v[pos16(2 * g + 1)];
        xo[256 * g] = make_float4(z0.x, z0.y, z1.x, z1.y);
    }
    float* po = p.pxu + unit * M + t;
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
        const float2 z = v[pos16(k3)];
        po[256 * k3] = z.x * z.x + z.y * z.y;
    }
}

// ---- input auto spectra per chunk: psx[q][cx][k] = sum over the chunk's pairs of the folded
// |W|^2 (fp64), one thread per bin.  grid = (ceil(NB / 256), n_chunks, n_cx)
__global__ __launch_bounds__(256) void k_px_sum(Args p) {
    const int k = blockIdx.x * 256 + threadIdx.x, cq = blockIdx.y, cx = blockIdx.z;
    if (k >= NB) return;
    const int p0 = (int)((int64_t)cq * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(cq + 1) * p.n_pairs / p.n_chunks);
    const float* __restrict__ pxu = p.pxu + (int64_t)cx * p.n_pairs * N;
    const int q = k & 3, kp = (k >> 2) & (M - 1), qm = (4 - q) & 3;
    const int ia = q * M + kp, ib = qm * M + fold_index(q, kp);
    double sum = 0.0;
    for (int pr = p0; pr < p1; ++pr) sum += (double)pxu[(int64_t)pr * N + ia] + (double)pxu[(int64_t)pr * N + ib];
    p.psx[((int64_t)cq * p.n_cx + cx) * NB + k] = (float)(0.5 * sum);
}

// ---- output channels: grid = (n_chunks * n_ch, 1, 4) -----------------------------------
// AUTO: auto spectra only (ds_welch_psd): no input spectra, no cross sums.
template <bool AUTO = false>
__global__ __launch_bounds__(NTB, 2) void k_y(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    const int tid = threadIdx.x, t = tid;
    float2* buf = lds;
    float2* tw2 = lds + 2 * w4::BUF_C;
    const int q = blockIdx.z, s = q & 1, hi = q >> 1;
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
    const int p0 = (int)((int64_t)cq * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(cq + 1) * p.n_pairs / p.n_chunks);
    __syncthreads();  // W256 table written
    float2 T[16];
    float P[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        T[j] = make_float2(0.f, 0.f);
        P[j] = 0.f;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.sig + (int64_t)c * p.ld), 0, (int)(uint32_t)(p.n_samples * 4), 0x00020000);
    const int64_t xch = (int64_t)(p.n_cx > 1 ? c : 0) * p.n_pairs;
    const float2 wt = p.twn[q * 256 + t];
    const float2* c64 = p.twn + 4 * 256 + q * 16;
    for (int pr = p0; pr < p1; ++pr) {
        float2 v[16];
        // not loop invariant for the compiler: the 64 window values and 16 twiddle products of a
        // thread would be hoisted out of the pair loop into 96 live registers
        const float* win = p.window;
        const float2* c64l = c64;
        asm volatile("" : "+s"(win), "+s"(c64l));
        front(v, rs, (uint32_t)((int64_t)(2 * pr) * p.hop), (uint32_t)p.hop, win, s, hi, wt, c64l, needs_drop(p, pr), t);
        float2 xw[16];
        auto issue_xs = [&]() {
            if (AUTO) return;
            __builtin_amdgcn_sched_barrier(0);
            const float4* __restrict__ xp = p.xs + (((xch + pr) * 4 + q) * (M / 2)) + t;
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
        w4::fft4096<true>(v, tw, buf, tw2, t, ph, prev, w4::NoHook(), issue_xs);
#else
        w4::fft4096<true>(v, tw, buf, tw2, t, w4::NoHook(), issue_xs);
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
    if (p.detrend && tid == 0 && q == 0) P[0] = 0.f;  // class 0, k' = 0: xs bin 0 is already 0 -> T = 0 there
    const int64_t unit = (((int64_t)cq * p.n_ch + c) * 4 + q) * M + t;
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
        if (!AUTO) p.tu[unit + 256 * k3] = T[k3];
        p.pu[unit + 256 * k3] = P[k3];
    }
}

// ---- fold k <-> N - k of a chunk's class sums, one thread per bin: grid = (ceil(NB / 256), n_chunks * n_ch)
template <bool AUTO = false>
__global__ __launch_bounds__(256) void k_fold(Args p) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= NB) return;
    const int64_t u = blockIdx.y;  // cq * n_ch + c
    const int q = k & 3, kp = (k >> 2) & (M - 1), qm = (4 - q) & 3;
    const int ia = q * M + kp, ib = qm * M + fold_index(q, kp);
    if (!AUTO) {
        const float2 a = p.tu[u * N + ia], b = p.tu[u * N + ib];
        p.pxy[u * NB + k] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    }
    p.pyy[u * NB + k] = 0.5f * (p.pu[u * N + ia] + p.pu[u * N + ib]);
}

// ---- host side ------------------------------------------------------------------------
struct Plan {
    int n_pairs, n_chunks;
    size_t bytes;
};
inline Plan plan(int n_frames, int n_cy, int n_cx = 1) {
    Plan pl;
    pl.n_pairs = (n_frames + 1) / 2;
    // two workgroups per CU (512 = 128 units x 4 classes) resident at once when there is enough
    // work; fp32 chains <= 64 pairs
    int want = (128 + n_cy - 1) / n_cy;
    want = (want + 7) & ~7;
    const int by_len = (pl.n_pairs + 63) / 64;
    if (want < by_len) want = (by_len + 7) & ~7;
    want = std::max(1, std::min(want, pl.n_pairs));
    if (want >= 8) want &= ~7;
    pl.n_chunks = want;
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    pl.bytes = pad(sizeof(float2) * (size_t)n_cx * pl.n_pairs * N) + pad(sizeof(float) * (size_t)n_cx * pl.n_pairs * N) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * NB) + pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cy * NB) + pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * N) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cy * N);
    return pl;
}

}  // namespace welch16k
