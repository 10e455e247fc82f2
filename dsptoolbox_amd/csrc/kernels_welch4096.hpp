// Welch H1/H2/H3 for the headline shape: nfft 4096, ONE input channel, any hop
// (50 % overlap has a shorter load path).  gfx950.
//
// Data flow (all fp32, finish() in fp64):
//   k_x   : one workgroup per PAIR of input frames (2p, 2p+1):
//             Wp = FFT4096( x_2p w + i x_2p+1 w )      -> xs[p][...] (L2 resident, thread-major)
//             (|Wp[k]|^2 + |Wp[N-k]|^2)/2              -> px[p][0..2048]
//   k_y   : workgroup = (output channel c, chunk q of frame pairs); per pair
//             Zp = FFT4096( y_2p w + i y_2p+1 w )
//             T[k] += conj(Wp[k]) Zp[k] ,  P[k] += |Zp[k]|^2      (k = all 4096 bins,
//             16 per thread, in registers -- NO Hermitian separation per frame)
//           at the end of the chunk, once:
//             Sxy[k] = (T[k] + conj T[N-k])/2 ,  Syy[k] = (P[k] + P[N-k])/2 ,  k <= 2048
//           which is exact because both halves of a pair belong to the same channel:
//             conj(W[k]) Z[k] + conj( conj(W[N-k]) Z[N-k] ) = 2 (conj(X_a) Y_a + conj(X_b) Y_b).
//           every workgroup of a chunk also sums a slice of the chunk's px rows (fp64) -> psx[q]
//   k_welch_finish (kernels_finish.hpp): chunks -> H, coherence.
//
// FFT: 4096 = 16 x 16 x 16, 256 threads, 16 complex values per thread,
//   n = 256 n1 + 16 n2 + n3 ,  k = k1 + 16 k2 + 256 k3
//   pass 1  thread t = 16 n2 + n3 : DFT16 over n1 of z[t + 256 n1]  (straight from HBM),
//           times W4096^(t k1) (15 per-thread constants kept in registers)
//           -> LDS image [k1][t], row stride 272 complex (bank-conflict-free b64 reads)
//   pass 2  thread u = 16 k1 + n3 : DFT16 over n2, times W256^(n3 k2) (2 KB LDS table)
//           -> LDS image rows v = 16 k2 + k1 of 16 complex (n3), row stride 18 complex
//           (conflict-free ds_read_b128)
//   pass 3  thread v = k1 + 16 k2 : DFT16 over n3 -> Z[v + 256 k3] in registers
// Detrend: subtracting the mean of the windowed frame changes only bin 0 of its
// DFT (to 0), so it is applied as "skip bin 0" -- bin 0 is 0/0 rounding noise in
// the reference too.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <utility>
#include <vector>

#include "kernels_finish.hpp"

namespace welch4096 {

#ifndef W4_ABLATE
#define W4_ABLATE 0  // timing-only diagnostics (wrong results): 1 no xs loads, 2 no sample loads, 4 no LDS
#endif

#ifndef W4_AB
#define W4_AB 0  // timing-only ablations (wrong results), see kernels_welch4096w.hpp; 16: no internal W16 twiddles in dft16_h
#endif

#ifndef W4_TIMING
#define W4_TIMING 0  // dev only: per-phase s_memtime stamps of wave 0 / workgroup 0 -> w4_timing[]
#endif
#if W4_TIMING
__device__ unsigned long long w4_timing[16];
#define W4_TS(i)                                                  \
    do {                                                          \
        __builtin_amdgcn_sched_barrier(0);                        \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
        __builtin_amdgcn_sched_barrier(0);                        \
        if ((i) > 0) w4_ph[(i)-1] += t_ - w4_prev;                \
        w4_prev = t_;                                             \
    } while (0)
#else
#define W4_TS(i)
#endif

constexpr int N = 4096, NT = 256, NB = N / 2 + 1;
constexpr int L1S = 272, L2S = 18;
constexpr int BUF_C = 256 * L2S;  // 4608 complex >= 16 * 272
constexpr int LDS_BYTES = BUF_C * 8 + 256 * 8;        // one exchange buffer + W256 table
constexpr int LDS_BYTES_2 = 2 * BUF_C * 8 + 256 * 8;  // two exchange buffers: 2 barriers per FFT

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// in-place radix-4 butterfly on (a,b,c,d), forward
__device__ __forceinline__ void r4(float2& a, float2& b, float2& c, float2& d) {
    float2 s0 = make_float2(a.x + c.x, a.y + c.y), d0 = make_float2(a.x - c.x, a.y - c.y);
    float2 s1 = make_float2(b.x + d.x, b.y + d.y), d1 = make_float2(b.x - d.x, b.y - d.y);
    a = make_float2(s0.x + s1.x, s0.y + s1.y);
    c = make_float2(s0.x - s1.x, s0.y - s1.y);
    b = make_float2(d0.x + d1.y, d0.y - d1.x);  // d0 - i d1
    d = make_float2(d0.x - d1.y, d0.y + d1.x);  // d0 + i d1
}

// 16-point DFT in registers.  Input v[n], n = n0 + 4 n1.  Output X[k] in v[4*(k&3) + (k>>2)].
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    constexpr float C8 = 0.92387953251128673848f, S8 = 0.38268343236508978178f;
    constexpr float R2 = 0.70710678118654752440f;
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) r4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    // position n0 + 4 k1 holds Y[n0][k1]; multiply by W16^(n0 k1)
    auto mulw = [](float2 z, float c, float s) {  // z * (c - i s)
        return make_float2(fmaf(z.x, c, z.y * s), fmaf(z.y, c, -z.x * s));
    };
    v[1 + 4] = mulw(v[1 + 4], C8, S8);                                  // W16^1
    v[1 + 8] = make_float2((v[9].x + v[9].y) * R2, (v[9].y - v[9].x) * R2);  // W16^2
    v[1 + 12] = mulw(v[1 + 12], S8, C8);                                // W16^3
    v[2 + 4] = make_float2((v[6].x + v[6].y) * R2, (v[6].y - v[6].x) * R2);  // W16^2
    v[2 + 8] = make_float2(v[10].y, -v[10].x);                          // W16^4 = -i
    v[2 + 12] = make_float2((v[14].y - v[14].x) * R2, -(v[14].x + v[14].y) * R2);  // W16^6
    v[3 + 4] = mulw(v[3 + 4], S8, C8);                                  // W16^3
    v[3 + 8] = make_float2((v[11].y - v[11].x) * R2, -(v[11].x + v[11].y) * R2);  // W16^6
    v[3 + 12] = mulw(v[3 + 12], -C8, -S8);                              // W16^9 = -W16^1
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) r4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
__device__ __forceinline__ constexpr int pos16(int k) { return 4 * (k & 3) + (k >> 2); }

// dft16 with a call-out after each of its eight radix-4 butterflies: stage A (n0 = 0..3, its
// three internal W16 twiddles folded in) then stage B (k1g = 0..3; afterwards v[4 k1g + j] holds
// X[k1g + 4 j]).  The call-outs carry the LDS / global traffic of the surrounding exchange so that
// it is issued a few operations at a time between the butterflies instead of in one burst (a burst
// of 16 ds_write_b64 from all four waves fills the LDS command queue: rocprof SQ_WAIT_INST_LDS was
// 20 % of the wave cycles).
#define W4_PIN() __builtin_amdgcn_sched_barrier(0)
template <typename PA, typename HA, typename HB>
__device__ __forceinline__ void dft16_h(float2 (&v)[16], PA pre_a, HA after_a, HB after_b) {
    constexpr float C8 = 0.92387953251128673848f, S8 = 0.38268343236508978178f;
    constexpr float R2 = 0.70710678118654752440f;
    auto mulw = [](float2 z, float c, float s) {  // z * (c - i s)
        return make_float2(fmaf(z.x, c, z.y * s), fmaf(z.y, c, -z.x * s));
    };
    pre_a(std::integral_constant<int, 0>{});
    r4(v[0], v[4], v[8], v[12]);
    after_a(0);
    pre_a(std::integral_constant<int, 1>{});
    r4(v[1], v[5], v[9], v[13]);
    if (!(W4_AB & 16)) {
    v[5] = mulw(v[5], C8, S8);                                                  // W16^1
    v[9] = make_float2((v[9].x + v[9].y) * R2, (v[9].y - v[9].x) * R2);         // W16^2
    v[13] = mulw(v[13], S8, C8);                                                // W16^3
    }
    after_a(1);
    pre_a(std::integral_constant<int, 2>{});
    r4(v[2], v[6], v[10], v[14]);
    if (!(W4_AB & 16)) {
    v[6] = make_float2((v[6].x + v[6].y) * R2, (v[6].y - v[6].x) * R2);         // W16^2
    v[10] = make_float2(v[10].y, -v[10].x);                                     // W16^4 = -i
    v[14] = make_float2((v[14].y - v[14].x) * R2, -(v[14].x + v[14].y) * R2);   // W16^6
    }
    after_a(2);
    pre_a(std::integral_constant<int, 3>{});
    r4(v[3], v[7], v[11], v[15]);
    if (!(W4_AB & 16)) {
    v[7] = mulw(v[7], S8, C8);                                                  // W16^3
    v[11] = make_float2((v[11].y - v[11].x) * R2, -(v[11].x + v[11].y) * R2);   // W16^6
    v[15] = mulw(v[15], -C8, -S8);                                              // W16^9
    }
    after_a(3);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        r4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        after_b(g);
    }
}
struct NoHookI {
    template <typename T>
    __device__ __forceinline__ void operator()(T) const {}
};


struct Tw {
    float2 w[15];  // W4096^(t k1), k1 = 1..15
};

// Transform the 16 register values (pass-1 inputs z[t + 256 n1]) into Z[t + 256 k3] (in
// v[pos16(k3)]).  `buf` is the workgroup's LDS exchange buffer, `tw2` the W256 table.
// The caller guarantees nobody still reads buf (one barrier before the first write here).
// TWO_BUF: pass-1 and pass-2 images live in different buffers (bufA, bufB); then the only
// hazards are write-after-read across iterations, which the two barriers already order.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// `behind_ex1`: called after the first exchange image has been written, before its barrier: work
// put there (issuing global loads) runs while the workgroup drains its LDS stores and waits.
template <bool TWO_BUF, typename Hook = NoHook, typename Hook2 = NoHook>
__device__ __forceinline__ void fft4096(float2 (&v)[16], const Tw& tw, float2* __restrict__ buf,
                                        const float2* __restrict__ tw2, int tid
#if W4_TIMING
                                        , unsigned long long (&w4_ph)[12], unsigned long long& w4_prev
#endif
                                        , Hook behind_ex1 = Hook(), Hook2 behind_ex2 = Hook2()) {
    float2* __restrict__ bufB = TWO_BUF ? buf + BUF_C : buf;
    dft16(v);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[pos16(k1)] = cmul(v[pos16(k1)], tw.w[k1 - 1]);
    W4_TS(2);
    const int k1u = tid >> 4, n3 = tid & 15;
    if (!(W4_ABLATE & 4)) {
        if (!TWO_BUF) __syncthreads();  // previous readers of buf are done
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) buf[k1 * L1S + tid] = v[pos16(k1)];
    }
    // the pass-2 twiddles do not depend on the exchange: fetch them (into the registers the
    // stored values just left) before the barrier, so their latency hides behind it
    float2 w2[15];
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) w2[k2 - 1] = tw2[k2 * 16 + n3];
    behind_ex1();
    if (!(W4_ABLATE & 4)) {
        W4_TS(3);
        __syncthreads();
        W4_TS(4);
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) v[n2] = buf[k1u * L1S + 16 * n2 + n3];
    }
    W4_TS(5);
    dft16(v);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) v[pos16(k2)] = cmul(v[pos16(k2)], w2[k2 - 1]);
    W4_TS(6);
    if (!(W4_ABLATE & 4)) {
        if (!TWO_BUF) __syncthreads();  // all pass-2 reads done
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) bufB[(16 * k2 + k1u) * L2S + n3] = v[pos16(k2)];
        behind_ex2();
        W4_TS(7);
        __syncthreads();
        W4_TS(8);
        const float4* row = reinterpret_cast<const float4*>(bufB + tid * L2S);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 r = row[j];
            v[2 * j] = make_float2(r.x, r.y);
            v[2 * j + 1] = make_float2(r.z, r.w);
        }
    }
    W4_TS(9);
    dft16(v);
    W4_TS(10);
}

// the transform without the dev-only phase stamps (other kernels built on it)
template <bool TWO_BUF>
__device__ __forceinline__ void fft4096_plain(float2 (&v)[16], const Tw& tw, float2* __restrict__ buf,
                                              const float2* __restrict__ tw2, int tid) {
#if W4_TIMING
    unsigned long long ph[12] = {}, prev = 0;
    fft4096<TWO_BUF>(v, tw, buf, tw2, tid, ph, prev);
#else
    fft4096<TWO_BUF>(v, tw, buf, tw2, tid);
#endif
}

struct Args {
    const float* sig;   // x (k_x) or y (k_y) planar
    int64_t n_samples, ld;
    int n_ch, hop, n_frames, n_pairs, detrend;
    int n_chunks, ppc;  // k_y: pairs per chunk
    const float* window;
    const float2* twt;  // host_tables()
    float2* xs;         // [n_pairs][8][256][2]: bins (tid + 256*2g, tid + 256*(2g+1)) of thread tid
                        // adjacent -> one coalesced 16-byte load per two bins
    float* px;          // [n_pairs][NB]
    float2* pxy;        // [n_chunks][n_ch][NB]
    float* pyy;         // [n_chunks][n_ch][NB]
    float* psx;         // [n_chunks][NB]: px summed over the pairs of a chunk (k_y)
    // k_y3 only: chunk q holds n_pairs / n_chunks pairs, plus one if bit q of `plus` is set (the
    // host places the remainder so that every XCD gets the same number of transforms); 0 = even split
    int use_plus;
    uint32_t plus[24];
    // k_x3 / k_y3 only: number of input channels (0 or 1: one input for every output channel; else
    // one per output channel: xs[cx][pair][...], px[cx][pair][NB], psx[chunk][cx][NB])
    int n_cx;
    const float* xsig;  // tools/exp/kernels_welch4096f.hpp only: the input channel (sig = the output channels)
};

// twt: fp64-computed tables, [15][256] W4096^(t k1) (k1 = 1..15) then [16][16] W256^(n3 k2)
constexpr int TWT_LEN = 15 * 256 + 256;
__device__ __forceinline__ void init_tables(Tw& tw, float (&win)[16], float2* tw2,
                                            const float* __restrict__ window,
                                            const float2* __restrict__ twt, int tid) {
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw.w[k1 - 1] = twt[(k1 - 1) * 256 + tid];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) win[n1] = window[tid + 256 * n1];
    tw2[tid] = twt[15 * 256 + tid];  // tw2[k2*16 + n3] = W256^(n3 k2)
    __syncthreads();  // the table is read before the first exchange barrier of the transform
}
inline void host_tables(std::vector<float2>& t) {
    t.resize(TWT_LEN);
    for (int k1 = 1; k1 < 16; ++k1)
        for (int tt = 0; tt < 256; ++tt) {
            double a = -2.0 * M_PI * (double)(tt * k1) / 4096.0;
            t[(k1 - 1) * 256 + tt] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k2 = 0; k2 < 16; ++k2)
        for (int n3 = 0; n3 < 16; ++n3) {
            double a = -2.0 * M_PI * (double)(n3 * k2) / 256.0;
            t[15 * 256 + k2 * 16 + n3] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
}

// Raw samples of the frame pair (2p, 2p+1) of one channel.  HALF_HOP (hop == 2048): the two
// frames share half their samples -> 24 loads s[m] = ch[start + tid + 256 m]; otherwise 32.
// Interior pairs take unconditional loads behind ONE wave-uniform test (a per-load
// "in range ? load : 0" makes hipcc branch around every load and drain vmcnt each time);
// the ragged tail clamps the address and selects the value.
template <bool HALF_HOP>
struct Raw {
    float s[HALF_HOP ? 24 : 32];
};

template <bool HALF_HOP>
__device__ __forceinline__ void load_raw(Raw<HALF_HOP>& r, const float* __restrict__ ch,
                                         int64_t n_samples, int64_t start0, int hop, int tid) {
    const float* __restrict__ src = ch + start0;  // wave-uniform base
    const int64_t remain = n_samples - start0;    // >= 1 for every valid pair
    const int span = HALF_HOP ? 3 * 2048 : hop + 4096;
    if (remain >= span) {
        if (HALF_HOP) {
#pragma unroll
            for (int m = 0; m < 24; ++m) r.s[m] = src[tid + 256 * m];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                r.s[n1] = src[tid + 256 * n1];
                r.s[16 + n1] = src[hop + tid + 256 * n1];
            }
        }
    } else {
        const int last = (int)(remain > (int64_t)(1 << 30) ? (1 << 30) : remain) - 1;
        constexpr int CNT = HALF_HOP ? 24 : 32;
#pragma unroll
        for (int m = 0; m < CNT; ++m) {
            int i = HALF_HOP ? tid + 256 * m : (m < 16 ? tid + 256 * m : hop + tid + 256 * (m - 16));
            float a = src[min(i, last)];
            r.s[m] = i <= last ? a : 0.f;
        }
    }
}

// z[n1] = frame_2p[n] w[n] + i frame_2p+1[n] w[n], n = tid + 256 n1
// (An odd frame count leaves the last pair without a second frame.  With the reference's framing
// F = ceil(N/hop) that frame would start at F*hop >= n_samples, so load_raw has already delivered
// zeros for it; a caller that passes fewer frames is handled by drop_second() on that one pair.)
template <bool HALF_HOP>
__device__ __forceinline__ void window_pair(float2 (&v)[16], const Raw<HALF_HOP>& r, const float (&win)[16]) {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        float b = HALF_HOP ? r.s[n1 + 8] : r.s[16 + n1];
        v[n1] = make_float2(r.s[n1] * win[n1], b * win[n1]);
    }
}

// the last pair of an odd frame count when frame F would still overlap the signal
__device__ __forceinline__ bool needs_drop(const Args& p, int pr) {
    return pr == p.n_pairs - 1 && (p.n_frames & 1) && (int64_t)p.n_frames * p.hop < p.n_samples;
}
__device__ __forceinline__ void drop_second(float2 (&v)[16]) {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1].y = 0.f;
}

// ---- input spectra -----------------------------------------------------------
template <bool HALF_HOP>
__global__ __launch_bounds__(NT) void k_x(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + BUF_C;
    const int tid = threadIdx.x, pr = blockIdx.x, cx = blockIdx.y;  // grid = (n_pairs, input channels)
    Tw tw;
    float win[16];
    float2 v[16];
    {
        // one transform per workgroup: the kernel is a latency chain, so the samples are
        // requested before the tables (whose barrier would otherwise be waited for first)
        Raw<HALF_HOP> raw;
        load_raw<HALF_HOP>(raw, p.sig + (int64_t)cx * p.ld, p.n_samples, (int64_t)(2 * pr) * p.hop, p.hop, tid);
        init_tables(tw, win, tw2, p.window, p.twt, tid);
        window_pair<HALF_HOP>(v, raw, win);
        if (needs_drop(p, pr)) drop_second(v);
    }
    fft4096_plain<false>(v, tw, buf, tw2, tid);
    if (p.detrend && tid == 0) v[pos16(0)] = make_float2(0.f, 0.f);
    float4* xo = reinterpret_cast<float4*>(p.xs + ((int64_t)cx * p.n_pairs + pr) * N) + tid;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float2 z0 = v[pos16(2 * g)], z1 = v[pos16(2 * g + 1)];
        xo[256 * g] = make_float4(z0.x, z0.y, z1.x, z1.y);
    }
    // symmetrised power for Sxx
    float* pw = reinterpret_cast<float*>(buf);
    __syncthreads();
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
        float2 z = v[pos16(k3)];
        pw[tid + 256 * k3] = z.x * z.x + z.y * z.y;
    }
    __syncthreads();
    float* po = p.px + ((int64_t)cx * p.n_pairs + pr) * NB;
    for (int k = tid; k < NB; k += NT) po[k] = 0.5f * (pw[k] + pw[(N - k) & (N - 1)]);
}

// ---- output channels ---------------------------------------------------------
// Two workgroups per CU (<= 256 VGPRs, 2 x 37 KB of LDS each).  Occupancy 3 (<= 168 VGPRs:
// on-the-fly Hann window, input spectrum loaded where used) and 4 were built and measured:
// the spills / exposed load latency cost more than the third workgroup hides (147 vs 108 us).
// AUTO: auto spectra only (ds_welch_psd): no input spectra, no cross sums, no psx.
template <bool HALF_HOP, bool AUTO = false>
__global__ __launch_bounds__(NT, 2) void k_y(Args p) {
    constexpr bool TWO_BUF = true;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + (TWO_BUF ? 2 : 1) * BUF_C;
    const int tid = threadIdx.x;
    // XCD-aware decode: blocks b, b+8, ... share an XCD (and its L2): give each XCD whole
    // chunks so the input spectra it re-reads for every channel stay in that L2.
    int q, c;
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
    Tw tw;
    float win[16];
    init_tables(tw, win, tw2, p.window, p.twt, tid);
    const float* ch = p.sig + (int64_t)c * p.ld;
    // balanced split of the pairs over the chunks (n_chunks stays a multiple of 8 for the XCD mapping)
    const int p0 = (int)((int64_t)q * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(q + 1) * p.n_pairs / p.n_chunks);
    if (!AUTO && p.n_cx <= 1) {  // (one input channel per output channel: welch4096::k_px_sum does it)
        // Input auto-spectrum of this chunk: the px rows k_x wrote, summed in fp64 -- instead of a
        // separate reduction kernel every workgroup of the chunk takes a slice of the bins
        // (8 row groups x 32 bins per sweep, independent loads, combined through LDS).
        double* red = reinterpret_cast<double*>(lds);  // [8][32], before the first transform
        const int bpc = (NB + p.n_ch - 1) / p.n_ch;
        const int b0 = c * bpc, b1 = min(b0 + bpc, NB);
        const int rg = tid >> 5, kl = tid & 31;
        for (int kb = b0; kb < b1; kb += 32) {
            const int k = kb + kl;
            double sum = 0.0;
            if (k < b1)
                for (int pr = p0 + rg; pr < p1; pr += 8) sum += (double)p.px[(int64_t)pr * NB + k];
            red[rg * 32 + kl] = sum;
            __syncthreads();
            if (rg == 0 && k < b1) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) t += red[j * 32 + kl];
                p.psx[(int64_t)q * NB + k] = (float)t;
            }
            __syncthreads();
        }
    }
    float2 T[16];
    float P[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        T[j] = make_float2(0.f, 0.f);
        P[j] = 0.f;
    }
    if constexpr (TWO_BUF) {
        // occupancy 2: registers to spare -> software pipeline: the raw samples of pair pr+1
        // and the input spectrum of pair pr are in flight while pair pr is transformed
        Raw<HALF_HOP> raw;
        if (p0 < p1) load_raw<HALF_HOP>(raw, ch, p.n_samples, (int64_t)(2 * p0) * p.hop, p.hop, tid);
#if W4_TIMING
        unsigned long long w4_ph[12] = {}, w4_prev = 0;
#endif
        for (int pr = p0; pr < p1; ++pr) {
            float2 v[16];
            W4_TS(0);
            window_pair<HALF_HOP>(v, raw, win);
            if (needs_drop(p, pr)) drop_second(v);
            float2 xw[16];
            // The raw samples of the next pair and the input spectrum of this one are requested
            // behind the first exchange's stores: the ~500 cycles it takes a wave to issue 32
            // global loads overlap the LDS store drain and the barrier instead of preceding the
            // first butterfly.  (They are consumed thousands of cycles later.)
            auto issue_loads = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                if (!(W4_ABLATE & 2) && pr + 1 < p1)
                    load_raw<HALF_HOP>(raw, ch, p.n_samples, (int64_t)(2 * pr + 2) * p.hop, p.hop, tid);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto issue_xs = [&]() {
                if (AUTO) return;
                __builtin_amdgcn_sched_barrier(0);
                if (W4_ABLATE & 1) {
#pragma unroll
                    for (int k3 = 0; k3 < 16; ++k3) xw[k3] = tw.w[k3 % 15];
                } else {
                    const float4* __restrict__ xp =
                        reinterpret_cast<const float4*>(p.xs + ((int64_t)(p.n_cx > 1 ? c : 0) * p.n_pairs + pr) * N) + tid;
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        float4 q = xp[256 * g];
                        xw[2 * g] = make_float2(q.x, q.y);
                        xw[2 * g + 1] = make_float2(q.z, q.w);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            W4_TS(1);
#if W4_TIMING
            fft4096<TWO_BUF>(v, tw, buf, tw2, tid, w4_ph, w4_prev, issue_loads, issue_xs);
#else
            fft4096<TWO_BUF>(v, tw, buf, tw2, tid, issue_loads, issue_xs);
#endif
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) {
                float2 z = v[pos16(k3)];
                if (!AUTO) {
                    float2 w = xw[k3];
                    T[k3].x = fmaf(w.x, z.x, fmaf(w.y, z.y, T[k3].x));   // conj(w) z
                    T[k3].y = fmaf(w.x, z.y, fmaf(-w.y, z.x, T[k3].y));
                }
                P[k3] = fmaf(z.x, z.x, fmaf(z.y, z.y, P[k3]));
            }
            W4_TS(11);
        }
#if W4_TIMING
        if (blockIdx.x == 0 && tid == 0) {
            for (int i = 0; i < 11; ++i) atomicAdd(&w4_timing[i], w4_ph[i]);
            atomicAdd(&w4_timing[15], (unsigned long long)(p1 - p0));
        }
#endif
    }
    if (p.detrend && tid == 0) P[0] = 0.f;  // xs bin 0 is already 0 -> T[0] = 0
    // fold k <-> N-k once per chunk, through LDS
    __syncthreads();
    const int64_t so = ((int64_t)q * p.n_ch + c) * NB;
    if (!AUTO) {
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) buf[tid + 256 * k3] = T[k3];
        __syncthreads();
        for (int k = tid; k < NB; k += NT) {
            float2 a = buf[k], b = buf[(N - k) & (N - 1)];
            p.pxy[so + k] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
        }
        __syncthreads();
    }
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) pw[tid + 256 * k3] = P[k3];
    __syncthreads();
    for (int k = tid; k < NB; k += NT) p.pyy[so + k] = 0.5f * (pw[k] + pw[(N - k) & (N - 1)]);
}

// ---- host side -----------------------------------------------------------------
// `want`: the caller's chunk count (ds_config::welch_chunks), 0 = choose here
inline int chunks_for(int n_pairs, int n_ch, int want = 0) {
    if (want <= 0) {
        // exactly two workgroups per CU (512 on the 256 CUs) when there is enough work: all of
        // them are resident at once, no tail; fp32 accumulation chains stay <= 64 pairs
        want = (512 + n_ch - 1) / n_ch;
        want = (want + 7) & ~7;
        int by_len = (n_pairs + 63) / 64;
        if (want < by_len) want = (by_len + 7) & ~7;
    }
    if (want > n_pairs) want = n_pairs;
    if (want >= 8) want &= ~7;  // whole chunks per XCD
    if (want < 1) want = 1;
    return want;
}

struct Plan {
    int n_pairs, n_chunks, ppc;
    size_t bytes;
};
inline Plan plan(int n_frames, int n_cy, int want_chunks = 0) {
    Plan pl;
    pl.n_pairs = (n_frames + 1) / 2;
    pl.n_chunks = chunks_for(pl.n_pairs, n_cy, want_chunks);  // chunk q = pairs [q n_pairs / n_chunks, (q+1) n_pairs / n_chunks)
    pl.ppc = (pl.n_pairs + pl.n_chunks - 1) / pl.n_chunks;
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    pl.bytes = pad(sizeof(float2) * (size_t)pl.n_pairs * N) + pad(sizeof(float) * (size_t)pl.n_pairs * NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * NB) + pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cy * NB);
    return pl;
}

}  // namespace welch4096
