// Welch H1 / H2 / H3, auto and cross spectra with windows of 2^15 ... 2^18 samples on the 4096-point register
// transform (reference: _welch, standard/_spectral_methods.py:10-173 -- windows up to 2^18 are allowed, :89-93;
// compute_transfer_function, transfer_functions/transfer_functions.py:476-534).  gfx950.  Round 4.
//
//   N = R x 4096, R = 8 ... 64.  Decimation in frequency: class r holds the bins R k' + r,
//       b_r[m] = ( sum_{s < R} z[m + 4096 s] W_R^(r s) ) W_N^(r m) ,   Z[R k' + r] = FFT4096(b_r)[k'] ,
//   z = frame_2p w + i frame_2p+1 w (two real frames ride one complex sequence, as everywhere).
//
//   Two passes instead of the four-step transform's five (columns, frame means, rows, unpack, frame sums: 2.0-2.7 ms
//   for 64 + 1 channels x 2^20 samples):
//     k_dif<R>   one thread per m: 2 R samples and R window values in (coalesced over m), an R-point DFT over s in
//                registers, the twiddle W_N^(r m) from an [R][4096] table (fp64-computed), R complex values out
//                (coalesced over m): every sample is read once per frame it belongs to, b is written once;
//     k_xc/k_yc  the headline kernel's loop (welch4096::fft4096_wi, three workgroups per CU) on the COMPLEX
//                sequences b_r -- no window, no frame packing, 16 eight-byte loads per transform --, one workgroup
//                per (chunk of pairs, channel, class): T[k'] += conj(W[k']) Z[k'], P[k'] += |Z[k']|^2, written
//                UNFOLDED per class;
//     k_fold     the fold k <-> N - k crosses the classes (N - (R k' + r) = R (4096 - k') - r: class 0 into itself,
//                class r into class R - r), once per chunk; k_px_sum the same for the input auto spectra;
//     k_welch_finish as for every other window length.
//   Detrend (mean of the windowed frame) only changes bin 0 = class 0, k' = 0: skipped there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "kernels_welch4096w.hpp"

namespace welchl {

namespace w4 = welch4096;
using w4::cmul;
using w4::pos16;
constexpr int M = 4096, NT = 256;
constexpr int LDS_BYTES = (16 * w4::L1S + 256) * 8;  // exchange image + W256 table: 36 864 B

struct Args {
    const float* sig;  // x (k_dif / k_xc) or y (k_dif / k_yc), planar
    int64_t n_samples, ld;
    int n_ch, hop, n_frames, n_pairs, detrend;
    int n_chunks;
    int R, lgR;           // window = R * 4096 samples
    const float* window;  // [R * 4096]
    const float2* twt;    // welch4096::host_tables()
    const float2* twl;    // host_tables(R): [R][4096] W_N^(r m), then [R] W_R^k
    float2* b;            // [channel][pair][R][4096]: k_dif's output, k_xc / k_yc's input
    float4* xs;           // [n_cx][pair][R][8][256]: input spectra in register layout
    float* pxu;           // [n_cx][pair][R][4096]: |W|^2 per class, unfolded
    float2* pxy;          // [n_chunks][n_ch][NB]
    float* pyy;           // [n_chunks][n_ch][NB]
    float* psx;           // [n_chunks][n_cx][NB]
    int n_cx;             // input channels: 1 (shared) or n_ch (one per output channel)
    float2* tu;           // [n_chunks][n_ch][R][4096]: cross sums per class, unfolded
    float* pu;            // [n_chunks][n_ch][R][4096]: output power sums per class, unfolded
};

// twl: [R][4096] W_N^(r m) (N = R 4096), then [R] W_R^k; fp64-computed
inline void host_tables(int R, std::vector<float2>& t) {
    const double N = (double)R * M;
    t.resize((size_t)R * M + R);
    for (int r = 0; r < R; ++r)
        for (int m = 0; m < M; ++m) {
            const double a = -2.0 * M_PI * (double)(((int64_t)r * m) % (int64_t)N) / N;
            t[(size_t)r * M + m] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k = 0; k < R; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)R;
        t[(size_t)R * M + k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
}

// samples are addressed with 32-bit byte offsets through the channel's buffer descriptor
inline bool buf_fits(int64_t n_samples, int n_frames, int hop, int W) {
    return n_samples < ((int64_t)1 << 29) && (int64_t)(n_frames + 2) * hop + W < ((int64_t)1 << 29);
}
inline int classes_of(int W) {
    return (W == 16384 || W == 32768 || W == 65536 || W == 131072 || W == 262144) ? W / M : 0;
}

__device__ __forceinline__ bool needs_drop(const Args& p, int pr) {
    return pr == p.n_pairs - 1 && (p.n_frames & 1) && (int64_t)p.n_frames * p.hop < p.n_samples;
}

// R-point DFT over s in registers: z[s] in, Z[r] out in natural order (forward, W_R = exp(-2 pi i / R)).
// wr[k] = W_R^k (wave-uniform table).
template <int R>
__device__ __forceinline__ void dft_small(float2 (&z)[R], const float2* __restrict__ wr) {
    if constexpr (R == 4) {
        w4::r4(z[0], z[1], z[2], z[3]);
    } else if constexpr (R == 8) {
        w4::r4(z[0], z[2], z[4], z[6]);  // E[k] in z[0], z[2], z[4], z[6]
        w4::r4(z[1], z[3], z[5], z[7]);  // O[k] in z[1], z[3], z[5], z[7]
        float2 out[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float2 e = z[2 * k], o = k ? cmul(z[2 * k + 1], wr[k]) : z[1];
            out[k] = make_float2(e.x + o.x, e.y + o.y);
            out[k + 4] = make_float2(e.x - o.x, e.y - o.y);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = out[k];
    } else if constexpr (R == 16) {
        float2 v[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = z[s];
        w4::dft16(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) z[k] = v[pos16(k)];
    } else {
        // R = 16 Q: s = Q s1 + s0.  Y_s0 = DFT16 over s1;  Z[k + 16 r1] = sum_s0 W_Q^(r1 s0) ( W_R^(k s0) Y_s0[k] )
        constexpr int Q = R / 16;
        float2 y[Q][16];
#pragma unroll
        for (int s0 = 0; s0 < Q; ++s0) {
            float2 v[16];
#pragma unroll
            for (int s1 = 0; s1 < 16; ++s1) v[s1] = z[Q * s1 + s0];
            w4::dft16(v);
#pragma unroll
            for (int k = 0; k < 16; ++k) y[s0][k] = (s0 && k) ? cmul(v[pos16(k)], wr[k * s0]) : v[pos16(k)];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (Q == 2) {
                z[k] = make_float2(y[0][k].x + y[1][k].x, y[0][k].y + y[1][k].y);
                z[k + 16] = make_float2(y[0][k].x - y[1][k].x, y[0][k].y - y[1][k].y);
            } else {
                float2 a = y[0][k], b = y[1][k], c = y[2][k], d = y[3][k];
                w4::r4(a, b, c, d);
                z[k] = a;
                z[k + 16] = b;
                z[k + 32] = c;
                z[k + 48] = d;
            }
        }
    }
}

// ---- pass 1: windowed frame pairs -> class sequences.  grid = (16, n_pairs, n_ch) ----------------------------
template <int R>
__global__ __launch_bounds__(NT) void k_dif(Args p) {
    const int m = (int)blockIdx.x * NT + (int)threadIdx.x, pr = blockIdx.y, c = blockIdx.z;
    const __amdgpu_buffer_rsrc_t rs = w4::channel_rsrc(p.sig + (int64_t)c * p.ld, p.n_samples);
    const uint32_t a0 = (uint32_t)((int64_t)(2 * pr) * p.hop) + (uint32_t)m;
    const bool drop = needs_drop(p, pr);
    float2 z[R];
#pragma unroll
    for (int s = 0; s < R; ++s) {
        const float w = p.window[m + M * s];
        const float a = w4::ld_sample(rs, (int)((a0 + (uint32_t)(M * s)) * 4u));
        const float b = w4::ld_sample(rs, (int)((a0 + (uint32_t)p.hop + (uint32_t)(M * s)) * 4u));
        z[s] = make_float2(a * w, drop ? 0.f : b * w);
    }
    dft_small<R>(z, p.twl + (size_t)R * M);
    float2* out = p.b + (((int64_t)c * p.n_pairs + pr) * R) * M + m;
#pragma unroll
    for (int r = 0; r < R; ++r) out[(int64_t)r * M] = r ? cmul(z[r], p.twl[(size_t)r * M + m]) : z[0];
}

// fold partner of bin R k' + r: class (R - r) mod R, index
__device__ __forceinline__ int fold_index(int r, int kp) { return r == 0 ? ((M - kp) & (M - 1)) : (M - 1 - kp); }

// ---- input spectra: one transform per workgroup.  grid = n_pairs * R * n_cx ---------------------------------
__global__ __launch_bounds__(NT) void k_xc(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    const int tid = threadIdx.x;
    const int64_t unit = blockIdx.x;  // (cx * n_pairs + pair) * R + r
    const int r = (int)(unit & (p.R - 1));
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    float2 v[16];
    const float2* src = p.b + unit * M + tid;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = src[256 * n1];
    w4::fft4096_w(v, tw, buf, tw2, tid);
    if (p.detrend && tid == 0 && r == 0) v[pos16(0)] = make_float2(0.f, 0.f);  // bin 0 = class 0, k' = 0
    float4* xo = p.xs + unit * (M / 2) + tid;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float2 z0 = v[pos16(2 * g)], z1 = v[pos16(2 * g + 1)];
        xo[256 * g] = make_float4(z0.x, z0.y, z1.x, z1.y);
    }
    const int bt = w4::bin_thread(tid);
    float* po = p.pxu + unit * M + bt;
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
        const float2 z = v[pos16(k3)];
        po[256 * k3] = z.x * z.x + z.y * z.y;
    }
}

// ---- input auto spectra per chunk: psx[q][cx][k] = sum over the chunk's pairs of the folded |W|^2 (fp64).
// grid = (ceil(NB / 256), n_chunks, n_cx)
__global__ __launch_bounds__(256) void k_px_sum(Args p) {
    const int nb = p.R * (M / 2) + 1, N = p.R * M;
    const int k = blockIdx.x * 256 + threadIdx.x, cq = blockIdx.y, cx = blockIdx.z;
    if (k >= nb) return;
    const int p0 = (int)((int64_t)cq * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(cq + 1) * p.n_pairs / p.n_chunks);
    const float* __restrict__ pxu = p.pxu + (int64_t)cx * p.n_pairs * N;
    const int r = k & (p.R - 1), kp = (k >> p.lgR) & (M - 1), rm = (p.R - r) & (p.R - 1);
    const int ia = r * M + kp, ib = rm * M + fold_index(r, kp);
    double sum = 0.0;
    for (int pr = p0; pr < p1; ++pr) sum += (double)pxu[(int64_t)pr * N + ia] + (double)pxu[(int64_t)pr * N + ib];
    p.psx[((int64_t)cq * p.n_cx + cx) * nb + k] = (float)(0.5 * sum);
}

// ---- output channels: one workgroup per (chunk, channel, class).  grid = n_chunks * n_ch * R ------------------
// AUTO: auto spectra only (ds_welch_psd): no input spectra, no cross sums.
// JIT (round 5 experiment, DSPTOOLBOX_AMD_WELCH_LONG_3PERCU=1): the cross loop without the next sequence in flight through
// the transform -- its sixteen loads are issued at the top of the pair's pass, three workgroups per CU cover the round trip.
template <bool AUTO = false, bool JIT = false>
__global__ __launch_bounds__(NT, (AUTO || JIT) ? 3 : 2) void k_yc(Args p) {  // (the cross loop needs 177 registers: the next sequence is 32 of them)
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    const int tid = threadIdx.x;
    const int r = (int)blockIdx.x & (p.R - 1);
    const int u = (int)blockIdx.x >> p.lgR;  // cq * n_ch + c
    const int cq = u / p.n_ch, c = u - cq * p.n_ch;
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    const int p0 = (int)((int64_t)cq * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(cq + 1) * p.n_pairs / p.n_chunks);
    float2 T[16];
    float P[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        T[j] = make_float2(0.f, 0.f);
        P[j] = 0.f;
    }
    // this channel's class-r sequences of the chunk, and the input spectra of the same pairs and class, as raw buffers
    // (32-bit offsets: a pair is R * 32 KB apart)
    const int64_t pair_stride = (int64_t)p.R * M;  // complex values
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        p.b + (((int64_t)c * p.n_pairs + p0) * p.R + r) * M, 0, (int)(uint32_t)(((int64_t)(p1 - p0 - 1) * pair_stride + M) * 8), 0x00020000);
    const int64_t xch = (int64_t)(p.n_cx > 1 ? c : 0) * p.n_pairs;
    const __amdgpu_buffer_rsrc_t xrs =
        AUTO ? brs
             : __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float2*>(p.xs) + ((xch + p0) * p.R + r) * M, 0,
                                                 (int)(uint32_t)(((int64_t)(p1 - p0 - 1) * pair_stride + M) * 8), 0x00020000);
    auto ld8 = [](__amdgpu_buffer_rsrc_t rs, int byte_off) {
        return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0));
    };
    float2 nx[16];
    if (!JIT && p0 < p1) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) nx[n1] = ld8(brs, 8 * (tid + 256 * n1));
    }
    w4::Stamp ts;
    __syncthreads();  // the W256 table
    const int pstep = (int)(pair_stride * 8);  // bytes between consecutive pairs (R <= 64: 2 MB)
    for (int pr = p0; pr < p1; ++pr) {
        float2 v[16];
        if (JIT) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = ld8(brs, (pr - p0) * pstep + 8 * (tid + 256 * n1));
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = nx[n1];
        }
        float2 xw[16];
        // the next pair's sequence (behind the chunk's last pair the range check returns zeros that nobody uses)
        const int boff = (pr + 1 - p0) * pstep + 8 * tid;
        const int xoff = (pr - p0) * pstep + 16 * tid;
        w4::fft4096_wi(
            v, tw, buf, tw2, tid,
            [&](int g) {
                if (!JIT) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) nx[4 * g + j] = ld8(brs, boff + 2048 * (4 * g + j));
                }
            },
            [&](int g) {
                if (!AUTO) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 q4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff + 4096 * (2 * g + j), 0, 0));
                        xw[2 * (2 * g + j)] = make_float2(q4.x, q4.y);
                        xw[2 * (2 * g + j) + 1] = make_float2(q4.z, q4.w);
                    }
                }
            },
            ts, 0);
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
    if (p.detrend && tid == 0 && r == 0) P[0] = 0.f;  // class 0, k' = 0 (the input spectrum's bin 0 is already 0 -> T = 0 there)
    const int bt = w4::bin_thread(tid);
    const int64_t o = ((int64_t)u * p.R + r) * M + bt;
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
        if (!AUTO) p.tu[o + 256 * k3] = T[k3];
        p.pu[o + 256 * k3] = P[k3];
    }
}

// ---- fold k <-> N - k of a chunk's class sums, one thread per bin: grid = (ceil(NB / 256), n_chunks * n_ch)
template <bool AUTO = false>
__global__ __launch_bounds__(256) void k_fold(Args p) {
    const int nb = p.R * (M / 2) + 1;
    const int64_t N = (int64_t)p.R * M;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nb) return;
    const int64_t u = blockIdx.y;  // cq * n_ch + c
    const int r = k & (p.R - 1), kp = (k >> p.lgR) & (M - 1), rm = (p.R - r) & (p.R - 1);
    const int64_t ia = (int64_t)r * M + kp, ib = (int64_t)rm * M + fold_index(r, kp);
    if (!AUTO) {
        const float2 a = p.tu[u * N + ia], b = p.tu[u * N + ib];
        p.pxy[u * nb + k] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
    }
    p.pyy[u * nb + k] = 0.5f * (p.pu[u * N + ia] + p.pu[u * N + ib]);
}

// ---- host side ------------------------------------------------------------------------
struct Plan {
    int n_pairs, n_chunks;
};
inline Plan plan(int n_frames, int n_cy, int R) {
    Plan pl;
    pl.n_pairs = (n_frames + 1) / 2;
    // three workgroups per CU resident at once (768 = chunks x channels x classes) where there is enough work;
    // fp32 accumulation chains stay <= 64 pairs
    int want = (768 + n_cy * R - 1) / (n_cy * R);
    const int by_len = (pl.n_pairs + 63) / 64;
    want = std::max(want, by_len);
    pl.n_chunks = std::max(1, std::min(want, pl.n_pairs));
    return pl;
}

}  // namespace welchl
