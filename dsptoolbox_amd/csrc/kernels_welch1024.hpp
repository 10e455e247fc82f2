// Welch H1/H2/H3 and auto spectra with windows of 256, 512 and 1024 samples (1024 is the
// reference's DEFAULT window,
// classes/signal.py:497-508; tests/test_transfer_functions.py:716-737 call
// compute_transfer_function with window_length_samples=1024) and ONE input channel: the algebra of
// kernels_welch4096.hpp on the wave-level transforms of kernels_stft1024.hpp (a team of L = N/16
// lanes per transform: 1, 2 or 4 teams per wave).  Written out below for N = 1024, L = 64.
//
//   k_x : one WAVE per pair of input frames (2p, 2p+1):
//           Wp = FFT1024( x_2p w + i x_2p+1 w )   -> xs[p][8][64] float4 (lane-major, L2 resident)
//           (|Wp[k]|^2 + |Wp[N-k]|^2)/2           -> px[p][0..512]
//   k_y : one wave per (chunk q of frame pairs, output channel c); the four waves of a workgroup
//         take four channels of the SAME chunk, so they walk the same input spectra together
//         (vector-L1 hits).  Per pair:  Zp = FFT1024( y_2p w + i y_2p+1 w ),
//           T[k] += conj(Wp[k]) Zp[k] ,  P[k] += |Zp[k]|^2   (all 1024 bins, 16 per lane)
//         folded k <-> N-k once per chunk.  No workgroup barrier inside the pair loop: a wave's LDS
//         exchanges are ordered by the hardware.  Each workgroup also sums a slice of the chunk's
//         px rows (fp64) -> psx[q].
//   k_welch_finish (kernels_finish.hpp): chunks -> H, coherence.
// Detrend = "skip bin 0" exactly as in kernels_welch4096.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_stft1024.hpp"

namespace welch1k {

using stft1k::fft_wave;
using stft1k::team_sync;
using stft1k::wave_sync;
constexpr int NTB = 256;  // threads per workgroup
template <int NN>
struct WG {
    using G = stft1k::Geo<NN>;
    static constexpr int N = NN, NB = NN / 2 + 1, L = G::L;
    static constexpr int TPB = NTB / L;  // teams (frame pairs / channels) per workgroup: 4, 8, 16
    static constexpr int REGION = G::REGION;
    static constexpr int LDS_BYTES = (TPB * REGION + G::TW_LEN) * 8 + NN * 4;  // + the window
};

struct Args {
    const float* sig;  // x (k_x) or y (k_y), planar
    int64_t n_samples, ld;
    int n_ch, hop, n_frames, n_pairs, detrend;
    int n_chunks, ppc;
    const float* window;
    const float2* twt;  // stft1k::host_tables<N>()
    float4* xs;         // [n_pairs][8][L]: lane t holds bins (t + L*2g, t + L*(2g+1))
    float* px;          // [n_pairs][NB]
    float2* pxy;        // [n_chunks][n_ch][NB]
    float* pyy;         // [n_chunks][n_ch][NB]
    float* psx;         // [n_chunks][n_cx][NB]
    int n_cx;           // input channels: 1 (shared by every output channel) or n_ch (one per output channel);
                        // then xs is [n_cx][n_pairs][8][L], px [n_cx][n_pairs][NB] and k_px_sum fills psx
};

// Raw samples of the frame pair (2p, 2p+1): HALF_HOP (hop == 512) -> 24 loads s[m] = ch[start +
// t + 64 m] (frame a: m 0..15, frame b: m 8..23); otherwise 32.  Interior pairs load
// unconditionally behind one wave-uniform test; the ragged tail clamps and selects.
template <bool HALF_HOP>
struct Raw {
    float s[HALF_HOP ? 24 : 32];
};
template <int NN, bool HALF_HOP>
__device__ __forceinline__ void load_raw(Raw<HALF_HOP>& r, const float* __restrict__ ch, int64_t n_samples,
                                         int64_t start0, int hop, int t) {
    constexpr int L = NN / 16;
    const float* __restrict__ src = ch + start0;
    const int64_t remain = n_samples - start0;
    const int span = HALF_HOP ? 3 * (NN / 2) : hop + NN;
    constexpr int CNT = HALF_HOP ? 24 : 32;
    if (remain >= span) {
        if (HALF_HOP) {
#pragma unroll
            for (int m = 0; m < 24; ++m) r.s[m] = src[t + L * m];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                r.s[n1] = src[t + L * n1];
                r.s[16 + n1] = src[hop + t + L * n1];
            }
        }
    } else {
        const int last = (int)(remain > (int64_t)(1 << 30) ? (1 << 30) : remain) - 1;  // >= 0
#pragma unroll
        for (int m = 0; m < CNT; ++m) {
            const int i = HALF_HOP ? t + L * m : (m < 16 ? t + L * m : hop + t + L * (m - 16));
            const float a = src[min(i, last)];
            r.s[m] = i <= last ? a : 0.f;
        }
    }
}
// The same samples through ONE raw-buffer descriptor over the whole planar signal: a 32-bit byte
// offset per load instead of 64-bit addresses (which the register cap of three workgroups per CU
// pushed into scratch: 120 bytes per lane for the 1024-point kernel, each reload behind an
// `s_waitcnt vmcnt(0)`).  Samples past the end come back as 0 from the hardware range check: a
// pair that reaches past n_samples gets out-of-range offsets where the sample does not exist.
template <int NN, bool HALF_HOP>
__device__ __forceinline__ void load_raw_buf(Raw<HALF_HOP>& r, __amdgpu_buffer_rsrc_t rs, uint32_t chan_samples,
                                             bool live, int64_t start0, int hop, int64_t n_samples, int t) {
    constexpr int L = NN / 16;
    constexpr int CNT = HALF_HOP ? 24 : 32;
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    const int span = HALF_HOP ? 3 * (NN / 2) : hop + NN;
    if (live && start0 + span <= n_samples) {
        const uint32_t o = (chan_samples + (uint32_t)start0 + (uint32_t)t) * 4u;
#pragma unroll
        for (int m = 0; m < CNT; ++m) {
            const uint32_t i = HALF_HOP ? L * m : (m < 16 ? L * m : hop + L * (m - 16));
            r.s[m] = stft1k::ld_f32(rs, o + 4u * i);
        }
    } else {
        // samples of this pair that exist (32-bit: the pair spans at most hop + NN samples)
        const int64_t rem = n_samples - start0;
        const int left = !live || rem <= 0 ? 0 : (rem > (int64_t)(1 << 20) ? (1 << 20) : (int)rem);
        // lane index against a SCALAR limit per load (a per-load lane value would be hoisted out of
        // the pair loop into 24 more registers)
        // (the lane index is laundered so that the per-load lane values of this rare path are
        // computed here and not hoisted out of the pair loop into registers the hot path needs)
        int tt = t;
        asm volatile("" : "+v"(tt));
        const uint32_t ot = (chan_samples + (uint32_t)start0 + (uint32_t)tt) * 4u;
#pragma unroll
        for (int m = 0; m < CNT; ++m) {
            const int im = HALF_HOP ? L * m : (m < 16 ? L * m : hop + L * (m - 16));  // wave-uniform
            r.s[m] = stft1k::ld_f32(rs, tt < left - im ? ot + 4u * (uint32_t)im : OOB);
        }
    }
}
inline bool buf_fits(int64_t n_samples, int n_ch, int64_t ld) {
    return ((int64_t)(n_ch - 1) * ld + n_samples) * 4 < ((int64_t)1 << 32) - ((int64_t)1 << 20);
}

template <int NN, bool HALF_HOP>
__device__ __forceinline__ void window_pair(float2 (&v)[16], const Raw<HALF_HOP>& r, const float* winl, int t) {
    constexpr int L = NN / 16;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const float b = HALF_HOP ? r.s[n1 + 8] : r.s[16 + n1];
        const float w = winl[t + L * n1];
        v[n1] = make_float2(r.s[n1] * w, b * w);
    }
}
// the last pair of an odd frame count when frame F would still overlap the signal
__device__ __forceinline__ bool needs_drop(const Args& p, int pr) {
    return pr == p.n_pairs - 1 && (p.n_frames & 1) && (int64_t)p.n_frames * p.hop < p.n_samples;
}

// twiddle tables and the window into LDS (the window is read from there at every pair: 16
// registers less per lane than keeping it)
template <int NN>
__device__ __forceinline__ void load_tables(float2* tw1, float* winl, const Args& p) {
    for (int i = threadIdx.x; i < stft1k::Geo<NN>::TW_LEN; i += NTB) tw1[i] = p.twt[i];
    for (int i = threadIdx.x; i < NN; i += NTB) winl[i] = p.window[i];
}

// ---- input spectra: grid = ceil(n_pairs / TPB) -----------------------------------
template <int NN, bool HALF_HOP>
__global__ __launch_bounds__(NTB) void k_x(Args p) {
    using W = WG<NN>;
    constexpr int L = W::L, NB = W::NB;
    extern __shared__ __align__(16) float2 lds[];
    // a 64-lane team is a wave: its index (and the LDS bases derived from it) is wave-uniform
    const int w = L == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)threadIdx.x / L;
    const int t = threadIdx.x % L;
    float2* buf = lds + w * W::REGION;
    float2* tw1 = lds + W::TPB * W::REGION;
    const float2* tw2 = tw1 + W::G::TW1;
    const int pr = blockIdx.x * W::TPB + w;
    const bool live = pr < p.n_pairs;
    const int cx = blockIdx.y;  // input channel
    Raw<HALF_HOP> raw;
    if (live) load_raw<NN, HALF_HOP>(raw, p.sig + (int64_t)cx * p.ld, p.n_samples, (int64_t)(2 * pr) * p.hop, p.hop, t);
    float* winl = reinterpret_cast<float*>(tw1 + W::G::TW_LEN);
    load_tables<NN>(tw1, winl, p);
    __syncthreads();
    if (!live) {
        if constexpr (L <= 64) return;  // (a team of two waves stays: the transform takes workgroup barriers)
#pragma unroll
        for (int m = 0; m < (HALF_HOP ? 24 : 32); ++m) raw.s[m] = 0.f;
    }
    float2 v[16], z[16];
    window_pair<NN, HALF_HOP>(v, raw, winl, t);
    if (live && needs_drop(p, pr)) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1].y = 0.f;
    }
    fft_wave<NN>(v, z, buf, tw1, tw2, t);
    if (p.detrend && t == 0) z[0] = make_float2(0.f, 0.f);
    if (live) {
        float4* xo = p.xs + ((int64_t)cx * p.n_pairs + pr) * (NN / 2) + t;
#pragma unroll
        for (int g = 0; g < 8; ++g) xo[L * g] = make_float4(z[2 * g].x, z[2 * g].y, z[2 * g + 1].x, z[2 * g + 1].y);
    }
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int m = 0; m < 16; ++m) pw[t + L * m] = z[m].x * z[m].x + z[m].y * z[m].y;
    team_sync<NN>();
    if (!live) return;
    float* po = p.px + ((int64_t)cx * p.n_pairs + pr) * NB;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = t + L * j;
        po[k] = 0.5f * (pw[k] + pw[(NN - k) & (NN - 1)]);
    }
    if (t == 0) po[NN / 2] = pw[NN / 2];
}

// ---- input auto spectra per chunk when every output channel has its own input channel:
// psx[q][cx][k] = sum over the chunk's pairs of px[cx][pair][k] (fp64).  grid = (n_chunks, n_cx).
// (With ONE input channel k_y's workgroups share this sum instead.)
template <int NN>
__global__ __launch_bounds__(256) void k_px_sum(Args p) {
    constexpr int NB = WG<NN>::NB;
    const int q = blockIdx.x, cx = blockIdx.y;
    const int p0 = (int)((int64_t)q * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(q + 1) * p.n_pairs / p.n_chunks);
    const float* __restrict__ px = p.px + (int64_t)cx * p.n_pairs * NB;
    for (int k = threadIdx.x; k < NB; k += 256) {
        double sum = 0.0;
        for (int pr = p0; pr < p1; ++pr) sum += (double)px[(int64_t)pr * NB + k];
        p.psx[((int64_t)q * p.n_cx + cx) * NB + k] = (float)sum;
    }
}

// ---- output channels: grid = n_chunks * ceil(n_ch / TPB) --------------------------
// AUTO: auto spectra only (Signal.get_spectrum's default call): no input spectra, no cross sums.
template <int NN, bool HALF_HOP, bool AUTO = false>
__global__ __launch_bounds__(NTB, 3) void k_y(Args p) {
    using W = WG<NN>;
    constexpr int L = W::L, NB = W::NB, TPB = W::TPB;
    extern __shared__ __align__(16) float2 lds[];
    const int w = L == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)threadIdx.x / L;
    const int t = threadIdx.x % L;
    float2* buf = lds + w * W::REGION;
    float2* tw1 = lds + TPB * W::REGION;
    const float2* tw2 = tw1 + W::G::TW1;
    const int n_grp = (p.n_ch + TPB - 1) / TPB;
    // XCD-aware decode (workgroups b, b+8, ... share an XCD and its L2): whole chunks per XCD, so
    // the input spectra a chunk's channel groups re-read stay in that L2
    int q, g;
    {
        const int b = blockIdx.x;
        if ((p.n_chunks & 7) == 0) {
            const int per = p.n_chunks >> 3;
            q = (b & 7) + 8 * ((b >> 3) % per);
            g = (b >> 3) / per;
        } else {
            q = b % p.n_chunks;
            g = b / p.n_chunks;
        }
    }
    const int c = g * TPB + w;
    const bool live = c < p.n_ch;
    // balanced split of the pairs over the chunks (n_chunks stays a multiple of 8 for the XCD mapping)
    const int p0 = (int)((int64_t)q * p.n_pairs / p.n_chunks), p1 = (int)((int64_t)(q + 1) * p.n_pairs / p.n_chunks);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.sig), 0, (int)(uint32_t)(((int64_t)(p.n_ch - 1) * p.ld + p.n_samples) * 4), 0x00020000);
    const uint32_t chan = live ? (uint32_t)((int64_t)c * p.ld) : 0u;
    Raw<HALF_HOP> raw;
    if (p0 < p1) load_raw_buf<NN, HALF_HOP>(raw, rs, chan, live, (int64_t)(2 * p0) * p.hop, p.hop, p.n_samples, t);
    float* winl = reinterpret_cast<float*>(tw1 + W::G::TW_LEN);
    load_tables<NN>(tw1, winl, p);
    if (!AUTO && p.n_cx <= 1) {
        // input auto-spectrum of this chunk: this workgroup's slice of the bins, px rows summed in fp64
        const int bpg = (NB + n_grp - 1) / n_grp;
        const int b0 = g * bpg, b1 = min(b0 + bpg, NB);
        for (int k = b0 + (int)threadIdx.x; k < b1; k += NTB) {
            double sum = 0.0;
            for (int pr = p0; pr < p1; ++pr) sum += (double)p.px[(int64_t)pr * NB + k];
            p.psx[(int64_t)q * NB + k] = (float)sum;
        }
    }
    __syncthreads();  // tables in LDS; the only workgroup barrier
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): keep the pre-loop loads out of the loop's wait counts
    if (!live) {
        if constexpr (L <= 64) return;  // a two-wave team stays (workgroup barriers ahead) and only skips its stores
    }
    float2 T[16];
    float P[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        T[j] = make_float2(0.f, 0.f);
        P[j] = 0.f;
    }
    // input spectra of this team's channel: the shared ones, or its own (an idle team reads channel 0)
    const float4* __restrict__ xs_ch = AUTO ? nullptr : p.xs + (int64_t)((p.n_cx > 1 && live) ? c : 0) * p.n_pairs * (NN / 2) + t;
    for (int pr = p0; pr < p1; ++pr) {
        float2 v[16], z[16];
        window_pair<NN, HALF_HOP>(v, raw, winl, t);
        if (needs_drop(p, pr)) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1].y = 0.f;
        }
        if (pr + 1 < p1)
            load_raw_buf<NN, HALF_HOP>(raw, rs, chan, live, (int64_t)(2 * pr + 2) * p.hop, p.hop, p.n_samples, t);
        if constexpr (AUTO) {
            fft_wave<NN>(v, z, buf, tw1, tw2, t);
#pragma unroll
            for (int m = 0; m < 16; ++m) P[m] = fmaf(z[m].x, z[m].x, fmaf(z[m].y, z[m].y, P[m]));
        } else {
            // this pair's input spectrum is requested behind the second exchange (registers of v free)
            float4 xq[8];
            auto issue_xs = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                const float4* __restrict__ xp = xs_ch + (int64_t)pr * (NN / 2);
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) xq[gg] = xp[L * gg];
                __builtin_amdgcn_sched_barrier(0);
            };
            fft_wave<NN>(v, z, buf, tw1, tw2, t, issue_xs);
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float2 xw = h ? make_float2(xq[gg].z, xq[gg].w) : make_float2(xq[gg].x, xq[gg].y);
                    const float2 zz = z[2 * gg + h];
                    float2& Tm = T[2 * gg + h];
                    Tm.x = fmaf(xw.x, zz.x, fmaf(xw.y, zz.y, Tm.x));  // conj(xw) zz
                    Tm.y = fmaf(xw.x, zz.y, fmaf(-xw.y, zz.x, Tm.y));
                    P[2 * gg + h] = fmaf(zz.x, zz.x, fmaf(zz.y, zz.y, P[2 * gg + h]));
                }
            }
        }
    }
    if (p.detrend && t == 0) P[0] = 0.f;  // xs bin 0 is already 0 -> T[0] = 0
    // fold k <-> N-k once per chunk through this team's LDS region
    const int64_t so = ((int64_t)q * p.n_ch + (live ? c : 0)) * NB;
    if (!AUTO) {
#pragma unroll
        for (int m = 0; m < 16; ++m) buf[t + L * m] = T[m];
        team_sync<NN>();
        if (live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = t + L * j;
                const float2 a = buf[k], b = buf[(NN - k) & (NN - 1)];
                p.pxy[so + k] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
            }
            if (t == 0) p.pxy[so + NN / 2] = make_float2(buf[NN / 2].x, 0.f);
        }
        team_sync<NN>();
    }
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int m = 0; m < 16; ++m) pw[t + L * m] = P[m];
    team_sync<NN>();
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = t + L * j;
        p.pyy[so + k] = 0.5f * (pw[k] + pw[(NN - k) & (NN - 1)]);
    }
    if (t == 0) p.pyy[so + NN / 2] = pw[NN / 2];
}

// ---- host side -------------------------------------------------------------------
struct Plan {
    int n_pairs, n_chunks, ppc;
    size_t bytes;
};
template <int NN>
inline Plan plan(int n_frames, int n_cy, int n_cx = 1, int want_chunks = 0) {  // want_chunks: ds_config::welch1k_chunks
    using W = WG<NN>;
    Plan pl;
    pl.n_pairs = (n_frames + 1) / 2;
    const int n_grp = (n_cy + W::TPB - 1) / W::TPB;
    // three workgroups per CU (768 on the 256 CUs) resident at once when there is enough work;
    // fp32 accumulation chains stay <= 64 pairs
    int want = (768 + n_grp - 1) / n_grp;
    want = (want + 7) & ~7;
    const int by_len = (pl.n_pairs + 63) / 64;
    if (want < by_len) want = (by_len + 7) & ~7;
    if (want_chunks > 0) want = want_chunks;
    want = std::max(1, std::min(want, pl.n_pairs));
    if (want >= 8) want &= ~7;  // whole chunks per XCD: the input spectra a chunk re-reads stay in one L2
    pl.n_chunks = want;         // chunk q = pairs [q n_pairs / n_chunks, (q+1) n_pairs / n_chunks)
    pl.ppc = (pl.n_pairs + want - 1) / want;
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    pl.bytes = pad(sizeof(float2) * (size_t)n_cx * pl.n_pairs * NN) + pad(sizeof(float) * (size_t)n_cx * pl.n_pairs * W::NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * W::NB) + pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * W::NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cy * W::NB);
    return pl;
}

}  // namespace welch1k
