// STFT with frames of 2^15 ... 2^18 samples on the 4096-point register transform (reference: _stft,
// standard/_spectral_methods.py:176-282; output X[bin][frame][channel] complex64, the reference's layout).  gfx950.
// Round 4.
//
//   nfft = R x 4096, R = 8 ... 64.  kernels_stft4096.hpp's k_stft_dif folds the radix-2 / radix-4 stage into the sample
//   loads (every residue re-reads its samples: R - 1 passes over a frame).  Beyond radix 4 the stage gets its own pass,
//   the one kernels_welch_long.hpp uses for Welch:
//     k_sdif<R>   one thread per m < 4096 of a (frame, channel pair): z = w (u_c + i u_{c+1}), two neighbouring
//                 CHANNELS ride one complex sequence as in k_stft; an R-point DFT over s in registers
//                 (welchl::dft_small), the twiddle W_nfft^(r m) from welchl::host_tables, R complex values out:
//                     b_r[m] = ( sum_{s < R} z[m + 4096 s] W_R^(r s) ) W_nfft^(r m) ,   Z[R k' + r] = FFT4096(b_r)[k'];
//     k_stft_cls  k_stft_dif's structure on those sequences: ONE 1024-thread workgroup per CU = four teams, a
//                 workgroup owns 8 channels; the mirror of bin R k' + r is bin R (4095 - k') + (R - r), so
//                   unit kind 0:      the four teams = the four channel pairs, residue 0 and then residue R / 2 (which
//                                     mirror into themselves): 64-byte runs of the output;
//                   kinds 1 ... R - 2: residues r and R - r (r = 1 ... R / 2 - 1) of TWO channel pairs, teams (2 j,
//                                     2 j + 1) = (r, R - r) of pair j: 32-byte runs.
//   The class sequences of a group of frames live in the context's workspace (frame groups bound it to 512 MB).
//   Frames shorter than the transform are zero-padded by k_sdif (W < nfft, no detrend); with W == nfft removing the
//   frame mean only clears bin 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "kernels_stft4096.hpp"
#include "kernels_welch_long.hpp"

namespace stftl {

using welch4096::N;  // 4096
using stft4k::IMG;
using stft4k::NT;
using stft4k::TEAMS;
constexpr int LDS_BYTES = TEAMS * IMG * 8 + 256 * 8;

struct Args {
    const float* x;  // [n_ch][ld]
    int64_t n_samples, ld, pad_front;
    int n_ch, W, hop, n_frames, detrend, n_chunks, n_groups;  // groups of 16 channels
    int R, lgR;   // nfft = R * 4096
    int f0, nf;   // this launch's frames [f0, f0 + nf)
    const float* window;  // [W]
    const float2* twt;    // welch4096::host_tables()
    const float2* twl;    // welchl::host_tables(R)
    float scale, edge_scale;
    float2* b;    // [channel pair][nf][R][4096]
    float2* out;  // [nfft / 2 + 1][n_frames][n_ch]
};

inline int classes_of(int nfft) { return (nfft == 32768 || nfft == 65536 || nfft == 131072 || nfft == 262144) ? nfft / N : 0; }
// frames per launch group: class sequences <= 512 MB
inline int frames_per_group(int n_ch, int nfft, int n_frames) {
    const int64_t per_frame = (int64_t)((n_ch + 1) / 2) * nfft * 8;
    return (int)std::max<int64_t>(1, std::min<int64_t>(n_frames, ((int64_t)512 << 20) / per_frame));
}

// ---- pass 1.  grid = (16, nf, channel pairs) --------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_sdif(Args p) {
    const int m = (int)blockIdx.x * 256 + (int)threadIdx.x, fl = blockIdx.y, pc = blockIdx.z;
    const int c0 = 2 * pc;
    const bool two = c0 + 1 < p.n_ch;
    const float* __restrict__ xa = p.x + (int64_t)c0 * p.ld;
    const float* __restrict__ xb = p.x + (int64_t)(two ? c0 + 1 : c0) * p.ld;
    const int64_t i0 = (int64_t)(p.f0 + fl) * p.hop - p.pad_front + m;
    float2 z[R];
#pragma unroll
    for (int s = 0; s < R; ++s) {
        const int j = m + N * s;
        const int64_t i = i0 + (int64_t)N * s;
        const bool in = j < p.W && i >= 0 && i < p.n_samples;  // zero padding around the signal and behind the window
        const float w = in ? p.window[j] : 0.f;
        const float a = in ? xa[i] : 0.f;
        const float bq = (in && two) ? xb[i] : 0.f;
        z[s] = make_float2(a * w, bq * w);
    }
    welchl::dft_small<R>(z, p.twl + (size_t)R * N);
    float2* out = p.b + (((int64_t)pc * p.nf + fl) * R) * N + m;
#pragma unroll
    for (int r = 0; r < R; ++r) out[(int64_t)r * N] = r ? welch4096::cmul(z[r], p.twl[(size_t)r * N + m]) : z[0];
}

// workgroups: 2 halves x n_groups x n_chunks, rounded up to whole XCD rows (stft4k::grid_size)
template <bool POWER>
__global__ __launch_bounds__(NT) void k_stft_cls(Args p) {
    using namespace welch4096;
    extern __shared__ __align__(16) float2 lds[];
    float2* tw2 = lds + TEAMS * IMG;
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), tid = (int)threadIdx.x & 255;
    float2* buf = lds + team * IMG;
    // blockIdx -> (XCD, slot): the two halves of a 16-channel group are neighbouring slots of one XCD (k_stft)
    const int x = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3, half = s & 1, u = (s >> 1) * 8 + x;
    const int g = u % p.n_groups, q = u / p.n_groups;
    const int cb = 16 * g + 8 * half;  // first of the workgroup's 8 channels
    if (q >= p.n_chunks || cb >= p.n_ch) return;
    const int R = p.R, kinds = R - 1;
    const int n_units = p.nf * kinds;
    const int u0 = (int)((int64_t)q * n_units / p.n_chunks), u1 = (int)((int64_t)(q + 1) * n_units / p.n_chunks);
    if (u0 >= u1) return;

    if (team == 0) tw2[tid] = p.twt[15 * 256 + tid];
    __syncthreads();  // table

    const float sc = p.scale, sce = p.scale * p.edge_scale;
    const float pe = p.scale, pee = p.scale * p.edge_scale * p.edge_scale;
    const float dc = p.detrend ? 0.f : 1.f;
    const int64_t F = p.n_frames, C = p.n_ch;
    const bool wide = !(p.n_ch & 1);
    const int nfft_half = R * (N / 2);

    auto put = [&](int k, int f, int rc, bool r_two, float2 P, float2 Q, bool edge) {
        float2 A = make_float2(0.5f * (P.x + Q.x), 0.5f * (P.y - Q.y));
        float2 B = make_float2(0.5f * (P.y + Q.y), -0.5f * (P.x - Q.x));
        if (POWER) {
            const float e = edge ? (k == 0 ? pee * dc : pee) : pe;
            A = make_float2((A.x * A.x + A.y * A.y) * e, 0.f);
            B = make_float2((B.x * B.x + B.y * B.y) * e, 0.f);
        } else {
            const float e = edge ? (k == 0 ? sce * dc : sce) : sc;
            A = make_float2(A.x * e, A.y * e);
            B = make_float2(B.x * e, B.y * e);
        }
        float2* o = p.out + ((int64_t)k * F + f) * C + rc;
        if (wide) {
            *reinterpret_cast<float4*>(o) = make_float4(A.x, A.y, B.x, B.y);
        } else {
            o[0] = A;
            if (r_two) o[1] = B;
        }
    };
    // one transform per team (class `res` of channel pair `pair`, frame fl of the group) and the workgroup's read-out
    // of the four images.  cross == false: the four images are the four pairs, residue rr mirrors into itself;
    // cross == true: images (2 j, 2 j + 1) are residues (rr, R - rr) of pair 2 h + j.
    auto pass = [&](auto crossc, const int fl, const int rr, const int h) {
        constexpr bool cross = decltype(crossc)::value;
        int ty = (int)threadIdx.x;
        asm volatile("" : "+v"(ty));
        const int tl = ty & 255, bt_l = bin_thread(tl);
        const int pair = cross ? 2 * h + (team >> 1) : team;
        const int res = cross ? ((team & 1) ? R - rr : rr) : rr;
        const int c0 = cb + 2 * pair;
        float2 v[16];
        if (c0 < p.n_ch) {
            const float2* __restrict__ src = p.b + ((((int64_t)(c0 >> 1)) * p.nf + fl) * R + res) * N + tl;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = src[256 * n1];
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = make_float2(0.f, 0.f);
        }
        Tw6 tw;
        load_tw6(tw, p.twt, tl);
        fft4096_w(v, tw, buf, tw2, tl);
        __syncthreads();  // every wave has read its rows of the images
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) buf[fold_pos(bt_l + 256 * k3)] = v[pos16(k3)];
        __syncthreads();
        ty = (int)threadIdx.x;
        asm volatile("" : "+v"(ty));
        const int rk = ty >> 2, f = p.f0 + fl;
        if (!cross) {
            const int rp = ty & 3;
            const float2* im = lds + rp * IMG;
            const int rc = cb + 2 * rp;
            if (rc < p.n_ch) {
                const bool r_two = rc + 1 < p.n_ch;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = rk + 256 * j;  // k' < 2048
                    const int km = rr ? N - 1 - kk : (N - kk) & (N - 1);
                    put(R * kk + rr, f, rc, r_two, im[fold_pos(kk)], im[fold_pos(km)], rr == 0 && kk == 0);
                }
                if (rr == 0 && rk == 0) put(nfft_half, f, rc, r_two, im[fold_pos(N / 2)], im[fold_pos(N / 2)], true);
            }
        } else {
            const int rp2 = ty & 1, hi = (ty >> 1) & 1;
            const float2* own = lds + (2 * rp2 + hi) * IMG;
            const float2* oth = lds + (2 * rp2 + 1 - hi) * IMG;
            const int rc = cb + 2 * (2 * h + rp2);
            if (rc < p.n_ch) {
                const bool r_two = rc + 1 < p.n_ch;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = rk + 256 * j;
                    put(R * kk + (hi ? R - rr : rr), f, rc, r_two, own[fold_pos(kk)], oth[fold_pos(N - 1 - kk)], false);
                }
            }
        }
        // (the next transform's first barrier stands between these reads and its image stores)
    };
    for (int un = u0; un < u1; ++un) {
        const int fl = un / kinds, kind = un - fl * kinds;
        if (kind == 0) {
            pass(std::false_type{}, fl, 0, 0);
            pass(std::false_type{}, fl, R / 2, 0);
        } else {
            pass(std::true_type{}, fl, 1 + ((kind - 1) >> 1), (kind - 1) & 1);
        }
    }
}

}  // namespace stftl
