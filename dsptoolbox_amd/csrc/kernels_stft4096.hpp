// STFT with 4096-, 8192- and 16384-point frames on the register-resident transform of kernels_welch4096w.hpp
// (reference: _stft, standard/_spectral_methods.py:196-283; output X[bin][frame][channel] complex64,
// the reference's own layout).  gfx950.
//
// The output is channel-fastest, so what a workgroup can write in one piece is decided by how many
// channels it transforms at once.  The wave-level kernels of kernels_stft1024.hpp put 16 channels into
// one workgroup (whole 128-byte lines); at 4096 points one transform needs 256 threads and a 34 KB
// exchange image, and a first version with one transform per workgroup (16-byte stores, every lane of a
// store instruction in another line) ran at 1.9 TB/s against 1.6 for the generic kernel.  Here:
//   * ONE 1024-thread workgroup per CU = FOUR teams of 256 threads; a team transforms TWO NEIGHBOURING
//     CHANNELS of one frame as one complex sequence z = u_c + i u_{c+1}, so the workgroup covers 8
//     channels of that frame: 64-byte runs of the output (four lanes x 16 bytes side by side);
//   * the two workgroups that share the 128-byte lines of 16 channels sit in neighbouring dispatch slots
//     of the SAME XCD (blockIdx -> (XCD, slot) -> (unit, half)), so a line is completed in one L2;
//   * after the transform every team writes its packed spectrum into its image in padded natural order
//     (as the chunk fold of k_y3); the read-out thread (pair p, bin k) separates
//     X_c[k] = (Z[k] + conj Z[N-k]) / 2, X_{c+1}[k] = (Z[k] - conj Z[N-k]) / (2i), scales and stores;
//   * a workgroup walks a chunk of consecutive frames; the samples of the next frame are requested
//     behind the second exchange of the current transform.
// 157 KB of LDS (4 images, the W256 table, the window), <= 128 registers (four waves per SIMD).
#pragma once
#include "kernels_welch4096w.hpp"

namespace stft4k {

using welch4096::L1S;
using welch4096::N;

constexpr int TEAMS = 4, NT = 256 * TEAMS;
constexpr int IMG = 16 * L1S + 16;  // complex per team image: + 128 bytes, so that the images of teams 0 / 1
                                    // (and 2 / 3) start in different halves of the 64 LDS banks
constexpr int LDS_BYTES = TEAMS * IMG * 8 + 256 * 8 + N * 4;

struct Args {
    const float* x;  // [n_ch][ld]
    int64_t n_samples, ld, pad_front;
    int n_ch, W, hop, n_frames, detrend, n_chunks, n_groups;  // groups of 16 channels
    const float* window;  // [W], W <= 4096 (shorter windows are zero-padded)
    const float2* twt;    // welch4096::host_tables()
    float scale, edge_scale;
    float2* out;  // [2049][n_frames][n_ch]
};

// workgroups: 2 halves x n_groups x n_chunks, rounded up to whole XCD rows
inline int grid_size(int n_groups, int n_chunks) { return 16 * ((n_groups * n_chunks + 7) / 8); }
// the byte offsets of the sample loads are 32-bit
inline bool fits(int64_t n_samples, int64_t pad_front) { return n_samples + pad_front + 4 * (int64_t)N < ((int64_t)1 << 29); }

template <bool POWER>
__global__ __launch_bounds__(NT) void k_stft(Args p) {
    using namespace welch4096;
    extern __shared__ __align__(16) float2 lds[];
    float2* tw2 = lds + TEAMS * IMG;
    float* winl = reinterpret_cast<float*>(tw2 + 256);
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), tid = (int)threadIdx.x & 255;
    float2* buf = lds + team * IMG;
    // blockIdx -> (XCD x, slot s of that XCD); the two halves of a unit are neighbouring slots of one XCD
    const int x = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3, half = s & 1, u = (s >> 1) * 8 + x;
    const int g = u % p.n_groups, q = u / p.n_groups;
    const int cb = 16 * g + 8 * half;  // first of the workgroup's 8 channels
    if (q >= p.n_chunks || cb >= p.n_ch) return;
    const int per = (p.n_frames + p.n_chunks - 1) / p.n_chunks;
    const int f0 = q * per, f1 = min(f0 + per, p.n_frames);
    if (f0 >= f1) return;
    const int c0 = cb + 2 * team;  // this team's channel pair (teams past the last channel transform zeros)
    const bool one = c0 < p.n_ch, two = c0 + 1 < p.n_ch;

    Tw6 tw;
    load_tw6(tw, p.twt, tid);
    if (team == 0) tw2[tid] = p.twt[15 * 256 + tid];
    for (int i = (int)threadIdx.x; i < N; i += NT) winl[i] = i < p.W ? p.window[i] : 0.f;
    // samples outside [0, n_samples) read as zero through the range check of the buffer loads: the zero
    // padding in front (pad_front) and behind the signal, and channels past the last one
    const __amdgpu_buffer_rsrc_t ra = channel_rsrc(p.x + (int64_t)(one ? c0 : 0) * p.ld, one ? p.n_samples : 0);
    const __amdgpu_buffer_rsrc_t rb = channel_rsrc(p.x + (int64_t)(two ? c0 + 1 : 0) * p.ld, two ? p.n_samples : 0);
    float sa[16], sb[16];
    auto load = [&](int f) {
        const int off = 4 * ((int)((int64_t)f * p.hop - p.pad_front) + tid);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            sa[n1] = ld_sample(ra, off + 1024 * n1);
            sb[n1] = ld_sample(rb, off + 1024 * n1);
        }
    };
    load(f0);
    __syncthreads();  // tables and window

    const float sc = p.scale, sce = p.scale * p.edge_scale;
    const float pe = p.scale, pee = p.scale * p.edge_scale * p.edge_scale;  // power mode
    const float dc = p.detrend ? 0.f : 1.f;  // W == 4096: removing the frame mean only clears bin 0
    const int64_t F = p.n_frames, C = p.n_ch;
    const bool wide = !(p.n_ch & 1);  // 16-byte stores need an even channel count (then a live pair has both channels)

    for (int f = f0; f < f1; ++f) {
        // every per-thread index is re-derived from the thread number once per frame: kept across the loop (or
        // hoisted out of it as loop-invariant addresses) they cost ~45 registers over the 128 this kernel has
        int tx = (int)threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int tid_l = tx & 255, bt_l = bin_thread(tid_l);
        // read-out: thread -> (pair rp = lane & 3, bin row rk = thread / 4): four neighbouring lanes write 64 bytes
        const int rp = tx & 3, rk_l = tx >> 2;
        const float2* rbuf = lds + rp * IMG;
        const int rc = cb + 2 * rp;
        const bool r_one = rc < p.n_ch, r_two = rc + 1 < p.n_ch;
        float2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const float w = winl[tid_l + 256 * n1];
            v[n1] = make_float2(sa[n1] * w, sb[n1] * w);
        }
        const bool more = f + 1 < f1;
        fft4096_w(v, tw, buf, tw2, tid_l, NoHook(), [&]() {
            if (more) load(f + 1);
        });
        __syncthreads();  // every wave has read its rows of the images
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) buf[fold_pos(bt_l + 256 * k3)] = v[pos16(k3)];
        __syncthreads();
        if (r_one) {
            float2* o = p.out + ((int64_t)rk_l * F + f) * C + rc;
            const int64_t ostep = 256 * F * C;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = rk_l + 256 * j;
                const float2 P = rbuf[fold_pos(k)], Q = rbuf[fold_pos((N - k) & (N - 1))];
                float2 A = make_float2(0.5f * (P.x + Q.x), 0.5f * (P.y - Q.y));
                float2 B = make_float2(0.5f * (P.y + Q.y), -0.5f * (P.x - Q.x));
                if (POWER) {
                    const float e = (j == 0 && rk_l == 0) ? pee * dc : pe;
                    A = make_float2((A.x * A.x + A.y * A.y) * e, 0.f);
                    B = make_float2((B.x * B.x + B.y * B.y) * e, 0.f);
                } else {
                    const float e = (j == 0 && rk_l == 0) ? sce * dc : sc;
                    A = make_float2(A.x * e, A.y * e);
                    B = make_float2(B.x * e, B.y * e);
                }
                if (wide) {
                    *reinterpret_cast<float4*>(o + j * ostep) = make_float4(A.x, A.y, B.x, B.y);
                } else {
                    o[j * ostep] = A;
                    if (r_two) o[j * ostep + 1] = B;
                }
            }
            if (rk_l == 0) {  // bin N/2 pairs with itself
                const float2 P = rbuf[fold_pos(N / 2)];
                float2 A, B;
                if (POWER) {
                    A = make_float2(P.x * P.x * pee, 0.f);
                    B = make_float2(P.y * P.y * pee, 0.f);
                } else {
                    A = make_float2(P.x * sce, 0.f);
                    B = make_float2(P.y * sce, 0.f);
                }
                float2* oe = p.out + ((int64_t)(N / 2) * F + f) * C + rc;
                oe[0] = A;
                if (r_two) oe[1] = B;
            }
        }
        // (the next transform's first barrier stands between these reads and its image stores)
    }
}

// ---- 8192 and 16384 points ------------------------------------------------------------------------------
// nfft = SUB x 4096 (SUB = 2, 4): a team transforms the SUB decimated sequences z_r[m] = z[r + SUB m] of its
// channel pair one after the other, each into its own image (the image is the transform's exchange area
// first and holds its packed spectrum S_r in padded natural order afterwards), and the read-out combines
//   Z[k] = sum_r w^r S_r[k mod 4096],   Z[N-k] = sum_r conj(w)^r S_r[(4096 - k) mod 4096],   w = exp(-2 pi i k / N)
// per output bin (w from sincospi, its powers by multiplication) before the same separation as above.
// A team's images are SUB x 34 KB: two teams (4 channels, 32-byte runs) at 8192 points, one team (16-byte
// runs) at 16384; the 16 / (2 TEAMS) workgroups of a 16-channel group are neighbouring slots of one XCD.
// The window is read from global memory (it would be 32 / 64 KB of LDS).
template <int SUB>
struct Long {
    static constexpr int NFFT = SUB * N, TEAMS = SUB == 2 ? 2 : 1, NT = 256 * TEAMS;
    static constexpr int TEAM_C = SUB * IMG + (SUB == 2 ? 16 : 0);  // complex per team (teams in different bank halves)
    static constexpr int LDS_BYTES = TEAMS * TEAM_C * 8 + 256 * 8;
    static constexpr int WPG = 8 / TEAMS;  // workgroups per group of 16 channels
    static int grid_size(int n_groups, int n_chunks) { return 8 * WPG * ((n_groups * n_chunks + 7) / 8); }
};
inline bool fits_long(int64_t n_samples, int64_t pad_front, int nfft) {
    return n_samples + pad_front + 4 * (int64_t)nfft < ((int64_t)1 << 29);
}

template <int SUB, bool POWER>
__global__ __launch_bounds__(Long<SUB>::NT) void k_stft_long(Args p) {
    using namespace welch4096;
    using G = Long<SUB>;
    constexpr int NFFT = G::NFFT, TEAMS = G::TEAMS;
    extern __shared__ __align__(16) float2 lds[];
    float2* tw2 = lds + TEAMS * G::TEAM_C;
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), tid = (int)threadIdx.x & 255;
    float2* img = lds + team * G::TEAM_C;
    const int x = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3, sub = s % G::WPG, u = (s / G::WPG) * 8 + x;
    const int g = u % p.n_groups, q = u / p.n_groups;
    const int cb = 16 * g + 2 * TEAMS * sub;  // first of the workgroup's 2 TEAMS channels
    if (q >= p.n_chunks || cb >= p.n_ch) return;
    const int per = (p.n_frames + p.n_chunks - 1) / p.n_chunks;
    const int f0 = q * per, f1 = min(f0 + per, p.n_frames);
    if (f0 >= f1) return;
    const int c0 = cb + 2 * team;
    const bool one = c0 < p.n_ch, two = c0 + 1 < p.n_ch;

    Tw6 tw;
    load_tw6(tw, p.twt, tid);
    if (team == 0) tw2[tid] = p.twt[15 * 256 + tid];
    const __amdgpu_buffer_rsrc_t ra = channel_rsrc(p.x + (int64_t)(one ? c0 : 0) * p.ld, one ? p.n_samples : 0);
    const __amdgpu_buffer_rsrc_t rb = channel_rsrc(p.x + (int64_t)(two ? c0 + 1 : 0) * p.ld, two ? p.n_samples : 0);
    const __amdgpu_buffer_rsrc_t rw = channel_rsrc(p.window, p.W);  // zero past the window: zero-padded frames
    float sa[16], sb[16], sw[16];
    auto load = [&](int f, int r) {  // z_r[m], m = tid + 256 n1: sample r + SUB m of the frame
        const int i0 = r + SUB * tid;
        const int off = 4 * ((int)((int64_t)f * p.hop - p.pad_front) + i0);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            sa[n1] = ld_sample(ra, off + 1024 * SUB * n1);
            sb[n1] = ld_sample(rb, off + 1024 * SUB * n1);
            sw[n1] = ld_sample(rw, 4 * i0 + 1024 * SUB * n1);
        }
    };
    load(f0, 0);
    __syncthreads();  // table

    const float sc = p.scale, sce = p.scale * p.edge_scale;
    const float pe = p.scale, pee = p.scale * p.edge_scale * p.edge_scale;
    const float dc = p.detrend ? 0.f : 1.f;  // W == nfft: removing the frame mean only clears bin 0
    const int bt = bin_thread(tid);
    const int64_t F = p.n_frames, C = p.n_ch;
    const bool wide = !(p.n_ch & 1);
    // read-out: thread -> (pair rp, bin row rk): TEAMS neighbouring lanes write 16 TEAMS bytes
    const int rp = (int)threadIdx.x & (TEAMS - 1), rk = (int)threadIdx.x / TEAMS;
    const float2* rimg = lds + rp * G::TEAM_C;
    const int rc = cb + 2 * rp;
    const bool r_one = rc < p.n_ch, r_two = rc + 1 < p.n_ch;

    for (int f = f0; f < f1; ++f) {
#pragma unroll
        for (int r = 0; r < SUB; ++r) {
            float2 v[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = make_float2(sa[n1] * sw[n1], sb[n1] * sw[n1]);
            const bool more = r + 1 < SUB || f + 1 < f1;
            float2* buf = img + r * IMG;
            fft4096_w(v, tw, buf, tw2, tid, NoHook(), [&]() {
                if (more) load(r + 1 < SUB ? f : f + 1, r + 1 < SUB ? r + 1 : 0);
            });
            __syncthreads();  // every wave has read its rows of the image
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) buf[fold_pos(bt + 256 * k3)] = v[pos16(k3)];
        }
        __syncthreads();
        if (r_one) {
            auto put = [&](int k, float2 A, float2 B, bool edge) {
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
            auto bin = [&](int k) {
                const int km = k & (N - 1), kmm = (N - km) & (N - 1);
                float sn, cs;
                sincospif(-2.0f * (float)k / (float)NFFT, &sn, &cs);
                const float2 w = make_float2(cs, sn);
                float2 P = rimg[fold_pos(km)], Q = rimg[fold_pos(kmm)];
                float2 wr = w;
#pragma unroll
                for (int r = 1; r < SUB; ++r) {
                    const float2 a = rimg[r * IMG + fold_pos(km)], b = rimg[r * IMG + fold_pos(kmm)];
                    P.x += wr.x * a.x - wr.y * a.y;  // + w^r S_r[km]
                    P.y += wr.x * a.y + wr.y * a.x;
                    Q.x += wr.x * b.x + wr.y * b.y;  // + conj(w)^r S_r[kmm]
                    Q.y += wr.x * b.y - wr.y * b.x;
                    if (r + 1 < SUB) wr = cmul(wr, w);
                }
                const float2 A = make_float2(0.5f * (P.x + Q.x), 0.5f * (P.y - Q.y));
                const float2 B = make_float2(0.5f * (P.y + Q.y), -0.5f * (P.x - Q.x));
                put(k, A, B, k == 0 || k == NFFT / 2);
            };
            for (int j = 0; j < NFFT / 512; ++j) bin(rk + 256 * j);
            if (rk == 0) bin(NFFT / 2);
        }
        // (the next transform's first barrier stands between these reads and its image stores)
    }
}

}  // namespace stft4k
