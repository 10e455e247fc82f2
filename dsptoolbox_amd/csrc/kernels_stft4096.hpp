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
#include <cmath>
#include <vector>

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
    float2* out;  // [nfft / 2 + 1][n_frames][n_ch]
    const float2* twn;  // k_stft_dif: W_nfft^j, j < nfft (host_twiddles)
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

inline bool fits_long(int64_t n_samples, int64_t pad_front, int nfft) {
    return n_samples + pad_front + 4 * (int64_t)nfft < ((int64_t)1 << 29);
}

// ---- 8192 and 16384 points, decimation in frequency -----------------------------------------------------------
// nfft = SUB x 4096.  One radix-SUB stage on the windowed samples, y_r[m] = (sum_s z[m + 4096 s] (-i)^(r s)) W_nfft^(r m),
// makes the 4096-point transform of y_r the bins k = SUB k' + r of the frame's spectrum -- final values, and the
// mirror bin nfft - k lies in residue (SUB - r) mod SUB: residues 0 and SUB/2 pair with THEMSELVES, so a team needs only
// the one image of the 4096 kernel and the workgroup keeps its FOUR teams = 8 channels = 64-byte runs:
//   8192:  phase 0: r = 0, phase 1: r = 1, the four teams = the four channel pairs in both;
//   16384: phases 0, 1: r = 0, 2 as above; phases 2, 3: residues 1 and 3 pair with EACH OTHER: teams (2 j, 2 j + 1)
//          transform residues 1 and 3 of channel pair j (phase 2) or 2 + j (phase 3): 32-byte runs for that half
//          of the bins.
// Every phase is one transform per team and one read-out; the samples (and window values, from global memory: the
// LDS holds the four images) of a phase are loaded in eight batches, one batch ahead of their use.  twn = W_nfft^j, j < nfft.
// (A decimation-in-TIME form -- the SUB sub-spectra of a channel pair kept in SUB images and combined per output
// bin at the read-out -- needs 70 / 140 KB per team, i.e. two teams / one team per workgroup and 32- / 16-byte runs:
// 0.252 / 0.279 ms on 64 x 512 000 samples where this form takes 0.182 / 0.263.)
template <int SUB>
struct Dif {
    static constexpr int NFFT = SUB * N;
    static constexpr int KINDS = SUB == 2 ? 1 : 3;  // work units per frame (k_stft_dif)
    static constexpr int LDS_BYTES = TEAMS * IMG * 8 + 256 * 8;
};
inline void host_twiddles(int nfft, std::vector<float2>& h) {
    h.resize((size_t)nfft);
    for (int j = 0; j < nfft; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)nfft;
        h[(size_t)j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
}

template <int SUB, bool POWER>
__global__ __launch_bounds__(NT) void k_stft_dif(Args p) {
    using namespace welch4096;
    constexpr int NFFT = SUB * N;
    constexpr int KINDS = Dif<SUB>::KINDS;  // units per frame: the self-mirroring residues, then the cross phases
    extern __shared__ __align__(16) float2 lds[];
    float2* tw2 = lds + TEAMS * IMG;
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), tid = (int)threadIdx.x & 255;
    float2* buf = lds + team * IMG;
    const int x = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3, half = s & 1, u = (s >> 1) * 8 + x;
    const int g = u % p.n_groups, q = u / p.n_groups;
    const int cb = 16 * g + 8 * half;  // first of the workgroup's 8 channels
    if (q >= p.n_chunks || cb >= p.n_ch) return;
    // the chunk's share of the (frame, kind) units
    const int n_units = p.n_frames * KINDS;
    const int u0 = (int)((int64_t)q * n_units / p.n_chunks), u1 = (int)((int64_t)(q + 1) * n_units / p.n_chunks);
    if (u0 >= u1) return;

    if (team == 0) tw2[tid] = p.twt[15 * 256 + tid];
    const __amdgpu_buffer_rsrc_t rw = channel_rsrc(p.window, p.W);  // zero past the window: zero-padded frames
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.twn), 0, NFFT * 8, 0x00020000);
    __syncthreads();  // table

    const float sc = p.scale, sce = p.scale * p.edge_scale;
    const float pe = p.scale, pee = p.scale * p.edge_scale * p.edge_scale;
    const float dc = p.detrend ? 0.f : 1.f;  // W == nfft: removing the frame mean only clears bin 0
    const int64_t F = p.n_frames, C = p.n_ch;
    const bool wide = !(p.n_ch & 1);

    // kind 0: residues 0 and SUB/2 of the team's own channel pair from ONE pass over the samples (a sample load
    // costs more than its share of the arithmetic: with half of them removed a phase took 30 % less), two transforms;
    // kinds 1, 2 (16384 points): residue 1 (even teams) or 3 (odd teams) of channel pair (team >> 1) + 2 (kind - 1)
    auto unit = [&](auto kc, const int f) {
        constexpr int kind = decltype(kc)::value;
        constexpr bool cross = kind > 0;
        // per-thread indices are re-derived per unit (see k_stft)
        int tx = (int)threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int tid_l = tx & 255;
        const int r = cross ? ((team & 1) ? 3 : 1) : 0;
        const int pair = cross ? 2 * (kind - 1) + (team >> 1) : team;
        const int c0 = cb + 2 * pair;
        const bool one = c0 < p.n_ch, two = c0 + 1 < p.n_ch;
        const __amdgpu_buffer_rsrc_t ra = channel_rsrc(p.x + (int64_t)(one ? c0 : 0) * p.ld, one ? p.n_samples : 0);
        const __amdgpu_buffer_rsrc_t rb = channel_rsrc(p.x + (int64_t)(two ? c0 + 1 : 0) * p.ld, two ? p.n_samples : 0);
        const int off0 = 4 * ((int)((int64_t)f * p.hop - p.pad_front) + tid_l);
        constexpr int RT = cross ? 0 : SUB / 2;  // the second residue of kind 0 (its twiddle W^(RT m))
        float2 v[16], v2[cross ? 1 : 16];
        // eight batches of two values, the loads of batch b + 1 requested before batch b is combined
        struct Batch {
            float za[2][SUB], zb[2][SUB], zw[2][SUB];
            float2 wn[2];
        };
        auto load = [&](Batch& qb, int b2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n1 = 2 * b2 + i;
#pragma unroll
                for (int sI = 0; sI < SUB; ++sI) {
                    const int o = 1024 * n1 + 4 * N * sI;
                    qb.za[i][sI] = ld_sample(ra, off0 + o);
                    qb.zb[i][sI] = ld_sample(rb, off0 + o);
                    qb.zw[i][sI] = ld_sample(rw, 4 * tid_l + o);
                }
                const int ot = 8 * (cross ? r : RT) * (tid_l + 256 * n1);
                qb.wn[i] = make_float2(ld_sample(rt, ot), ld_sample(rt, ot + 4));
            }
        };
        auto combine = [&](const Batch& qb, int b2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float2 z[SUB];
#pragma unroll
                for (int sI = 0; sI < SUB; ++sI) z[sI] = make_float2(qb.za[i][sI] * qb.zw[i][sI], qb.zb[i][sI] * qb.zw[i][sI]);
                constexpr int S2 = SUB == 4 ? 2 : 0, S3 = SUB == 4 ? 3 : 0;  // (indices valid for SUB == 2 too)
                if (!cross) {
                    float2 e0, e1;
                    if (SUB == 2) {
                        e0 = z[0], e1 = z[1];
                    } else {
                        e0 = make_float2(z[0].x + z[S2].x, z[0].y + z[S2].y);
                        e1 = make_float2(z[1].x + z[S3].x, z[1].y + z[S3].y);
                    }
                    v[2 * b2 + i] = make_float2(e0.x + e1.x, e0.y + e1.y);                               // residue 0
                    float2 y2 = cmul(make_float2(e0.x - e1.x, e0.y - e1.y), qb.wn[i]);  // residue SUB/2
                    asm volatile("" : "+v"(y2.x), "+v"(y2.y));  // parked as the product, not as its two factors
                    v2[cross ? 0 : 2 * b2 + i] = y2;
                } else {  // r = 1: (z0 - z2) - i (z1 - z3);  r = 3: (z0 - z2) + i (z1 - z3)
                    const float2 d0 = make_float2(z[0].x - z[S2].x, z[0].y - z[S2].y);
                    const float2 d1 = make_float2(z[1].x - z[S3].x, z[1].y - z[S3].y);
                    const float2 y = r == 1 ? make_float2(d0.x + d1.y, d0.y - d1.x) : make_float2(d0.x - d1.y, d0.y + d1.x);
                    v[2 * b2 + i] = cmul(y, qb.wn[i]);
                }
            }
        };
        {
            Batch cur;
#pragma unroll
            for (int b2 = 0; b2 < 8; ++b2) {
                load(cur, b2);
                __builtin_amdgcn_sched_barrier(0);
                combine(cur, b2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        auto put = [&](int k, int rc, bool r_two, float2 P, float2 Q, bool edge) {
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
        // one transform of the team's 16 values and the workgroup's read-out of the four images
        auto pass = [&](auto rrc) {  // (transforms v: handed over as an argument, the arrays ended up in scratch)
            constexpr int rr = decltype(rrc)::value;  // kind 0: the residue (0 or SUB/2); cross: unused
            int ty = (int)threadIdx.x;
            asm volatile("" : "+v"(ty));
            const int tl = ty & 255, bt_l = bin_thread(tl);
            Tw6 tw;  // six 8-byte loads per transform: held across the unit they are 12 registers this kernel does not have
            load_tw6(tw, p.twt, tl);
            fft4096_w(v, tw, buf, tw2, tl);
            __syncthreads();  // every wave has read its rows of the images
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) buf[fold_pos(bt_l + 256 * k3)] = v[pos16(k3)];
            __syncthreads();
            // the read-out's indices and output addresses are derived HERE: from a value known before the transform
            // they are computed before it and held through it (70 dwords of scratch)
            ty = (int)threadIdx.x;
            asm volatile("" : "+v"(ty));
            const int rk = ty >> 2;
            if (!cross) {  // four pairs, residue rr mirrors into itself
                const int rp = ty & 3;
                const float2* im = lds + rp * IMG;
                const int rc = cb + 2 * rp;
                if (rc < p.n_ch) {
                    const bool r_two = rc + 1 < p.n_ch;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kk = rk + 256 * j;  // k' < 2048
                        const int km = rr ? N - 1 - kk : (N - kk) & (N - 1);
                        put(SUB * kk + rr, rc, r_two, im[fold_pos(kk)], im[fold_pos(km)], rr == 0 && kk == 0);
                    }
                    if (rr == 0 && rk == 0) put(NFFT / 2, rc, r_two, im[fold_pos(N / 2)], im[fold_pos(N / 2)], true);
                }
            } else {  // two pairs x residues 1 and 3, which mirror into each other
                const int rp2 = ty & 1, res3 = (ty >> 1) & 1;
                const float2* own = lds + (2 * rp2 + res3) * IMG;
                const float2* oth = lds + (2 * rp2 + 1 - res3) * IMG;
                const int rc = cb + 2 * (2 * (kind - 1) + rp2);
                if (rc < p.n_ch) {
                    const bool r_two = rc + 1 < p.n_ch;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kk = rk + 256 * j;
                        put(SUB * kk + (res3 ? 3 : 1), rc, r_two, own[fold_pos(kk)], oth[fold_pos(N - 1 - kk)], false);
                    }
                }
            }
            // (the next transform's first barrier stands between these reads and its image stores)
        };
        pass(std::integral_constant<int, 0>{});
        if constexpr (!cross) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = v2[i];
            pass(std::integral_constant<int, SUB / 2>{});
        }
    };
    for (int un = u0; un < u1; ++un) {
        const int f = un / KINDS, kind = un - f * KINDS;
        if (kind == 0)
            unit(std::integral_constant<int, 0>{}, f);
        else if (KINDS > 1 && kind == 1)
            unit(std::integral_constant<int, (KINDS > 1 ? 1 : 0)>{}, f);
        else if (KINDS > 2)
            unit(std::integral_constant<int, (KINDS > 2 ? 2 : 0)>{}, f);
    }
}

// ---- inverse STFT, 4096-point frames at 50 % overlap -----------------------------------------------------------
// (reference: transforms.istft, transforms/transforms.py:444-586; the overlap-add's semantics: dsk::k_istft_fused.)
// The mirror image of k_stft: a team handles TWO NEIGHBOURING CHANNELS of ONE frame -- Z = X_c + i X_{c+1} built from
// the one-sided spectra, x_c + i x_{c+1} = ifft(Z) = conj(fft(conj Z)) on fft4096_w --, the workgroup's four teams read
// the bins of 8 channels together (64-byte runs of the channel-fastest spectrogram) into the four images.  A thread
// ends with samples n = bt + 256 k3 of both channels; k3 < 8 is the first half of the frame, so the overlap-add with
// the frame before is thread-local (8 + 8 carried sums); the finished half frame goes through the image once more
// (as (c, c + 1) pairs in padded natural order) and leaves as 1 KB runs per channel.  A workgroup walks a chunk of
// frames, after the frame in front of it (for the carry).
constexpr int ISTFT_LDS_BYTES = TEAMS * IMG * 8 + 256 * 8 + (N / 2) * 4;  // images, W256, 1 / envelope

// (q.a.tw = welch4096::host_tables(), q.a.fpw = the number of frame chunks; WIDE: an even channel count, one 16-byte load
// per bin and channel pair; the spectrogram is smaller than 4 GB, the host checks)
template <bool WIDE>
__global__ __launch_bounds__(NT) void k_istft(dsk::IstftFusedArgs q) {
    using namespace welch4096;
    constexpr int STEP = N / 2;
    const dsk::IstftArgs& p = q.a;
    extern __shared__ __align__(16) float2 lds[];
    float2* tw2 = lds + TEAMS * IMG;
    float* inv_env = reinterpret_cast<float*>(tw2 + 256);
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), tid = (int)threadIdx.x & 255;
    float2* buf = lds + team * IMG;
    // blockIdx -> (XCD, slot): the two halves of a 16-channel group are neighbouring slots of one XCD (k_stft)
    const int x = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3, half = s & 1, u = (s >> 1) * 8 + x;
    const int n_groups = (p.n_ch + 15) / 16;
    const int g = u % n_groups, qc = u / n_groups;  // qc: chunk of frames
    const int cb = 16 * g + 8 * half;
    const int n_chunks = p.fpw;  // (the host passes the number of chunks here)
    if (qc >= n_chunks || cb >= p.n_ch) return;
    const int per = (p.n_frames + n_chunks - 1) / n_chunks;
    const int fa = qc * per, fb = min(fa + per, p.n_frames);
    if (fa >= fb) return;
    const int c0 = cb + 2 * team;
    const bool one = c0 < p.n_ch, two = c0 + 1 < p.n_ch;

    if (team == 0) tw2[tid] = p.tw[15 * 256 + tid];
    for (int m = (int)threadIdx.x; m < STEP; m += NT) {
        const double w0 = (double)p.window[m], w1 = (double)p.window[m + STEP];
        const double e = w0 * w0 + w1 * w1;
        inv_env[m] = (float)(1.0 / (e < 1e-4 ? 1e-4 : e));
    }
    const int bt = bin_thread(tid);
    const int64_t F = p.n_frames, C = p.n_ch;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float2*>(p.stft), 0, (int)(uint32_t)((int64_t)p.n_bins * F * C * 8), 0x00020000);
    float ca[8], cbv[8];  // second half of the frame before, channels c0 and c0 + 1
#pragma unroll
    for (int j = 0; j < 8; ++j) ca[j] = cbv[j] = 0.f;
    float* oa = q.out + (int64_t)(one ? c0 : 0) * q.ld;
    float* ob = q.out + (int64_t)(two ? c0 + 1 : 0) * q.ld;
    auto emit = [&](float* o, int64_t pos, float sum, bool fast, int n) {
        if (pos < 0 || pos >= q.total_length) return;
        if (fast) {
            o[pos] = sum * inv_env[n];
            return;
        }
        const int64_t fs = pos / STEP;
        const int m = (int)(pos - fs * STEP);
        double env = 0.0;
        if (fs < q.n_total) {
            const float w = p.window[m];
            env += (double)w * (double)w;
        }
        if (fs >= 1 && fs - 1 < q.n_total) {
            const float w = p.window[m + STEP];
            env += (double)w * (double)w;
        }
        o[pos] = (float)((double)sum / (env < 1e-4 ? 1e-4 : env));
    };
    // positions in front of the first frame slot
    if (fa == 0)
        for (int64_t n = tid; n < (int64_t)q.off * STEP; n += 256) {
            if (one) emit(oa, n, 0.f, false, 0);
            if (two) emit(ob, n, 0.f, false, 0);
        }
    for (int f = fa > 0 ? fa - 1 : 0; f < fb; ++f) {
        const bool owned = f >= fa;
        int tx = (int)threadIdx.x;
        asm volatile("" : "+v"(tx));
        __syncthreads();  // tables / the previous frame's read-out
        {   // load: thread -> (pair rp, bin row rk); conj(Z) and its mirror half into image rp, padded natural order.
            // All nine loads are requested before the first value is placed, as raw-buffer loads whose offset lies
            // behind the end where there is nothing to read (zero): until round 4 each was a plain load inside
            // `if (channel and bin exist)`, which hipcc follows with a full wait -- nine memory round trips per frame
            // with the sixteen waves of the CU in step.
            const int rp = tx & 3, rk = tx >> 2;
            float2* im = lds + rp * IMG;
            const int rc = cb + 2 * rp;
            const bool r_one = rc < p.n_ch, r_two = rc + 1 < p.n_ch;
            auto fetch = [&](int k) {
                const bool ok = r_one && k < p.n_bins;
                const uint32_t off = ok ? (uint32_t)((((int64_t)k * F + f) * C + rc) * 8) : 0xfffffff0u;
                if constexpr (WIDE) {
                    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srs, (int)off, 0, 0));
                } else {
                    const float2 A = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(srs, (int)off, 0, 0));
                    const uint32_t off2 = (ok && r_two) ? off + 8u : 0xfffffff0u;
                    const float2 B = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(srs, (int)off2, 0, 0));
                    return make_float4(A.x, A.y, B.x, B.y);
                }
            };
            float4 gq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) gq[j] = fetch(rk + 256 * j);
            float2 ge;  // real parts of bin N / 2 (the threads with rk == 0; everybody else reads behind the end)
            {
                const bool ok = rk == 0 && r_one && N / 2 < p.n_bins;
                const uint32_t off = ok ? (uint32_t)((((int64_t)(N / 2) * F + f) * C + rc) * 8) : 0xfffffff0u;
                const uint32_t off2 = (ok && r_two) ? off + 8u : 0xfffffff0u;
                ge.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, (int)off, 0, 0));
                ge.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srs, (int)off2, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = rk + 256 * j;
                const float2 A = make_float2(gq[j].x, gq[j].y), B = make_float2(gq[j].z, gq[j].w);
                if (k == 0) {
                    im[fold_pos(0)] = make_float2(A.x, -B.x);  // conj(A.x + i B.x): numpy's irfft drops the imaginary parts
                } else {
                    im[fold_pos(k)] = make_float2(A.x - B.y, -A.y - B.x);      // conj(A + i B)
                    im[fold_pos(N - k)] = make_float2(A.x + B.y, A.y - B.x);  // conj(conj A + i conj B)
                }
            }
            if (rk == 0) im[fold_pos(N / 2)] = make_float2(ge.x, -ge.y);
        }
        __syncthreads();
        const int tl = tx & 255, bt_l = bin_thread(tl);
        float2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = buf[fold_pos(tl + 256 * n1)];
        Tw6 tw;  // six 8-byte loads per frame (L1): held across the loop they are 12 registers too many
        load_tw6(tw, p.tw, tl);
        fft4096_w(v, tw, buf, tw2, tl);  // (its first barrier stands behind these reads)
        // v[pos16(k3)] = N conj(x_c + i x_{c+1})[bt + 256 k3]
        const int64_t P0 = (int64_t)(f + q.off) * STEP;
        float2 fin[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // window[bt + 256 k3] * scale, read per frame (L1): 16 registers this kernel does not have
            const float wl = p.window[bt_l + 256 * j] * p.scale, wh = p.window[bt_l + 256 * (j + 8)] * p.scale;
            const float2 lo = v[pos16(j)], hi = v[pos16(j + 8)];
            fin[j] = make_float2(ca[j] + lo.x * wl, cbv[j] - lo.y * wl);  // the frame before + this frame's first half
            ca[j] = hi.x * wh;
            cbv[j] = -hi.y * wh;
        }
        __syncthreads();  // every wave has read its rows of the images
#pragma unroll
        for (int j = 0; j < 8; ++j) buf[fold_pos(bt_l + 256 * j)] = fin[j];
        __syncthreads();
        if (owned) {
            const bool fast = f + q.off >= 1 && f + q.off < q.n_total;  // both covering frame slots exist
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = tl + 256 * j;
                const float2 sv = buf[fold_pos(n)];
                if (one) emit(oa, P0 + n, sv.x, fast, n);
                if (two) emit(ob, P0 + n, sv.y, fast, n);
            }
        }
    }
    // behind the last frame: its second half, then nothing but the envelope's floor
    if (fb == p.n_frames) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) buf[fold_pos(bt + 256 * j)] = make_float2(ca[j], cbv[j]);
        __syncthreads();
        const int64_t P = (int64_t)(p.n_frames + q.off) * STEP;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = tid + 256 * j;
            const float2 sv = buf[fold_pos(n)];
            if (one) emit(oa, P + n, sv.x, false, n);
            if (two) emit(ob, P + n, sv.y, false, n);
        }
        for (int64_t n = P + STEP + tid; n < q.total_length; n += 256) {
            if (one) emit(oa, n, 0.f, false, 0);
            if (two) emit(ob, n, 0.f, false, 0);
        }
    }
}

}  // namespace stft4k
