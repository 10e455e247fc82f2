// Inverse STFT with frames of 8192 ... 262144 points on the 4096-point register transform: the mirror image of
// kernels_stft_long.hpp (reference: transforms.istft, transforms/transforms.py:444-586; overlap-add semantics:
// k_istft_ola, kernels_generic.hpp).  gfx950.  Round 4.
//
//   nfft = R x 4096, R = 2 ... 64.  Z = X_c + i X_{c+1} (two neighbouring channels ride one complex sequence; the upper
//   half of the spectrum from the Hermitian symmetry of both), class r = the bins R k' + r:
//       c_r[m] = FFT4096( conj Z[R k' + r] )[m]                                  (= conj of the inverse transform)
//       N conj(z[m + 4096 s]) = sum_r W_R^(r s) ( W_nfft^(r m) c_r[m] )          (forward R-point DFT over the classes)
//   k_icls   k_stft_cls backwards: ONE 1024-thread workgroup per CU = four teams, 8 channels per workgroup; the mirror of
//            bin R k' + r is bin R (4095 - k') + (R - r), so a thread that reads bin n of a channel pair (one 16-byte
//            load, 64- / 32-byte runs of the channel-fastest spectrogram) also places the mirror value: kind 0 = classes 0
//            and R / 2 of four channel pairs, kinds 1 ... R - 2 = classes (r, R - r) of two channel pairs on teams
//            (2 j, 2 j + 1).  One transform per team, c_r[m] out (class sequences in the workspace).
//   k_isdif  one thread per m < 4096 of a channel pair, walking a chunk of frames: R loads, the twiddles, the R-point
//            DFT in registers (welchl::dft_small), synthesis window and scale.  FUSED (W == nfft, step == nfft / 2, the
//            reference's default): frame f's second half and frame f + 1's first half meet in the SAME thread
//            (s >= R / 2 against s' = s - R / 2), so the overlap-add is a carry of R / 2 complex values, the envelope
//            division happens on the way out and no frame is ever stored.  Otherwise the windowed frames go to
//            frames[c][f][W] for k_istft_ola.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "kernels_stft4096.hpp"
#include "kernels_welch_long.hpp"

namespace istftl {

using welch4096::N;  // 4096
using stft4k::IMG;
using stft4k::NT;
using stft4k::TEAMS;
constexpr int LDS_BYTES = TEAMS * IMG * 8 + 256 * 8 + 6 * 256 * 8;  // four images, W256, the six per-thread twiddles

struct Args {
    const float2* stft;  // [n_bins][n_frames][n_ch]
    int n_bins, n_frames, n_ch, W, step;
    int n_chunks, n_groups;  // k_icls: chunks of (frame, kind) units, groups of 16 channels; k_isdif: chunks of frames
    int R, lgR;              // nfft = R * 4096
    int f0, nf;              // this launch's frames [f0, f0 + nf)
    const float* window;     // [W]
    const float2* twt;       // welch4096::host_tables()
    const float2* twl;       // welchl::host_tables(R)
    float scale;
    float2* cq;     // [channel pair][nf][R][4096]
    float* frames;  // unfused: [n_ch][n_frames][W]
    int off, n_total;  // fused: frame slots in front of the data, frame slots in all
    int64_t total_length, ld;
    float* out;
};

inline int classes_of(int nfft) {
    return (nfft >= 8192 && nfft <= 262144 && (nfft & (nfft - 1)) == 0) ? nfft / N : 0;
}
inline int frames_per_group(int n_ch, int nfft, int n_frames) {  // class sequences <= 512 MB
    const int64_t per_frame = (int64_t)((n_ch + 1) / 2) * nfft * 8;
    return (int)std::max<int64_t>(1, std::min<int64_t>(n_frames, ((int64_t)512 << 20) / per_frame));
}

template <bool WIDE>  // WIDE: an even channel count, one 16-byte load per bin and channel pair
__global__ __launch_bounds__(NT) void k_icls(Args p) {
    using namespace welch4096;
    extern __shared__ __align__(16) float2 lds[];
    float2* tw2 = lds + TEAMS * IMG;
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), tid = (int)threadIdx.x & 255;
    float2* buf = lds + team * IMG;
    const int x = (int)blockIdx.x & 7, s = (int)blockIdx.x >> 3, half = s & 1, u = (s >> 1) * 8 + x;
    const int g = u % p.n_groups, q = u / p.n_groups;
    const int cb = 16 * g + 8 * half;  // first of the workgroup's 8 channels
    if (q >= p.n_chunks || cb >= p.n_ch) return;
    const int R = p.R, kinds = R - 1;
    const int n_units = p.nf * kinds;
    const int u0 = (int)((int64_t)q * n_units / p.n_chunks), u1 = (int)((int64_t)(q + 1) * n_units / p.n_chunks);
    if (u0 >= u1) return;
    float2* tw6l = tw2 + 256;  // [6][256]: welch4096::load_tw6's values, from LDS (no global load in the loop but the gather)
    if (team == 0) tw2[tid] = p.twt[15 * 256 + tid];
    if (team == 1)
        for (int j = 0; j < 3; ++j) {
            tw6l[j * 256 + tid] = p.twt[j * 256 + tid];
            tw6l[(3 + j) * 256 + tid] = p.twt[(4 * (j + 1) - 1) * 256 + tid];
        }
    const int64_t F = p.n_frames, C = p.n_ch;
    const int nfft_half = R * (N / 2);

    // Passes of a frame: 0 -> classes 0, 1 -> R / 2 (four channel pairs each; they mirror into themselves);
    // 2 + 2 (r - 1) + h -> classes (r, R - r) of the channel pairs 2 h, 2 h + 1 (teams (2 j, 2 j + 1) = (r, R - r) of pair
    // 2 h + j).  A pass = gather (one 16-byte load per thread and bin, the value and its mirror placed in the images),
    // one transform per team, class sequence out.  The gather of pass i + 1 is REQUESTED inside the transform of pass i
    // (one workgroup per CU, all sixteen waves in the same phase: nobody else would hide that latency).
    struct Pass {
        int fl, rr, h;
        bool cross;
    };
    const int per_frame = R;  // (kind 0 counts twice)
    auto decode = [&](int pi) {
        Pass a;
        a.fl = pi / per_frame;
        const int j = pi - a.fl * per_frame;
        a.cross = j >= 2;
        a.rr = j == 0 ? 0 : (j == 1 ? R / 2 : 1 + ((j - 2) >> 1));
        a.h = a.cross ? (j - 2) & 1 : 0;
        return a;
    };
    // the chunk's passes: whole units (u0 .. u1) -> passes
    auto first_pass = [&](int un) {
        const int fl = un / kinds, kind = un - fl * kinds;
        return fl * per_frame + (kind == 0 ? 0 : kind + 1);
    };
    const int p0 = first_pass(u0), p1 = u1 == n_units ? p.nf * per_frame : first_pass(u1);

    // which channel pair / class this thread gathers in pass a
    auto gather_of = [&](const Pass& a, int tx, int& rc, int& res) {
        if (!a.cross) {
            rc = cb + 2 * (tx & 3);
            res = a.rr;
        } else {
            rc = cb + 2 * (2 * a.h + (tx & 1));
            res = ((tx >> 1) & 1) ? R - a.rr : a.rr;
        }
    };
    // Raw-buffer loads (the spectrogram is smaller than 4 GB here, the host checks): what is not there -- bins past the
    // stored ones, channels past the last -- is an offset behind the end and reads as zero, without a branch.  (A plain
    // load inside a conditional block makes the compiler wait for EVERY outstanding load at the next use of any of them,
    // and hipcc turns `ok ? load : 0` back into such a block.)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float2*>(p.stft), 0, (int)(uint32_t)((int64_t)p.n_bins * F * C * 8), 0x00020000);
    typedef int desc_t __attribute__((ext_vector_type(4)));
    desc_t cq_desc;  // raw-buffer descriptor of this launch's class sequences (smaller than 4 GB as well)
    {
        const uint64_t ba = (uint64_t)(uintptr_t)p.cq;
        cq_desc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)ba);
        cq_desc.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((ba >> 32) & 0xffffu));
        cq_desc.z = __builtin_amdgcn_readfirstlane((int)(uint32_t)((int64_t)((p.n_ch + 1) / 2) * p.nf * R * N * 8));
        cq_desc.w = 0x00020000;
    }
    auto fetch = [&](int n, int f, int rc) {
        const bool ok = rc < p.n_ch && n < p.n_bins;
        const uint32_t off = ok ? (uint32_t)((((int64_t)n * F + f) * C + rc) * 8) : 0xfffffff0u;
        if constexpr (WIDE) {
            return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
        } else {
            const float2 A = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0));
            const uint32_t off2 = (ok && rc + 1 < p.n_ch) ? off + 8u : 0xfffffff0u;
            const float2 B = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off2, 0, 0));
            return make_float4(A.x, A.y, B.x, B.y);
        }  // (A.x, A.y, B.x, B.y) = X_c[n], X_{c+1}[n]
    };
    float4 gq[8];
    float2 ge = make_float2(0.f, 0.f);  // real parts of bin nfft / 2 (pass 0, the threads with rk == 0)
    auto request = [&](const Pass& a, int j0, int j1) {
        int tx = (int)threadIdx.x;
        asm volatile("" : "+v"(tx));
        int rc, res;
        gather_of(a, tx, rc, res);
        const int rk = tx >> 2, f = p.f0 + a.fl;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j >= j0 && j < j1) gq[j] = fetch(R * (rk + 256 * j) + res, f, rc);
        if (j1 == 8) {  // (every thread issues the load; all but four get zeros from behind the end)
            const bool edge = !a.cross && a.rr == 0 && rk == 0;
            // (two 4-byte loads of exactly the two values: a wider load's unused destination registers get reused
            // by the transform, and writing them waits for every outstanding load)
            const bool ok = edge && rc < p.n_ch && nfft_half < p.n_bins;
            const uint32_t off = ok ? (uint32_t)((((int64_t)nfft_half * F + f) * C + rc) * 8) : 0xfffffff0u;
            const uint32_t off2 = (ok && (WIDE || rc + 1 < p.n_ch)) ? off + 8u : 0xfffffff0u;
            ge.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
            ge.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off2, 0, 0));
        }
    };
    if (p0 < p1) request(decode(p0), 0, 8);
    // Order of a pass (vmcnt counts loads and stores together, and with both kinds outstanding a wait for the loads is a
    // wait for everything): gather values of pass i + 1 requested inside the transform of pass i -> placed in the images
    // right behind it (the stores of pass i - 1 are old by then) -> the class sequence of pass i sent to memory -> an
    // LDS-only barrier (a __syncthreads() would wait for those stores) -> transform of pass i + 1.
    auto place = [&](const Pass& a) {
        int tx = (int)threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int rk = tx >> 2;
        if (!a.cross) {
            float2* im = lds + (tx & 3) * IMG;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = rk + 256 * j;  // k' < 2048
                const float4 q4 = gq[j];
                if (a.rr == 0 && kk == 0) {
                    im[fold_pos(0)] = make_float2(q4.x, -q4.z);  // bin 0: numpy's irfft drops the imaginary parts
                } else {
                    const int km = a.rr ? N - 1 - kk : N - kk;
                    im[fold_pos(kk)] = make_float2(q4.x - q4.w, -q4.y - q4.z);  // conj(A + i B)
                    im[fold_pos(km)] = make_float2(q4.x + q4.w, q4.y - q4.z);   // conj(conj A + i conj B)
                }
            }
            if (a.rr == 0 && rk == 0) im[fold_pos(N / 2)] = make_float2(ge.x, -ge.y);  // bin nfft / 2 = class 0, k' = 2048
        } else {
            const int rp2 = tx & 1, hi = (tx >> 1) & 1;
            float2* own = lds + (2 * rp2 + hi) * IMG;
            float2* oth = lds + (2 * rp2 + 1 - hi) * IMG;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = rk + 256 * j;
                const float4 q4 = gq[j];
                own[fold_pos(kk)] = make_float2(q4.x - q4.w, -q4.y - q4.z);
                oth[fold_pos(N - 1 - kk)] = make_float2(q4.x + q4.w, q4.y - q4.z);
            }
        }
        // every requested value is consumed HERE on every path (the edge bin only matters to pass 0): a load the compiler
        // still counts as outstanding at the head of the loop costs a full wait there -- which the hidden stores join
        asm volatile("" ::"v"(ge.x), "v"(ge.y));
    };
    Stamp ts;
    __syncthreads();  // tables
    place(decode(p0));
    for (int pi = p0; pi < p1; ++pi) {
        const Pass a = decode(pi);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the images are complete
        int tx = (int)threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int tl = tx & 255, bt_l = bin_thread(tl);
        float2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = buf[fold_pos(tl + 256 * n1)];
        Tw6 tw;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            tw.a[j] = tw6l[j * 256 + tl];
            tw.b[j] = tw6l[(3 + j) * 256 + tl];
        }
        const bool more = pi + 1 < p1;
        const Pass nx = decode(more ? pi + 1 : pi);
        fft4096_wi(
            v, tw, buf, tw2, tl, [&](int g) { request(nx, 2 * g, 2 * g + 2); },  // (behind the last pass: the same addresses again)
            [&](int) {}, ts, 0);
        // (fft4096_wi ends with a barrier behind its last image read)
        place(nx);  // (unconditionally -- behind the last pass its own values once more: a request left unconsumed on one
                    // path makes the compiler wait for it, and with it for the hidden stores, at the head of the loop)
        const int pair = a.cross ? 2 * a.h + (team >> 1) : team;
        const int res = a.cross ? ((team & 1) ? R - a.rr : a.rr) : a.rr;
        const int c0 = cb + 2 * pair;
        // The class sequence leaves through stores the compiler does not see (inline assembly): it keeps ONE count of
        // outstanding loads and stores, cannot tell them apart at a wait, and would drain these stores in front of the
        // next transform -- here they drain beside it.  (Hidden outstanding operations only make a counted wait
        // stricter, never looser; 8-byte store data is read at issue, so v may be overwritten right away.)
        // Order in memory: value m = bin_thread(t) + 256 k3 of thread t at index t + 256 k3 -- whole 512-byte runs per
        // wave and store; k_isdif undoes the permutation of each 256-block (bin_thread is its own inverse).
        if (c0 < p.n_ch) {  // (uniform per team)
            const uint32_t base = (uint32_t)((((((int64_t)(c0 >> 1)) * p.nf + a.fl) * R + res) * N + tl) * 8);
#pragma unroll
            for (int k3 = 0; k3 < 16; k3 += 2) {
                const uint32_t off = base + 2048u * (uint32_t)k3;
                const uint64_t d0 = __builtin_bit_cast(uint64_t, v[pos16(k3)]), d1 = __builtin_bit_cast(uint64_t, v[pos16(k3 + 1)]);
                asm volatile("buffer_store_dwordx2 %0, %2, %3, 0 offen\n\tbuffer_store_dwordx2 %1, %2, %3, 0 offen offset:2048"
                             :
                             : "v"(d0), "v"(d1), "v"(off), "s"(cq_desc)
                             : "memory");
            }
        }
    }
}

// R-point DFT over the classes of one m: t[s] = sum_r W_R^(r s) W_nfft^(r m) c_r[m] = nfft conj(z[m + 4096 s]).
// src: the 256-block of the class sequences this workgroup owns (class r at src + r * 4096), in k_icls's order: the
// value of m = blk * 256 + bin_thread(j) sits at index j.  Loaded as stored (coalesced), turned through xch[R][272].
__device__ __forceinline__ int xch_pos(int k) { return k + (k >> 4); }
constexpr int XCH_ROW = 272;
inline size_t xch_bytes(int R) { return (size_t)R * XCH_ROW * sizeof(float2); }
template <int R>
__device__ __forceinline__ void classes_to_samples(const Args& p, const float2* __restrict__ src, int m, float2* xch,
                                                   float2 (&t)[R]) {
    const int tid = threadIdx.x, perm = xch_pos(welch4096::bin_thread(tid));
    __syncthreads();  // the previous frame's reads
#pragma unroll
    for (int r = 0; r < R; ++r) xch[r * XCH_ROW + perm] = src[(int64_t)r * N + tid];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float2 c = xch[r * XCH_ROW + xch_pos(tid)];
        t[r] = r ? welch4096::cmul(c, p.twl[(size_t)r * N + m]) : c;
    }
    if constexpr (R == 2) {
        const float2 a = t[0], b = t[1];
        t[0] = make_float2(a.x + b.x, a.y + b.y);
        t[1] = make_float2(a.x - b.x, a.y - b.y);
    } else {
        welchl::dft_small<R>(t, p.twl + (size_t)R * N);
    }
}

// ---- unfused: windowed frames for k_istft_ola.  grid = (16, nf, channel pairs) -----------------------------------
template <int R>
__global__ __launch_bounds__(256) void k_isdif_frames(Args p) {
    const int m = (int)blockIdx.x * 256 + (int)threadIdx.x, fl = blockIdx.y, pc = blockIdx.z;
    const int c0 = 2 * pc;
    extern __shared__ __align__(16) float2 xch[];
    float2 t[R];
    classes_to_samples<R>(p, p.cq + (((int64_t)pc * p.nf + fl) * R) * N + (int)blockIdx.x * 256, m, xch, t);
    float* fa = p.frames + ((int64_t)c0 * p.n_frames + p.f0 + fl) * p.W;
    float* fb = fa + (int64_t)p.n_frames * p.W;
    const bool two = c0 + 1 < p.n_ch;
#pragma unroll
    for (int s = 0; s < R; ++s) {
        const int n = m + N * s;
        if (n < p.W) {
            const float w = p.window[n] * p.scale;
            fa[n] = t[s].x * w;
            if (two) fb[n] = -t[s].y * w;
        }
    }
}

// ---- fused overlap-add (W == nfft, step == nfft / 2).  grid = (16, n_chunks, channel pairs); all frames in one launch
template <int R>
__global__ __launch_bounds__(256) void k_isdif_ola(Args p) {
    constexpr int H = R / 2;
    extern __shared__ __align__(16) float2 xch[];
    const int m = (int)blockIdx.x * 256 + (int)threadIdx.x, qc = blockIdx.y, pc = blockIdx.z;
    const int c0 = 2 * pc;
    const bool two = c0 + 1 < p.n_ch;
    const int per = (p.n_frames + p.n_chunks - 1) / p.n_chunks;
    const int fa = qc * per, fb = min(fa + per, p.n_frames);
    if (fa >= fb) return;
    float* oa = p.out + (int64_t)c0 * p.ld;
    float* ob = p.out + (int64_t)(two ? c0 + 1 : c0) * p.ld;
    const int64_t step = p.step;
    // A thread's positions are pos = fs * step + mm with mm = m + 4096 s < step: no division anywhere.  Frame SLOT fs
    // covers pos with window[mm], slot fs - 1 with window[mm + step] (k_istft_ola's rule for the envelope); where both
    // exist -- everywhere but at the two ends -- the envelope depends on (m, s) only.
    float wl[H], wh[H], inv_full[H];
#pragma unroll
    for (int s = 0; s < H; ++s) {
        const float a = p.window[m + N * s], b = p.window[m + N * (s + H)];
        const double e = (double)a * (double)a + (double)b * (double)b;
        inv_full[s] = (float)(1.0 / (e < 1e-4 ? 1e-4 : e));
        wl[s] = a * p.scale;
        wh[s] = b * p.scale;
    }
    auto emit = [&](int64_t fs, int s, float sa, float sb) {
        const int64_t pos = fs * step + m + N * s;
        if (pos >= p.total_length) return;
        const bool lo = fs < p.n_total, hi = fs >= 1 && fs - 1 < p.n_total;
        float inv = inv_full[s];
        if (!(lo && hi)) {
            const double a = lo ? (double)p.window[m + N * s] : 0.0, b = hi ? (double)p.window[m + N * (s + H)] : 0.0;
            const double env = a * a + b * b;
            inv = (float)(1.0 / (env < 1e-4 ? 1e-4 : env));
        }
        oa[pos] = sa * inv;
        if (two) ob[pos] = sb * inv;
    };
    // frame slots in front of the data (the reference's empty frame when the signal was not padded): zeros
    if (fa == 0)
        for (int64_t fs = 0; fs < p.off; ++fs)
#pragma unroll
            for (int s = 0; s < H; ++s) emit(fs, s, 0.f, 0.f);
    float2 carry[H];
#pragma unroll
    for (int j = 0; j < H; ++j) carry[j] = make_float2(0.f, 0.f);
    for (int f = fa > 0 ? fa - 1 : 0; f < fb; ++f) {
        float2 t[R];
        classes_to_samples<R>(p, p.cq + (((int64_t)pc * p.n_frames + f) * R) * N + (int)blockIdx.x * 256, m, xch, t);
        const bool owned = f >= fa;
#pragma unroll
        for (int s = 0; s < H; ++s) {
            const float sa = carry[s].x + t[s].x * wl[s], sb = carry[s].y - t[s].y * wl[s];
            carry[s] = make_float2(t[s + H].x * wh[s], -t[s + H].y * wh[s]);
            if (owned) emit(f + p.off, s, sa, sb);
        }
    }
    if (fb == p.n_frames) {  // the last frame's second half, then nothing but zeros
        const int64_t fl = (int64_t)p.n_frames + p.off;
#pragma unroll
        for (int s = 0; s < H; ++s) emit(fl, s, carry[s].x, carry[s].y);
        for (int64_t fs = fl + 1; fs * step < p.total_length; ++fs)
#pragma unroll
            for (int s = 0; s < H; ++s) emit(fs, s, 0.f, 0.f);
    }
}

}  // namespace istftl
