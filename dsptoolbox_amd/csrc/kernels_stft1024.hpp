// STFT with a 256-, 512- or 1024-point transform (1024 samples is the reference's default window)
// (standard/_spectral_methods.py:126-148, 260-268 with window_length_samples = 1024).
//
// The generic kernel (kernels_generic.hpp: k_stft<1024>) spends ~1500 VALU instructions per frame
// pair -- padded-index arithmetic of the LDS passes, twiddles fetched from global memory, a
// branchy in-place separation and read-out -- and is VALU-bound (rocprof: 48 M VALU instructions
// per launch on the 64-microphone CSM shape = 55 us of issue alone).  Here one wave owns a frame
// pair and keeps the transform in registers:
//
//   1024 = 16 x 16 x 4,  n = 64 n1 + 4 n2 + n3,  k = k1 + 16 k2 + 256 k3
//   pass 1  lane t = 4 n2 + n3 : DFT16 over n1, times W1024^(t k1)        (table in LDS, shared)
//   pass 2  lane u = 4 k1 + n3 : DFT16 over n2, times W64^(n3 k2)
//   pass 3  lane v            : four radix-4 butterflies over n3 for the pairs (k1,k2) = v + 64 j
//   -> lane v holds Z[v + 64 m], m = 0..15
//
// Two wave-local LDS exchanges (a wave's LDS operations execute in order: no s_barrier), then the
// two packed real frames are separated against Z[N-k] (upper half through LDS) and written as the
// row image  row 2k = frame f0 bin k, row 2k+1 = frame f0+1 bin k,  which the whole workgroup
// streams out channel-fastest with one LDS read and one store per element and no index math:
// out[(k F + f) C + c].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <vector>

#include "kernels_generic.hpp"
#include "kernels_welch4096.hpp"

namespace stft1k {

namespace w4 = welch4096;
using dsk::FrameSrc;
using dsk::RawPair;
using dsk::StftArgs;
using w4::cmul;
using w4::pos16;

// Geometry of the wave-level transform of N = 256, 512 or 1024 points: N = 16 x 16 x R3 on
// L = N/16 lanes (1, 2 or 4 transforms per wave), 16 complex values per lane.
template <int NN>
struct Geo {
    static_assert(NN == 256 || NN == 512 || NN == 1024 || NN == 2048,
                  "team-level transform: 256, 512, 1024 points (one wave or less) or 2048 (two waves)");
    static constexpr int N = NN, L = NN / 16, R3 = NN / 256;
    static constexpr int S1 = L + R3;   // exchange-1 row stride (the R3-lane groups of pass 2 spread over the banks)
    static constexpr int REGION = ((16 * S1 > NN + 2 ? 16 * S1 : NN + 2) + 31) / 32 * 32;  // complex per image
    static constexpr int TW1 = 15 * L, TW_LEN = TW1 + 16 * R3;
};
constexpr int N = 1024, NB = N / 2 + 1;
constexpr int S1 = Geo<1024>::S1, REGION = Geo<1024>::REGION, TW1 = Geo<1024>::TW1, TW_LEN = Geo<1024>::TW_LEN;

// channel stride: REGION + 32/ct complex, so the ct channels x L/ct rows a wave reads in the
// channel-fastest read-out fall on distinct banks; even (float4-aligned images): ct <= 16
template <int NN>
__host__ __device__ constexpr int ch_stride(int ct) { return Geo<NN>::REGION + (ct > 1 ? 32 / ct : 0); }
// images + twiddle tables (+ the window, NN floats, zero beyond the caller's window length, for the
// lengths whose images leave room: at 2048 points eight 17 KB images fill the CU)
template <int NN>
__host__ __device__ constexpr bool win_in_lds() { return NN <= 1024; }
template <int NN>
inline size_t lds_bytes(int ct) {
    return ((size_t)ct * ch_stride<NN>(ct) + Geo<NN>::TW_LEN) * sizeof(float2) +
           (win_in_lds<NN>() ? (size_t)NN * sizeof(float) : 0);
}

// [15][L] W_N^(t k1) (k1 = 1..15), then [16][R3] W_(16 R3)^(n3 k2); fp64-computed
template <int NN>
inline void host_tables(std::vector<float2>& t) {
    using G = Geo<NN>;
    t.resize(G::TW_LEN);
    for (int k1 = 1; k1 < 16; ++k1)
        for (int tt = 0; tt < G::L; ++tt) {
            double a = -2.0 * M_PI * (double)(tt * k1) / (double)NN;
            t[(k1 - 1) * G::L + tt] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (int k2 = 0; k2 < 16; ++k2)
        for (int n3 = 0; n3 < G::R3; ++n3) {
            double a = -2.0 * M_PI * (double)(n3 * k2) / (double)(16 * G::R3);
            t[G::TW1 + k2 * G::R3 + n3] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
}
inline void host_tables(std::vector<float2>& t) { host_tables<1024>(t); }

// order the LDS traffic of ONE wave (hardware executes it in program order; this keeps the
// compiler from moving a read above the write of another lane it cannot see)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// synchronise the lanes of one transform: up to 64 lanes are (part of) one wave; a 128-lane team
// (2048 points) spans two waves and takes the workgroup barrier -- every team of the workgroup then
// has to walk through the transform together
template <int NN>
__device__ __forceinline__ void team_sync() {
    if constexpr (NN / 16 <= 64)
        wave_sync();
    else
        __syncthreads();
}

// 8-point DFT in registers, natural order in and out (forward)
__device__ __forceinline__ void dft8(float2 (&x)[8]) {
    constexpr float R2 = 0.70710678118654752440f;
    w4::r4(x[0], x[2], x[4], x[6]);  // even samples -> E[0..3] in x[0], x[2], x[4], x[6]
    w4::r4(x[1], x[3], x[5], x[7]);  // odd samples  -> O[0..3] in x[1], x[3], x[5], x[7]
    // O[k] W8^k
    const float2 o0 = x[1];
    const float2 o1 = make_float2((x[3].x + x[3].y) * R2, (x[3].y - x[3].x) * R2);    // (1 - i)/sqrt2
    const float2 o2 = make_float2(x[5].y, -x[5].x);                                    // -i
    const float2 o3 = make_float2((x[7].y - x[7].x) * R2, -(x[7].x + x[7].y) * R2);   // (-1 - i)/sqrt2
    const float2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    x[0] = make_float2(e0.x + o0.x, e0.y + o0.y);
    x[4] = make_float2(e0.x - o0.x, e0.y - o0.y);
    x[1] = make_float2(e1.x + o1.x, e1.y + o1.y);
    x[5] = make_float2(e1.x - o1.x, e1.y - o1.y);
    x[2] = make_float2(e2.x + o2.x, e2.y + o2.y);
    x[6] = make_float2(e2.x - o2.x, e2.y - o2.y);
    x[3] = make_float2(e3.x + o3.x, e3.y + o3.y);
    x[7] = make_float2(e3.x - o3.x, e3.y - o3.y);
}

struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// The transform of one team of L lanes: v[n1] = z[t + L n1] (consumed) -> zo[m] = Z[t + L m].
// `buf`: the team's LDS region (>= Geo::REGION complex, 16-byte aligned), tw1/tw2 the tables of
// host_tables<N>() in LDS.  Every lane of the wave (2048 points: of the workgroup) must call it.  Ends with
// buf free for reuse.  `behind_ex2` runs after the second exchange image has been written (v is
// dead there): global loads issued from it overlap the LDS round trip and the last pass.
template <int NN, typename Hook = NoHook>
__device__ __forceinline__ void fft_wave(float2 (&v)[16], float2 (&z)[16], float2* buf,
                                         const float2* tw1, const float2* tw2, int t,
                                         Hook behind_ex2 = Hook()) {
    using G = Geo<NN>;
    constexpr int L = G::L, R3 = G::R3, S1 = G::S1;
    const int k1u = t / R3, n3 = t % R3;
    // ---- pass 1
    // (spreading the 16 image stores between the butterflies, as the 4096-point kernel does, was
    // A/B-tested here in one process on one GPU: no difference for these wave-local exchanges)
    w4::dft16(v);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[pos16(k1)] = cmul(v[pos16(k1)], tw1[(k1 - 1) * L + t]);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) buf[k1 * S1 + t] = v[pos16(k1)];
    team_sync<NN>();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = buf[k1u * S1 + R3 * n2 + n3];
    team_sync<NN>();
    // ---- pass 2
    w4::dft16(v);
    if constexpr (R3 == 1) {
        // 256 points: lane k1 already holds Z[k1 + 16 k2]
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) z[k2] = v[pos16(k2)];
        behind_ex2();
    } else {
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) v[pos16(k2)] = cmul(v[pos16(k2)], tw2[k2 * R3 + n3]);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) buf[L * k2 + t] = v[pos16(k2)];  // [(k1 + 16 k2)][n3]
        behind_ex2();
        team_sync<NN>();
        // ---- pass 3: pair (k1,k2) = t + L j, radix R3 over n3 -> Z[t + L (j + (16/R3) k3)]
        if constexpr (R3 == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4* q = reinterpret_cast<const float4*>(buf + 4 * (t + L * j));
                const float4 lo = q[0], hi = q[1];
                float2 x0 = make_float2(lo.x, lo.y), x1 = make_float2(lo.z, lo.w);
                float2 x2 = make_float2(hi.x, hi.y), x3 = make_float2(hi.z, hi.w);
                w4::r4(x0, x1, x2, x3);
                z[j] = x0;
                z[j + 4] = x1;
                z[j + 8] = x2;
                z[j + 12] = x3;
            }
        } else if constexpr (R3 == 8) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4* q = reinterpret_cast<const float4*>(buf + 8 * (t + L * j));
                float2 x[8];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const float4 r = q[h];
                    x[2 * h] = make_float2(r.x, r.y);
                    x[2 * h + 1] = make_float2(r.z, r.w);
                }
                dft8(x);
#pragma unroll
                for (int k3 = 0; k3 < 8; ++k3) z[j + 2 * k3] = x[k3];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 q = *reinterpret_cast<const float4*>(buf + 2 * (t + L * j));
                z[j] = make_float2(q.x + q.z, q.y + q.w);
                z[j + 8] = make_float2(q.x - q.z, q.y - q.w);
            }
        }
        team_sync<NN>();
    }
}
template <typename Hook = NoHook>
__device__ __forceinline__ void fft1024(float2 (&v)[16], float2 (&z)[16], float2* buf,
                                        const float2* tw1, const float2* tw2, int t,
                                        Hook behind_ex2 = Hook()) {
    fft_wave<1024, Hook>(v, z, buf, tw1, tw2, t, behind_ex2);
}

// Raw samples of a frame pair through ONE raw-buffer descriptor over the whole planar signal
// (32-bit byte offsets in a single register per frame).  The generic loader keeps a 64-bit
// address per group of loads; at the 128 registers two 512-thread workgroups per CU allow those
// addresses were spilled and every reload carried an `s_waitcnt vmcnt(0)` that drained the previous
// pair's output stores.  Interior pairs load unconditionally; for a pair that touches the front
// padding, the end of the signal or a missing second frame each offset is replaced by an
// out-of-range one where the sample does not exist, and the hardware range check returns 0.
struct Raw16 {
    float a[16], b[16];
};
__device__ __forceinline__ float ld_f32(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
template <int L>
__device__ __forceinline__ void load_pair_buf(Raw16& r, __amdgpu_buffer_rsrc_t rs, uint32_t chan_samples, bool live,
                                              int start_a, int start_b, bool have_b, int n_samples, int t) {
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    constexpr int NN = 16 * L;
    const bool interior = live && start_a >= 0 && (have_b ? start_b : start_a) + NN <= n_samples;
    if (interior) {
        const uint32_t oa = (chan_samples + (uint32_t)(start_a + t)) * 4u;
        const uint32_t ob = have_b ? (chan_samples + (uint32_t)(start_b + t)) * 4u : OOB - 64u * L;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            r.a[n1] = ld_f32(rs, oa + 4u * L * n1);
            r.b[n1] = ld_f32(rs, have_b ? ob + 4u * L * n1 : OOB);
        }
    } else {
        const uint32_t ns = live ? (uint32_t)n_samples : 0u;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int sa = start_a + t + L * n1, sb = start_b + t + L * n1;
            r.a[n1] = ld_f32(rs, (uint32_t)sa < ns ? (chan_samples + (uint32_t)sa) * 4u : OOB);
            r.b[n1] = ld_f32(rs, (have_b && (uint32_t)sb < ns) ? (chan_samples + (uint32_t)sb) * 4u : OOB);
        }
    }
}
// the host takes this kernel only when the byte offsets fit (stft_wave_fits)
inline bool stft_wave_fits(int64_t n_samples, int n_ch, int64_t ld, int64_t pad_front, int nfft) {
    return ((int64_t)(n_ch - 1) * ld + n_samples) * 4 < ((int64_t)1 << 32) - ((int64_t)1 << 20) &&
           n_samples + pad_front + 4 * (int64_t)nfft < ((int64_t)1 << 30);
}

// grid = (ceil(ceil(n_frames/2)/fpw), ceil(n_ch/ct)); block = ct teams of L lanes; p.tw =
// host_tables<NN>(); p.W <= NN (shorter windows are zero-padded); detrend requires p.W == NN, where
// removing the frame mean only clears bin 0 (a constant has no other bin).
template <int NN, bool POWER>
__global__ __launch_bounds__(1024) void k_stft_wave(StftArgs p) {
    using G = Geo<NN>;
    constexpr int L = G::L;
    extern __shared__ __align__(16) float2 lds[];
    // a 64-lane team is a wave: its index (and every LDS base derived from it) is wave-uniform
    const int team = L == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)threadIdx.x / L;
    const int t = threadIdx.x % L;
    const int CHS = ch_stride<NN>(p.ct);
    float2* buf = lds + team * CHS;
    float2* tw1 = lds + p.ct * CHS;
    const float2* tw2 = tw1 + G::TW1;
    for (int i = threadIdx.x; i < G::TW_LEN; i += blockDim.x) tw1[i] = p.tw[i];
    // the window lives in LDS (16 registers less per lane: with them the kernel spilled 19 dwords
    // at the 128 registers that two 512-thread workgroups per CU allow)
    float* winl = reinterpret_cast<float*>(tw1 + G::TW_LEN);
    float winr[win_in_lds<NN>() ? 1 : 16];
    if constexpr (win_in_lds<NN>()) {
        for (int i = threadIdx.x; i < NN; i += blockDim.x) winl[i] = i < p.W ? p.window[i] : 0.f;
    } else {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1)
            winr[n1] = p.window[min(t + L * n1, p.W - 1)] * (t + L * n1 < p.W ? 1.f : 0.f);
    }
    const int c0 = blockIdx.y * p.ct;
    const int ctv = min(p.ct, p.n_ch - c0);
    const int c = c0 + team;
    const bool live = c < p.n_ch;  // idle teams transform zeros
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)(uint32_t)(((int64_t)(p.n_ch - 1) * p.ld + p.n_samples) * 4), 0x00020000);
    const uint32_t chan = live ? (uint32_t)((int64_t)c * p.ld) : 0u;
    const int64_t F = p.n_frames, Cn = p.n_ch;
    const int lct = __ffs(p.ct) - 1;
    const int cl = threadIdx.x & (p.ct - 1), r0 = threadIdx.x >> lct;  // read-out: channel, first row (< L)
    const int n_fp = (p.n_frames + 1) >> 1;
    const int fp0 = blockIdx.x * p.fpw, fp1 = min(fp0 + p.fpw, n_fp);
    Raw16 raw;
    auto load = [&](int fp) {
        const int f0 = 2 * fp;
        const int sa = (int)((int64_t)f0 * p.hop - p.pad_front);
        load_pair_buf<L>(raw, rs, chan, live, sa, sa + p.hop, f0 + 1 < p.n_frames, (int)p.n_samples, t);
    };
    if (fp0 < fp1) load(fp0);
    // Drain the loads above before the loop: otherwise the wait for them is merged into the loop body
    // (vmcnt is a FIFO count) and stalls every iteration on the prefetch it has just issued.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // 0.5 from the separation folded into the scale; lane 0 owns the edge bins 0 and N/2
    const float sc = p.scale, sce = p.scale * p.edge_scale;
    const float pe = p.scale, pee = p.scale * p.edge_scale * p.edge_scale;  // power mode
    const float s0 = (t == 0) ? sce : sc, e0 = (t == 0) ? pee : pe;
    const float dc = (p.detrend && t == 0) ? 0.f : 1.f;

    for (int fp = fp0; fp < fp1; ++fp) {
        const int f0 = 2 * fp;
        const bool v1 = f0 + 1 < p.n_frames;
        float2 v[16];
        __syncthreads();  // tables written / the previous pair has been streamed out
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const float w = win_in_lds<NN>() ? winl[t + L * n1] : winr[win_in_lds<NN>() ? 0 : n1];
            v[n1] = make_float2(raw.a[n1] * w, raw.b[n1] * w);
        }
        if (fp + 1 < fp1) load(fp + 1);
        float2 z[16];
        fft_wave<NN>(v, z, buf, tw1, tw2, t);
        // ---- separation: bins k = t + L j (j < 8) against Z[N - k], which sits in the upper half
#pragma unroll
        for (int m = 8; m < 16; ++m) buf[t + L * m] = z[m];
        team_sync<NN>();
        float2 qc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) qc[j] = buf[(NN - (t + L * j)) & (NN - 1)];
        if (t == 0) qc[0] = z[0];  // bin 0 pairs with itself
        team_sync<NN>();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float2 P = z[j], Q = qc[j];
            float2 A = make_float2(0.5f * (P.x + Q.x), 0.5f * (P.y - Q.y));
            float2 B = make_float2(0.5f * (P.y + Q.y), -0.5f * (P.x - Q.x));
            if (POWER) {
                const float e = (j == 0 ? e0 * dc : pe);
                A = make_float2((A.x * A.x + A.y * A.y) * e, 0.f);
                B = make_float2((B.x * B.x + B.y * B.y) * e, 0.f);
            } else {
                const float s = (j == 0 ? s0 * dc : sc);
                A = make_float2(A.x * s, A.y * s);
                B = make_float2(B.x * s, B.y * s);
            }
            *reinterpret_cast<float4*>(buf + 2 * (t + L * j)) = make_float4(A.x, A.y, B.x, B.y);
        }
        if (t == 0) {  // bin N/2 pairs with itself
            const float2 P = z[8];
            float4 r;
            if (POWER)
                r = make_float4(P.x * P.x * pee, 0.f, P.y * P.y * pee, 0.f);
            else
                r = make_float4(P.x * sce, 0.f, P.y * sce, 0.f);
            *reinterpret_cast<float4*>(buf + NN) = r;
        }
        __syncthreads();
        // ---- read-out: rows r0 + L i of channel cl; row r -> out[((r>>1) F + f0 + (r&1)) C + c]
        // (p.decim = D > 1: frames of NN / D samples -- their transform is every D-th bin of the NN-point transform
        // of the zero-padded frame; only those rows are stored, at bin (r >> 1) / D.  D divides L / 2.)
        const int D = p.decim;
        if (cl < ctv && (v1 || !(r0 & 1)) && ((r0 >> 1) & (D - 1)) == 0) {
            const float2* s = lds + cl * CHS + r0;
            float2* o = p.out + ((int64_t)((r0 >> 1) / D) * F + f0 + (r0 & 1)) * Cn + c0 + cl;
            const int64_t ostep = (L / 2 / D) * F * Cn;
            float2 g[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] = s[L * i];
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i * ostep] = g[i];
            if (r0 < 2) o[16 * ostep] = s[NN];
        }
    }
}

// ---- inverse STFT on the same wave-level transform ------------------------------------------------------
// (reference: transforms.istft, transforms/transforms.py:444-586; the generic kernels and the semantics of the
// overlap-add are in kernels_generic.hpp: k_istft_fused.)  Full-length frames at 50 % overlap, NN = 256 ... 2048.
// The inverse transform is the forward one between two conjugations: ifft(Z) = conj(fft(conj Z)) (unnormalised; the
// caller's scale carries 1 / N), so fft_wave serves unchanged.  A workgroup = ct teams (channels); per frame pair
//   * all threads read the bins of the ct channels together (runs of 8 ct bytes of the channel-fastest spectrogram)
//     and lay conj(Z) = conj(A + i B) and its mirror half into the ct images in natural order;
//   * every team takes its 16 values per lane out of its image (z[t + L n1]), transforms, and holds the two frames'
//     samples n = t + L m: frame f0 = Re / frame f0 + 1 = -Im; m < 8 is the first half of a frame, m >= 8 the second:
//     the overlap-add is lane-local -- 8 sums travel in registers to the next pair -- and finished samples leave
//     as 4 L-byte runs per channel; 1 / envelope from an LDS table wherever two frame slots cover a sample.
// A workgroup owns the output from the start of its first frame to the start of the next workgroup's and transforms
// the pair in front of its range once more for the carry (k_istft_fused).
template <int NN>
inline size_t istft_lds_bytes(int ct) {
    return ((size_t)ct * ch_stride<NN>(ct) + Geo<NN>::TW_LEN) * sizeof(float2) + (size_t)(NN / 2) * sizeof(float);
}
template <int NN>
__global__ __launch_bounds__(NN == 2048 ? 512 : (NN >= 1024 ? 1024 : NN)) void k_istft_wave(dsk::IstftFusedArgs q) {
    using G = Geo<NN>;
    constexpr int L = G::L, STEP = NN / 2;
    const dsk::IstftArgs& p = q.a;
    extern __shared__ __align__(16) float2 lds[];
    const int team = L == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)threadIdx.x / L;
    const int t = threadIdx.x % L;
    const int CHS = ch_stride<NN>(p.ct);
    float2* buf = lds + team * CHS;
    float2* tw1 = lds + p.ct * CHS;
    const float2* tw2 = tw1 + G::TW1;
    float* inv_env = reinterpret_cast<float*>(tw1 + G::TW_LEN);
    for (int i = threadIdx.x; i < G::TW_LEN; i += blockDim.x) tw1[i] = p.tw[i];
    for (int m = threadIdx.x; m < STEP; m += blockDim.x) {
        const double w0 = (double)p.window[m], w1 = (double)p.window[m + STEP];
        const double e = w0 * w0 + w1 * w1;
        inv_env[m] = (float)(1.0 / (e < 1e-4 ? 1e-4 : e));
    }
    // window[t + L m] * scale: in registers, except at 2048 points (128 registers: they are read again per pair)
    constexpr bool WIN_REG = NN < 2048;
    float win[WIN_REG ? 16 : 1];
    if constexpr (WIN_REG) {
#pragma unroll
        for (int m = 0; m < 16; ++m) win[m] = p.window[t + L * m] * p.scale;
    }
    auto wv = [&](int m) { return WIN_REG ? win[WIN_REG ? m : 0] : p.window[t + L * m] * p.scale; };
    const int c0 = blockIdx.y * p.ct;
    const int ctv = min(p.ct, p.n_ch - c0);
    const int c = c0 + team;
    const bool live = c < p.n_ch;
    const int64_t F = p.n_frames, Cn = p.n_ch;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(  // (smaller than 4 GB: the host checks)
        const_cast<float2*>(p.stft), 0, (int)(uint32_t)((int64_t)p.n_bins * F * Cn * 8), 0x00020000);
    const int lct = __ffs(p.ct) - 1;
    const int cl = threadIdx.x & (p.ct - 1);
    float2* img = lds + cl * CHS;
    const int n_fp = (p.n_frames + 1) >> 1;
    const int fp0 = blockIdx.x * p.fpw, fp1 = min(fp0 + p.fpw, n_fp);
    if (fp0 >= fp1) return;
    float* oc = q.out + (int64_t)(live ? c : 0) * q.ld;
    auto emit = [&](int64_t pos, float sum) {  // the ends of the signal: one frame slot, or none
        if (pos < 0 || pos >= q.total_length) return;
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
        oc[pos] = (float)((double)sum / (env < 1e-4 ? 1e-4 : env));
    };
    float carry[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) carry[m] = 0.f;
    if (fp0 == 0 && live)
        for (int64_t n = t; n < (int64_t)q.off * STEP; n += L) emit(n, 0.f);
    for (int fp = fp0 > 0 ? fp0 - 1 : 0; fp < fp1; ++fp) {
        const int f0 = 2 * fp;
        const bool v1 = f0 + 1 < p.n_frames;
        const bool owned = fp >= fp0;  // the pair in front of the range only yields the carry
        __syncthreads();  // tables written / the previous pair's images have been consumed
        {
            // Bins k0 + (NN / 16) j of channel cl, frames f0 and f0 + 1: all eighteen loads are requested before the first
            // value is placed, as raw-buffer loads whose offset lies behind the end where there is nothing to read.
            // (Until round 4: a rolled loop of plain loads inside `if (channel and bin exist)`, each followed by a full
            // wait -- nine memory round trips per frame pair with every wave of the workgroup in step.)
            constexpr int KS = NN / 16;  // = blockDim.x >> lct
            const int k0 = (int)threadIdx.x >> lct;
            float2 ga[9], gb[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int k = k0 + KS * j;
                const bool ok = cl < ctv && k <= NN / 2 && k < p.n_bins;
                const uint32_t off = ok ? (uint32_t)((((int64_t)k * F + f0) * Cn + c0 + cl) * 8) : 0xfffffff0u;
                const uint32_t off2 = (ok && v1) ? off + (uint32_t)(Cn * 8) : 0xfffffff0u;
                ga[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(srs, (int)off, 0, 0));
                gb[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(srs, (int)off2, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int k = k0 + KS * j;
                const float2 A = ga[j], B = gb[j];
                if (k == 0 || k == NN / 2) {
                    img[k] = make_float2(A.x, -B.x);  // conj(A.x + i B.x)
                } else if (k < NN / 2) {
                    img[k] = make_float2(A.x - B.y, -A.y - B.x);      // conj(A + i B)
                    img[NN - k] = make_float2(A.x + B.y, A.y - B.x);  // conj(conj A + i conj B)
                }
            }
        }
        __syncthreads();
        float2 v[16], z[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = buf[t + L * n1];
        team_sync<NN>();  // the image is in registers: the transform may use the region
        fft_wave<NN>(v, z, buf, tw1, tw2, t);
        if (live) {
            const int64_t P0 = (int64_t)(f0 + q.off) * STEP;
            // both covering frame slots exist: fs - 1 >= 0 and fs < n_total (uniform per segment)
            const bool fast0 = f0 + q.off >= 1 && f0 + q.off < q.n_total;
            const bool fast1 = f0 + q.off + 1 < q.n_total;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int n = t + L * m;
                // frame f0: Re, frame f0 + 1: -Im (the result is the conjugate of a + i b)
                const float wl = wv(m), wh = wv(m + 8);
                const float a_lo = z[m].x * wl, b_lo = -z[m].y * wl;
                const float a_hi = z[m + 8].x * wh, b_hi = -z[m + 8].y * wh;
                if (owned) {
                    const float s0 = carry[m] + a_lo;  // second half of the frame before + first half of f0
                    const float s1 = a_hi + b_lo;      // second half of f0 + first half of f0 + 1
                    if (fast0 && P0 + n < q.total_length)
                        oc[P0 + n] = s0 * inv_env[n];
                    else
                        emit(P0 + n, s0);
                    if (fast1 && P0 + STEP + n < q.total_length)
                        oc[P0 + STEP + n] = s1 * inv_env[n];
                    else
                        emit(P0 + STEP + n, s1);
                }
                carry[m] = b_hi;  // second half of f0 + 1 (zero if it does not exist)
            }
        }
    }
    // behind the last frame: its second half, then nothing but the envelope's floor
    if (fp1 == n_fp && live) {
        const int64_t P = (int64_t)(2 * n_fp + q.off) * STEP;
#pragma unroll
        for (int m = 0; m < 8; ++m) emit(P + t + L * m, carry[m]);
        for (int64_t n = P + STEP + t; n < q.total_length; n += L) emit(n, 0.f);
    }
}

}  // namespace stft1k
