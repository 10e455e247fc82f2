// Welch with a 2048-sample window at 50 % overlap on the 4096-point register machine of kernels_welch4096w.hpp
// (round 5; VERDICT r4 next 8: the 2048-sample window sat at 1.5 x the per-sample time of its neighbours on the
// wave / team kernels of kernels_welch1024.hpp -- two waves per transform, a workgroup barrier inside the pair loop).
//
// 4096 = 16 x 256 and 2048 = 8 x 256: the machine's passes 2 and 3 are sixteen independent 256-point transforms over the
// thread index, one per row of the pass-1 image.  So ONE pass of the machine carries TWO 2048-point transforms -- frame
// pairs (2 P, 2 P + 1) of one channel, P = 2 q and 2 q + 1, i.e. the four frames 4 q ... 4 q + 3 -- if only pass 1 changes:
//     thread t holds  zA[t + 256 m] in v[m] and zB[t + 256 m] in v[8 + m], m = 0 .. 7
//     pass 1:  two 8-point DFTs over m, times W2048^(t k1) (seven per-thread constants, shared by A and B)
//              -> image rows 0 .. 7 (A, k1 = row) and 8 .. 15 (B, k1 = row - 8)
//     passes 2, 3: unchanged; thread (row r = tid >> 4, k2 = tid & 15) ends with the bins
//              k = k1 + 8 k2 + 128 k3 (k3 = 0 .. 15, in v[pos16(k3)]) of transform A (r < 8) or B.
// Everything else is welch4096::k_y3: T += conj(W) Z and P += |Z|^2 per thread in registers (the A and the B threads
// accumulate different frames of the same channel; the fold at the end of the chunk adds them), the next pass's sixteen new
// samples requested between the butterflies (hop 1024: the four frames of a pass cover 20 quarter blocks per thread, the
// last four are the first four of the next pass), window in LDS (8 KB), 45 KB of LDS and <= 168 registers: three
// workgroups per CU.  Per pass 4096 new samples of a channel, as for the 4096-sample window.
#pragma once
#include "kernels_welch4096w.hpp"

namespace welch2048h {

namespace w4 = welch4096;
using w4::Args;
using w4::cmul;
using w4::L1S;
using w4::L3S;
using w4::NT;
using w4::pos16;
constexpr int W = 2048, NBW = W / 2 + 1, HOP = 1024;
constexpr int PASS = 4096;  // complex values of a pass (two transforms); also the samples a pass advances by
constexpr int LDS_BYTES = 16 * L1S * 8 + 256 * 8 + W * 4;  // exchange + W256 + window = 45056

struct Tw7 {
    float2 w[7];  // W2048^(t k1), k1 = 1 .. 7 = W4096^(t 2 k1): rows 2 k1 - 1 of welch4096::host_tables()
};
__device__ __forceinline__ void load_tw7(Tw7& tw, const float2* __restrict__ twt, int tid) {
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) tw.w[k1 - 1] = twt[(2 * k1 - 1) * 256 + tid];
}

// 8-point DFT in place on v[b .. b + 7] (input index m, forward): X[k] lands in v[b + 4 (k & 1) + (k >> 1)]
template <int B>
__device__ __forceinline__ void dft8(float2 (&v)[16]) {
    constexpr float R2 = 0.70710678118654752440f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 a = v[B + j], c = v[B + j + 4];
        v[B + j] = make_float2(a.x + c.x, a.y + c.y);
        v[B + j + 4] = make_float2(a.x - c.x, a.y - c.y);
    }
    {  // odd half times W8^j
        const float2 o1 = v[B + 5], o2 = v[B + 6], o3 = v[B + 7];
        v[B + 5] = make_float2((o1.x + o1.y) * R2, (o1.y - o1.x) * R2);   // W8^1 = R2 (1 - i)
        v[B + 6] = make_float2(o2.y, -o2.x);                              // W8^2 = -i
        v[B + 7] = make_float2((o3.y - o3.x) * R2, -(o3.x + o3.y) * R2);  // W8^3 = R2 (-1 - i)
    }
    w4::r4(v[B], v[B + 1], v[B + 2], v[B + 3]);      // X[0], X[2], X[4], X[6]
    w4::r4(v[B + 4], v[B + 5], v[B + 6], v[B + 7]);  // X[1], X[3], X[5], X[7]
}
__device__ __forceinline__ constexpr int pos8(int k) { return 4 * (k & 1) + (k >> 1); }

// The machine with the two-transform pass 1 (passes 2 and 3 as welch4096::fft4096_wi: same LDS image, same counted
// waits, same barriers).  ld_a(h), h = 0, 1: call-outs behind the two 8-point transforms (eight sample loads each);
// ld_b(g), g = 0 .. 3: call-outs of pass 2's stage B (the input spectrum).
template <typename LA, typename LB>
__device__ __forceinline__ void fft2x2048_wi(float2 (&v)[16], const Tw7& tw, float2* __restrict__ buf,
                                             const float2* __restrict__ tw2, int tid, LA ld_a, LB ld_b) {
    using w4::lds_rd64;
    using w4::lgkm_wait;
    using w4::static_for;
    const int k1u = tid >> 4, n3 = tid & 15;
    float2* __restrict__ col = buf + tid;
    auto half = [&](auto hc) {
        constexpr int h = decltype(hc)::value;
        dft8<8 * h>(v);
        W4_PIN();
        ld_a(h);
        W4_PIN();
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            float2 z = v[8 * h + pos8(k1)];
            if (k1) z = cmul(z, tw.w[k1 - 1]);
            W4_PIN();
            col[(8 * h + k1) * L1S] = z;
            W4_PIN();
        }
    };
    half(std::integral_constant<int, 0>{});
    half(std::integral_constant<int, 1>{});
    __syncthreads();
    float2* __restrict__ row = buf + k1u * L1S;
    const uint32_t a_row = w4::lds_addr(row + n3), a_tw2 = w4::lds_addr(tw2 + n3);
    // pass-2 inputs v[n2] = row[16 n2 + n3], requested in the order the butterflies consume them
    W4_PIN();
    static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value, n2 = 4 * (i & 3) + (i >> 2);
        lds_rd64<16 * n2 * 8>(v[n2], a_row);
    });
    float2 w2[16];
    w4::dft16_h(
        v,
        [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            lgkm_wait<(g == 0 ? 12 : 11)>();
        },
        [&](int g) {
            W4_PIN();
            static_for<4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (g == 0 && j > 0) lds_rd64<(0 + j) * 16 * 8>(w2[0 + j], a_tw2);
                if (g == 1) lds_rd64<(4 + j) * 16 * 8>(w2[4 + j], a_tw2);
                if (g == 2) lds_rd64<(8 + j) * 16 * 8>(w2[8 + j], a_tw2);
                if (g == 3) lds_rd64<(12 + j) * 16 * 8>(w2[12 + j], a_tw2);
            });
            if (g == 3) lgkm_wait<0>();  // every table value is in before stage B multiplies
            W4_PIN();
        },
        [&](int g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k2 = g + 4 * j;
                float2 z = v[4 * g + j];
                if (k2) z = cmul(z, w2[k2]);
                W4_PIN();
                row[n3 * L3S + k2] = z;
                W4_PIN();
            }
            W4_PIN();
            ld_b(g);
            W4_PIN();
        });
    w4::wave_sync();
    W4_PIN();
    static_for<16>([&](auto ic) {  // lane now plays k2 = n3
        constexpr int i = decltype(ic)::value, j = 4 * (i & 3) + (i >> 2);
        lds_rd64<j * L3S * 8>(v[j], a_row);
    });
    w4::dft16_h(
        v,
        [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            lgkm_wait<12 - 4 * g>();
            if (g == 3) __syncthreads();  // all reads of the image are back: the next pass's stores may begin
        },
        w4::NoHookI(), w4::NoHookI());
}

// position of this thread's bin k3 in the 2 x 2048 fold image (A at 0, B at 2048), and the padded LDS slot of a position
__device__ __forceinline__ int bin_base(int tid) { return ((tid >> 7) << 11) + ((tid >> 4) & 7) + 8 * (tid & 15); }
__device__ __forceinline__ int fold_pos(int p) { return p + (p >> 4); }

// zero the frames of pass `pr` that lie at or past n_frames (only a caller that asks for fewer frames than the signal
// holds gets there: past the signal the buffer range check has delivered zeros already)
__device__ __forceinline__ void drop_frames(float2 (&v)[16], const Args& p, int pr) {
    const int f0 = 4 * pr;
    if (f0 + 3 < p.n_frames) return;
    asm volatile("" ::: "memory");  // a real (wave-uniform) branch: as selects this was 24 instructions in every pass
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (f0 + 0 >= p.n_frames) v[m].x = 0.f;
        if (f0 + 1 >= p.n_frames) v[m].y = 0.f;
        if (f0 + 2 >= p.n_frames) v[8 + m].x = 0.f;
        if (f0 + 3 >= p.n_frames) v[8 + m].y = 0.f;
    }
}

// ---- input spectra: one pass per workgroup.  Args::n_pairs counts PASSES here (xs[cx][pass][4096], px[cx][pass][NBW]).
__global__ __launch_bounds__(NT) void k_x2h(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * L1S;
    const int tid = threadIdx.x;
    const int cx = (int)blockIdx.x / p.n_pairs, pr = (int)blockIdx.x - cx * p.n_pairs;
    Tw7 tw;
    float2 v[16];
    {
        const __amdgpu_buffer_rsrc_t rs = w4::channel_rsrc(p.sig + (int64_t)cx * p.ld, p.n_samples);
        const int off0 = 4 * (PASS * pr + tid);
        float s[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) s[j] = w4::ld_sample(rs, off0 + 1024 * j);
        load_tw7(tw, p.twt, tid);
        tw2[tid] = p.twt[15 * 256 + tid];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float w = p.window[tid + 256 * m];
            v[m] = make_float2(s[m] * w, s[4 + m] * w);
            v[8 + m] = make_float2(s[8 + m] * w, s[12 + m] * w);
        }
        drop_frames(v, p, pr);
    }
    __syncthreads();  // the W256 table
    auto none = [](int) {};
    fft2x2048_wi(v, tw, buf, tw2, tid, none, none);
    if (p.detrend && (tid & 127) == 0) v[pos16(0)] = make_float2(0.f, 0.f);  // bin 0 of A (thread 0) and of B (thread 128)
    float4* xo = reinterpret_cast<float4*>(p.xs + ((int64_t)cx * p.n_pairs + pr) * PASS) + tid;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float2 z0 = v[pos16(2 * g)], z1 = v[pos16(2 * g + 1)];
        xo[256 * g] = make_float4(z0.x, z0.y, z1.x, z1.y);
    }
    // symmetrised power of both transforms for Sxx
    float* pw = reinterpret_cast<float*>(buf);
    const int bb = bin_base(tid);
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
        const float2 z = v[pos16(k3)];
        pw[fold_pos(bb + 128 * k3)] = z.x * z.x + z.y * z.y;
    }
    __syncthreads();
    float* po = p.px + ((int64_t)cx * p.n_pairs + pr) * NBW;
    for (int k = tid; k < NBW; k += NT) {
        const int km = (W - k) & (W - 1);
        po[k] = 0.5f * ((pw[fold_pos(k)] + pw[fold_pos(km)]) + (pw[fold_pos(W + k)] + pw[fold_pos(W + km)]));
    }
}

// paired inputs: px rows of every input channel summed over the passes of each chunk (fp64); grid = (n_chunks, n_cx)
__global__ __launch_bounds__(256) void k_px_sum(Args p) {
    const int q = blockIdx.x, cx = blockIdx.y;
    int p0, p1;
    w4::chunk_range(p, q, p0, p1);
    for (int k = threadIdx.x; k < NBW; k += 256) {
        double sum = 0.0;
        for (int pr = p0; pr < p1; ++pr) sum += (double)p.px[((int64_t)cx * p.n_pairs + pr) * NBW + k];
        p.psx[((int64_t)q * p.n_cx + cx) * NBW + k] = (float)sum;
    }
}

// ---- output channels: workgroup = (chunk of passes, channel) -------------------------------------------------------
template <bool AUTO = false>
__global__ __launch_bounds__(NT, 3) void k_y2h(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * L1S;
    float* winl = reinterpret_cast<float*>(lds + 16 * L1S + 256);
    const int tid = threadIdx.x;
    int q, c;
    {  // XCD-aware decode as in k_y3: each XCD takes a contiguous run of (chunk, channel) units in chunk-major order
        const int b = blockIdx.x, total = p.n_chunks * p.n_ch;
        const int u = (total & 7) == 0 ? (b & 7) * (total >> 3) + (b >> 3) : b;
        q = u / p.n_ch;
        c = u - q * p.n_ch;
    }
    Tw7 tw;
    load_tw7(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
#pragma unroll
    for (int m = 0; m < 8; ++m) winl[tid + 256 * m] = p.window[tid + 256 * m];
    const float* ch = p.sig + (int64_t)c * p.ld;
    int p0, p1;
    w4::chunk_range(p, q, p0, p1);
    const int cx = p.n_cx > 1 ? c : 0;
    if (!AUTO && p.n_cx <= 1) {
        // input auto spectrum of this chunk: every workgroup of the chunk sums a slice of the bins over the chunk's px rows
        double* red = reinterpret_cast<double*>(lds);
        const int bpc = (NBW + p.n_ch - 1) / p.n_ch;
        const int b0 = c * bpc, b1 = min(b0 + bpc, NBW);
        const int lw = bpc <= 32 ? 5 : (bpc <= 64 ? 6 : (bpc <= 128 ? 7 : 8)), width = 1 << lw, rows = NT >> lw;
        const int rg = tid >> lw, kl = tid & (width - 1);
        for (int kb = b0; kb < b1; kb += width) {
            const int k = kb + kl;
            double sum = 0.0;
            if (k < b1)
                for (int pr = p0 + rg; pr < p1; pr += rows) sum += (double)p.px[(int64_t)pr * NBW + k];
            red[rg * width + kl] = sum;
            __syncthreads();
            if (rg == 0 && k < b1) {
                double t = 0.0;
                for (int j = 0; j < rows; ++j) t += red[j * width + kl];
                p.psx[(int64_t)q * NBW + k] = (float)t;
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
    // quarter blocks j = 0 .. 19 of the current pass: s[j] = ch[4096 pass + tid + 256 j]; 0 .. 3 carried, 4 .. 19 in nx
    float carry[4], nx[16];
    const __amdgpu_buffer_rsrc_t rs = w4::channel_rsrc(ch, p.n_samples);
    const __amdgpu_buffer_rsrc_t xrs =
        AUTO ? rs
             : __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.xs + ((int64_t)cx * p.n_pairs + p0) * PASS), 0,
                                                 (int)(uint32_t)((p1 - p0) * (PASS * 8)), 0x00020000);
    if (p0 < p1) {
        const int off0 = 4 * (PASS * p0 + tid);
#pragma unroll
        for (int j = 0; j < 4; ++j) carry[j] = w4::ld_sample(rs, off0 + 1024 * j);
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = w4::ld_sample(rs, off0 + 1024 * (4 + j));
    }
    __syncthreads();  // window and W256 table in LDS
    // the window values of a pass are requested from LDS at the end of the previous one (into registers the
    // accumulation has just freed): they are not live during the transform
    float winr[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) winr[m] = winl[tid + 256 * m];
    for (int pr = p0; pr < p1; ++pr) {
        float2 v[16];
        const int level16 = ((p1 - pr - 1) * 16) / (p1 - p0);
        w4::set_prio(level16, (pr * 5) & 3);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float w = winr[m];
            const float s0 = m < 4 ? carry[m] : nx[m - 4];
            v[m] = make_float2(s0 * w, nx[m] * w);              // frames 4 pr, 4 pr + 1: s[m], s[4 + m]
            v[8 + m] = make_float2(nx[4 + m] * w, nx[8 + m] * w);  // frames 4 pr + 2, 4 pr + 3: s[8 + m], s[12 + m]
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) carry[j] = nx[12 + j];
        drop_frames(v, p, pr);
        float2 xw[16];
        const int off1 = 4 * (PASS * (pr + 1) + tid) + 1024 * 4;
        const int xoff = (pr - p0) * (PASS * 8) + tid * 16;
        fft2x2048_wi(
            v, tw, buf, tw2, tid,
            [&](int h) {  // samples of the next pass, eight per call-out (past the signal the range check gives 0)
#pragma unroll
                for (int j = 0; j < 8; ++j) nx[8 * h + j] = w4::ld_sample(rs, off1 + 1024 * (8 * h + j));
            },
            [&](int g) {  // input spectrum of this pass, two 16-byte loads per call-out
                if (!AUTO) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 q4 =
                            __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff + 4096 * (2 * g + j), 0, 0));
                        xw[2 * (2 * g + j)] = make_float2(q4.x, q4.y);
                        xw[2 * (2 * g + j) + 1] = make_float2(q4.z, q4.w);
                    }
                }
            });
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) {
            const float2 z = v[pos16(k3)];
            if (!AUTO) {
                const float2 w = xw[k3];
                T[k3].x = fmaf(w.x, z.x, fmaf(w.y, z.y, T[k3].x));
                T[k3].y = fmaf(w.x, z.y, fmaf(-w.y, z.x, T[k3].y));
            }
            P[k3] = fmaf(z.x, z.x, fmaf(z.y, z.y, P[k3]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) winr[m] = winl[tid + 256 * m];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (p.detrend && (tid & 127) == 0) P[0] = 0.f;  // (xs bin 0 is already 0 -> T[0] = 0)
    // fold k <-> W - k and add the A and the B threads' sums, once per chunk, through LDS
    const int bb = bin_base(tid);
    __syncthreads();
    const int64_t so = ((int64_t)q * p.n_ch + c) * NBW;
    if (!AUTO) {
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) buf[fold_pos(bb + 128 * k3)] = T[k3];
        __syncthreads();
        for (int k = tid; k < NBW; k += NT) {
            const int km = (W - k) & (W - 1);
            const float2 a = buf[fold_pos(k)], b = buf[fold_pos(km)], a2 = buf[fold_pos(W + k)], b2 = buf[fold_pos(W + km)];
            p.pxy[so + k] = make_float2(0.5f * ((a.x + b.x) + (a2.x + b2.x)), 0.5f * ((a.y - b.y) + (a2.y - b2.y)));
        }
        __syncthreads();
    }
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) pw[fold_pos(bb + 128 * k3)] = P[k3];
    __syncthreads();
    for (int k = tid; k < NBW; k += NT) {
        const int km = (W - k) & (W - 1);
        p.pyy[so + k] = 0.5f * ((pw[fold_pos(k)] + pw[fold_pos(km)]) + (pw[fold_pos(W + k)] + pw[fold_pos(W + km)]));
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct Plan {
    int n_passes, n_chunks;
    size_t bytes;
};
inline Plan plan(int n_frames, int n_cy, int n_cx, int want_chunks = 0) {
    Plan pl;
    pl.n_passes = (n_frames + 3) / 4;
    pl.n_chunks = w4::chunks_for3(pl.n_passes, n_cy, want_chunks);
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    pl.bytes = pad(sizeof(float2) * (size_t)n_cx * pl.n_passes * PASS) + pad(sizeof(float) * (size_t)n_cx * pl.n_passes * NBW) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * NBW) + pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * NBW) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cy * NBW);
    return pl;
}
// the raw-buffer loads carry byte offsets in 32 bits
inline bool fits(int64_t n_samples, int n_frames) {
    return n_samples < ((int64_t)1 << 30) - 16384 && ((int64_t)n_frames + 8) * HOP < ((int64_t)1 << 30) - 16384;
}

}  // namespace welch2048h
