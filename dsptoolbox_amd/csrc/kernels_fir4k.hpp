// FIR filter bank, up to 4097 taps: uniformly partitioned overlap-save on the headline kernel's
// 4096-point register transform (three independent 256-thread workgroups per CU).  gfx950.
// (Filter.filter_signal / FilterBank.filter_signal: _lfilter_fir, classes/filter_helpers.py:454-503,
// y = oaconvolve(x, b)[:N]; _filterbank_on_signal :385-451.)
//
//   hop B = 2048 new samples per block, transform length 4096, partitions of 2049 taps:
//       h_0[j] = h[j]            j = 0 .. 2048
//       h_p[j] = h[2048 p + j]   j = 1 .. 2048 ,  h_p[0] = 0        (p >= 1)
//   so that partition p meets the input delayed by exactly p blocks.  With
//       X_b = FFT4096( x[(b-1) B .. (b+1) B) )        (two channels ride one transform: xa + i xb)
//       H_p = FFT4096( h_p, zero-padded ) / 4096
//   the block's output is the second half of IFFT4096( X_b H_0 + X_{b-1} H_1 + ... ): the circular
//   convolution of a 4096-sample segment with 2049 taps is exact from sample 2048 on.
//
//   A workgroup owns a run of blocks of one channel pair.  Per block: ONE forward transform, its
//   spectrum stays in registers (X_b, and X_{b-1} handed down from the previous block) while ALL
//   filters are applied: 16-byte loads of the tap spectra (stored in the transform's own register
//   layout by k_taps, 2 MB for 32 x 2 partitions: L2 resident), multiply-add, inverse transform
//   (the mirror image of welch4096::fft4096_w: one LDS image, the 16 x 16 transpose wave-local in the
//   row, two workgroup barriers), 2 x 8 coalesced 4-byte stores per thread.  Every output sample is
//   produced and stored once; signal edges are the buffer range check (loads return 0 before the
//   first and past the last sample, stores past the end are dropped): no edge code path.
//
//   Against kernels_fir16k.hpp (16384-point blocks, one 1024-thread workgroup per CU): 24 instead of
//   18.7 transform points per output sample, but the 4096-point transform runs at 2.4 x the rate in
//   three independent workgroups per CU (226 against 94 transforms per microsecond, measured on the
//   Welch / FIR benchmarks of round 2).
#pragma once
#include "kernels_fir16k.hpp"
#include "kernels_welch4096w.hpp"

namespace fir4k {

namespace w4 = welch4096;
using fir16k::cmulc;
using fir16k::idft16;
using w4::cmul;
constexpr int N = 4096, NT = 256, HOP = 2048, PART = 2049;
constexpr int LDS_BYTES = (16 * w4::L1S + 256) * 8 + 2304;  // exchange image + W256 table + its padded copy
// Two partitions: X_b, X_{b-1}, the working set and two tap spectra in flight are 192 registers; the 168
// of three workgroups per CU were tried (half of X_{b-1} parked in LDS, staged tap loads: hipcc kept
// 10-46 spilled values inside the filter loop) -- that kernel is built for TWO workgroups per CU.

inline int partitions(int n_taps) { return n_taps <= 1 ? 1 : (n_taps - 2) / HOP + 1; }  // ceil((T - 1) / 2048)

__device__ __forceinline__ void apply_tw6_conj(float2 (&v)[16], const w4::Tw6& tw) {
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const int lo = k1 & 3, hi = k1 >> 2;
        float2 z = v[w4::pos16(k1)];
        if (lo) z = cmulc(z, tw.a[lo - 1]);
        if (hi) z = cmulc(z, tw.b[hi - 1]);
        v[w4::pos16(k1)] = z;
    }
}

// Mirror image of welch4096::fft4096_w: v[pos16(k3)] = Y[bt + 256 k3] (bt = bin_thread(tid)) on entry,
// v[n1] = sum_k Y[k] W4096^(-k (tid + 256 n1)) on return (unscaled).  `buf`: the workgroup's one
// exchange image.  Two workgroup barriers: in front of the first write of the image (the column reads
// of the previous transform) and behind the last.
//   after_t():  called behind the wave-local transpose (the caller's loads go there)
template <typename HT = w4::NoHook, typename HB = w4::NoHook>
__device__ __forceinline__ void ifft4096_w(float2 (&v)[16], const w4::Tw6& tw, float2* __restrict__ buf,
                                           const float2* __restrict__ tw2, int tid, HT after_t = HT(), HB after_b = HB()) {
    const int k1u = tid >> 4, n3 = tid & 15;
    idft16(v);  // over k3: v[j], the lane plays k2 = n3
    float2* __restrict__ row = buf + k1u * w4::L1S;
    __syncthreads();  // every wave has read its columns (last pass of the previous transform)
#pragma unroll
    for (int j = 0; j < 16; ++j) row[j * w4::L3S + n3] = v[j];
    w4::wave_sync();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) v[w4::pos16(k2)] = row[n3 * w4::L3S + k2];  // lane now plays n3
    after_t();
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) v[w4::pos16(k2)] = cmulc(v[w4::pos16(k2)], tw2[k2 * 16 + n3]);
    idft16(v);  // over k2: v[n2]
    w4::wave_sync();  // the transpose reads of this row (the same 16 lanes) are done
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) row[16 * n2 + n3] = v[n2];
    after_b();
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[w4::pos16(k1)] = buf[k1 * w4::L1S + tid];
    apply_tw6_conj(v, tw);
    idft16(v);  // over k1: v[n1]
}

// idft16 with call-outs (mirror of welch4096::dft16_h): pre_a(g) / after_a(g) around the four first-stage
// butterflies (inputs X[k] in v[4 g + i] = pos16(g + 4 i)), after_b(g) behind the four second-stage
// butterflies: x[g], x[g + 4], x[g + 8], x[g + 12] are final in v[g + 4 i].
template <typename PA, typename HA, typename HB>
__device__ __forceinline__ void idft16_h(float2 (&v)[16], PA pre_a, HA after_a, HB after_b) {
    constexpr float C8 = 0.92387953251128673848f, S8 = 0.38268343236508978178f;
    constexpr float R2 = 0.70710678118654752440f;
    using fir16k::r4i;
    auto mulw = [](float2 z, float c, float s) {  // z * (c + i s)
        return make_float2(fmaf(z.x, c, -z.y * s), fmaf(z.y, c, z.x * s));
    };
    pre_a(std::integral_constant<int, 0>{});
    r4i(v[0], v[1], v[2], v[3]);
    after_a(0);
    pre_a(std::integral_constant<int, 1>{});
    r4i(v[4], v[5], v[6], v[7]);
    v[5] = mulw(v[5], C8, S8);                                                  // W16^-1
    v[6] = make_float2((v[6].x - v[6].y) * R2, (v[6].x + v[6].y) * R2);         // W16^-2
    v[7] = mulw(v[7], S8, C8);                                                  // W16^-3
    after_a(1);
    pre_a(std::integral_constant<int, 2>{});
    r4i(v[8], v[9], v[10], v[11]);
    v[9] = make_float2((v[9].x - v[9].y) * R2, (v[9].x + v[9].y) * R2);         // W16^-2
    v[10] = make_float2(-v[10].y, v[10].x);                                     // W16^-4 = +i
    v[11] = make_float2(-(v[11].x + v[11].y) * R2, (v[11].x - v[11].y) * R2);   // W16^-6
    after_a(2);
    pre_a(std::integral_constant<int, 3>{});
    r4i(v[12], v[13], v[14], v[15]);
    v[13] = mulw(v[13], S8, C8);                                                // W16^-3
    v[14] = make_float2(-(v[14].x + v[14].y) * R2, (v[14].x - v[14].y) * R2);   // W16^-6
    v[15] = mulw(v[15], -C8, -S8);                                              // W16^-9
    after_a(3);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        r4i(v[g], v[g + 4], v[g + 8], v[g + 12]);
        after_b(g);
    }
}

// ifft4096_w with the exchange traffic spread between the butterflies (mirror of
// welch4096::fft4096_wi): the transposed stores of the first pass leave four at a time behind the
// butterfly that made them, the row stores of the second pass likewise; the barrier in front of the
// first image write sits behind the first pass's stage A (whose arithmetic needs no LDS), the reads of
// the second and third pass are requested in the order their butterflies consume them and waited for
// with counted lgkmcnt (8-byte reads that stay single: welch4096::lds_rd64).  The W256 twiddles of the
// middle exchange ride on the first pass's OUTPUTS (lane k2, values j: row k2 of the symmetric table,
// eight 16-byte reads from `tw2p`, a copy with rows of 18 values: conflict-free) instead of on the
// second pass's inputs (fifteen 8-byte reads on the critical path of its first butterfly).
//   ld_a(g), g = 0..3: call-outs of the first pass's stage A;  ld_b(g): of the second pass's stage B
constexpr int TW2P_ROW = 18;                    // float2 per padded table row
constexpr int TW2P_BYTES = 16 * TW2P_ROW * 8;   // 2304
__device__ __forceinline__ void fill_tw2p(float2* tw2p, const float2* __restrict__ twt, int tid) {
    tw2p[(tid >> 4) * TW2P_ROW + (tid & 15)] = twt[15 * 256 + tid];  // W256^((tid >> 4) (tid & 15))
}
template <typename LA, typename LB>
__device__ __forceinline__ void ifft4096_wi(float2 (&v)[16], const w4::Tw6& tw, float2* __restrict__ buf,
                                            const float2* __restrict__ tw2p, int tid, LA ld_a, LB ld_b) {
    using w4::lds_rd64;
    using w4::lgkm_wait;
    using w4::static_for;
    const int k1u = tid >> 4, n3 = tid & 15;
    float2* __restrict__ row = buf + k1u * w4::L1S;
    // ---- over k3 (inputs in registers); the lane plays k2 = n3: x[j] W256^(-j k2) -> row[j L3S + k2]
    float2 w2[16];
    {
        const float4* __restrict__ tr = reinterpret_cast<const float4*>(tw2p + n3 * TW2P_ROW);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 q = tr[i];
            w2[2 * i] = make_float2(q.x, q.y);
            w2[2 * i + 1] = make_float2(q.z, q.w);
        }
    }
    idft16_h(
        v, w4::NoHookI(),
        [&](int g) {
            W4_PIN();
            ld_a(g);
            W4_PIN();
            if (g == 3) __syncthreads();  // every wave has read its columns (last pass of the previous transform)
        },
        [&](int g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = g + 4 * i;
                float2 z = v[j];
                if (j) z = cmulc(z, w2[j]);
                W4_PIN();
                row[j * w4::L3S + n3] = z;
                W4_PIN();
            }
        });
    w4::wave_sync();
    // ---- over k2: slot 4 a + b <-> k2 = a + 4 b, butterfly a consumes slots 4 a .. 4 a + 3
    const uint32_t a_rt = w4::lds_addr(row + n3 * w4::L3S);
    W4_PIN();
    static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value, a = i >> 2, b = i & 3;
        lds_rd64<(a + 4 * b) * 8>(v[4 * a + b], a_rt);
    });
    idft16_h(
        v,
        [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            lgkm_wait<12 - 4 * g>();
        },
        w4::NoHookI(),
        [&](int g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                W4_PIN();
                row[16 * (g + 4 * i) + n3] = v[g + 4 * i];
                W4_PIN();
            }
            W4_PIN();
            ld_b(g);
            W4_PIN();
        });
    __syncthreads();
    // ---- over k1: v[pos16(k1)] = buf[k1 L1S + tid] W4096^(-tid k1)
    const uint32_t a_col = w4::lds_addr(buf + tid);
    W4_PIN();
    static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value, a = i >> 2, b = i & 3;
        lds_rd64<(a + 4 * b) * w4::L1S * 8>(v[4 * a + b], a_col);
    });
    idft16_h(
        v,
        [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            lgkm_wait<12 - 4 * g>();
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float2 z = v[4 * g + b];
                if constexpr (g > 0) z = cmulc(z, tw.a[g - 1]);  // k1 = g + 4 b: lo = g, hi = b
                if (b) z = cmulc(z, tw.b[b - 1]);
                v[4 * g + b] = z;
            }
        },
        w4::NoHookI(), w4::NoHookI());
}

// ---- tap spectra in register layout ---------------------------------------------------------------
// hp[((f P + p) 8 + g) 256 + tid] = (H_p[slot 2g], H_p[slot 2g + 1]) of thread tid, slot s = pos16(k3)
// <-> bin bin_thread(tid) + 256 k3;  1 / 4096 folded in.  grid = n_filt * P.
struct TapArgs {
    const float* taps;  // [n_filt][n_taps]
    int n_filt, n_taps, n_part;
    const float2* twt;  // welch4096::host_tables()
    float4* hp;
};
__global__ __launch_bounds__(NT) void k_taps(TapArgs p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    const int tid = threadIdx.x;
    const int f = (int)blockIdx.x / p.n_part, part = (int)blockIdx.x - f * p.n_part;
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    const float* h = p.taps + (int64_t)f * p.n_taps;
    float2 v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int j = tid + 256 * n1, src = HOP * part + j;
        const bool in = j <= HOP && (part == 0 || j >= 1) && src < p.n_taps;
        v[n1] = make_float2(in ? h[src] : 0.f, 0.f);
    }
    w4::fft4096_w(v, tw, buf, tw2, tid);
    const float s = 1.0f / (float)N;
    float4* o = p.hp + (int64_t)blockIdx.x * (8 * 256) + tid;
#pragma unroll
    for (int g = 0; g < 8; ++g) o[256 * g] = make_float4(v[2 * g].x * s, v[2 * g].y * s, v[2 * g + 1].x * s, v[2 * g + 1].y * s);
}

struct Args {
    const float* x;
    int64_t n_samples, ldx, ld_y;
    int n_ch, n_filt;
    int n_blocks;  // ceil(n_samples / 2048) per channel
    int n_chunks;  // workgroups per channel pair
    const float2* twt;
    const float4* hp;  // [n_filt][P][8][256]
    float* y;          // [(f n_ch + c) ld_y + n]
};

// STAGE (round 5 experiment; DSPTOOLBOX_AMD_FIR_STAGE=1): the 2 x 8 four-byte stores of a thread
// (256 contiguous bytes per wave instruction) go through a per-wave LDS strip instead -- 16 ds_write_b32, 4
// ds_read_b128 -- and leave as 2 x 2 sixteen-byte stores (four 256-byte runs per wave instruction).
constexpr int STAGE_BYTES = 4 * 2 * 8 * 64 * 4;  // waves x channels x groups of 256 samples x lanes x float
// grid = ceil(n_ch / 2) * n_chunks
template <int P, bool STAGE = false>
__global__ __launch_bounds__(NT, P == 1 ? 3 : 2) void k_fir(Args p) {
    static_assert(P == 1 || P == 2, "one or two partitions of 2049 taps");
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    float* strip = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + LDS_BYTES) + (threadIdx.x >> 6) * (2 * 8 * 64);
    const int tid = threadIdx.x;
    const int cp = (int)blockIdx.x / p.n_chunks, q = (int)blockIdx.x - cp * p.n_chunks;
    const int b0 = (int)((int64_t)q * p.n_blocks / p.n_chunks), b1 = (int)((int64_t)(q + 1) * p.n_blocks / p.n_chunks);
    const int ca = 2 * cp, cb = ca + 1;
    const bool vb = cb < p.n_ch;
    const uint32_t sig_bytes = (uint32_t)(p.n_samples * 4);
    // a missing second channel is a buffer of no records: loads give 0, stores are dropped
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)ca * p.ldx), 0, (int)sig_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)(vb ? cb : ca) * p.ldx), 0, vb ? (int)sig_bytes : 0, 0x00020000);
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    float2* tw2p = lds + 16 * w4::L1S + 256;
    fill_tw2p(tw2p, p.twt, tid);

    // spectrum of the segment that ends with block b (samples [(b - 1) 2048, (b + 1) 2048)); a negative
    // byte offset wraps to a huge unsigned one: out of range, 0 -- the zeros in front of the signal
    auto forward = [&](float2 (&v)[16], int b) {
        const int off0 = 4 * ((b - 1) * HOP + tid);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1)
            v[n1] = make_float2(w4::ld_sample(ra, off0 + 1024 * n1), w4::ld_sample(rb, off0 + 1024 * n1));
        w4::fft4096_w(v, tw, buf, tw2, tid);
    };
    float2 xp[16];
    if (P == 2) forward(xp, b0 - 1);
    // the tap spectra through ONE raw-buffer descriptor with 32-bit byte offsets (64-bit addresses per
    // load are what hipcc spills first in these kernels)
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4*>(p.hp), 0, (int)((uint32_t)p.n_filt * P * (8 * 256 * 16)), 0x00020000);
    auto ld_h = [&](int byte_off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, byte_off, 0, 0));
    };
    const int hq = 16 * tid;
    for (int b = b0; b < b1; ++b) {
        float2 xc[16];
        forward(xc, b);
        const int out_off = 4 * (b * HOP + tid);
        float4 h0[8], h1[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            h0[g] = ld_h(hq + 4096 * g);
            if (P == 2) h1[g] = ld_h(hq + 4096 * (8 + g));
        }
        for (int f = 0; f < p.n_filt; ++f) {
            float2 v[16];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                v[2 * g] = cmul(xc[2 * g], make_float2(h0[g].x, h0[g].y));
                v[2 * g + 1] = cmul(xc[2 * g + 1], make_float2(h0[g].z, h0[g].w));
                if (P == 2) {
                    const float2 a = xp[2 * g], c = xp[2 * g + 1];
                    v[2 * g].x = fmaf(a.x, h1[g].x, fmaf(-a.y, h1[g].y, v[2 * g].x));
                    v[2 * g].y = fmaf(a.x, h1[g].y, fmaf(a.y, h1[g].x, v[2 * g].y));
                    v[2 * g + 1].x = fmaf(c.x, h1[g].z, fmaf(-c.y, h1[g].w, v[2 * g + 1].x));
                    v[2 * g + 1].y = fmaf(c.x, h1[g].w, fmaf(c.y, h1[g].z, v[2 * g + 1].y));
                }
            }
            // The next filter's tap spectra ride in the transform's call-outs (two 16-byte loads each; the
            // second partition's in the first pass, the first partition's in the second).  Behind the last
            // filter the same spectra are fetched again: a branch around loads in the middle of the
            // transform costs more than eight L2 hits.
            const int hn = hq + min(f + 1, p.n_filt - 1) * (P * 8 * 4096);
            ifft4096_wi(
                v, tw, buf, tw2p, tid,
                [&](int g) {
                    if (P == 2) {
                        h1[2 * g] = ld_h(hn + 4096 * (8 + 2 * g));
                        h1[2 * g + 1] = ld_h(hn + 4096 * (9 + 2 * g));
                    }
                },
                [&](int g) {
                    h0[2 * g] = ld_h(hn + 4096 * (2 * g));
                    h0[2 * g + 1] = ld_h(hn + 4096 * (2 * g + 1));
                });
            float* __restrict__ ya = p.y + ((int64_t)f * p.n_ch + ca) * p.ld_y;
            const __amdgpu_buffer_rsrc_t oa = __builtin_amdgcn_make_buffer_rsrc(ya, 0, (int)sig_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(ya + p.ld_y, 0, vb ? (int)sig_bytes : 0, 0x00020000);
            if constexpr (STAGE) {
                const int l = tid & 63, g = l >> 4, q4 = 4 * (l & 15);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    strip[m * 64 + l] = v[8 + m].x;
                    strip[(8 + m) * 64 + l] = v[8 + m].y;
                }
                w4::wave_sync();
                // lane (g, q): samples 4 q .. 4 q + 3 of this wave's 64 in the groups m = g and m = 4 + g
                const int so = 4 * (b * HOP + (tid & ~63) + q4);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int m = 4 * jj + g;
                    const float4 a4 = *reinterpret_cast<const float4*>(strip + m * 64 + q4);
                    const float4 b4 = *reinterpret_cast<const float4*>(strip + (8 + m) * 64 + q4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, a4), oa, so + 1024 * m, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, b4), ob, so + 1024 * m, 0, 0);
                }
                w4::wave_sync();  // the strip is rewritten by the next filter
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[8 + m].x), oa, out_off + 1024 * m, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[8 + m].y), ob, out_off + 1024 * m, 0, 0);
                }
            }
        }
        if (P == 2) {
#pragma unroll
            for (int s = 0; s < 16; ++s) xp[s] = xc[s];
        }
    }
}

// ---- two partitions at THREE workgroups per CU (round 5; default, DSPTOOLBOX_AMD_FIR_3PERCU=0 keeps k_fir<2>) -------------
// k_fir<2> holds the next filter's two tap spectra in flight through the whole inverse transform (64 registers: 214 in
// all, two workgroups per CU).  Here both are requested at the top of the filter's pass (PF = 0) -- the L2 round trip is
// covered by the two other workgroups of the CU --: 162 registers, no scratch, the same arithmetic, loads and stores
// per pass.  Measured on the bench shape (32 x 4097 taps, 8 x 2^22 samples; same box, alternating, three repetitions):
// 1.664-1.681 ms against 1.739-1.761 ms.  Keeping part of the prefetch does not pay: the first PF = 2 / 4 sixteen-byte
// values of the first partition still requested in the call-outs of the second transform pass (165 / 168 registers) gave
// 1.745-1.769 / 1.710-1.725 ms (profiles/r05_one_more_workgroup.txt); PF = 6 spills.
template <int PF>
__global__ __launch_bounds__(NT, 3) void k_fir3(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    const int tid = threadIdx.x;
    const int cp = (int)blockIdx.x / p.n_chunks, q = (int)blockIdx.x - cp * p.n_chunks;
    const int b0 = (int)((int64_t)q * p.n_blocks / p.n_chunks), b1 = (int)((int64_t)(q + 1) * p.n_blocks / p.n_chunks);
    const int ca = 2 * cp, cb = ca + 1;
    const bool vb = cb < p.n_ch;
    const uint32_t sig_bytes = (uint32_t)(p.n_samples * 4);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)ca * p.ldx), 0, (int)sig_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)(vb ? cb : ca) * p.ldx), 0, vb ? (int)sig_bytes : 0, 0x00020000);
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    float2* tw2p = lds + 16 * w4::L1S + 256;
    fill_tw2p(tw2p, p.twt, tid);
    auto forward = [&](float2 (&v)[16], int b) {
        const int off0 = 4 * ((b - 1) * HOP + tid);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1)
            v[n1] = make_float2(w4::ld_sample(ra, off0 + 1024 * n1), w4::ld_sample(rb, off0 + 1024 * n1));
        w4::fft4096_w(v, tw, buf, tw2, tid);
    };
    float2 xp[16];
    forward(xp, b0 - 1);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4*>(p.hp), 0, (int)((uint32_t)p.n_filt * 2 * (8 * 256 * 16)), 0x00020000);
    auto ld_h = [&](int byte_off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(hrs, byte_off, 0, 0));
    };
    const int hq = 16 * tid;
    for (int b = b0; b < b1; ++b) {
        float2 xc[16];
        forward(xc, b);
        const int out_off = 4 * (b * HOP + tid);
        float4 h0[8];
#pragma unroll
        for (int g = 0; g < PF; ++g) h0[g] = ld_h(hq + 4096 * g);
        for (int f = 0; f < p.n_filt; ++f) {
            const int hf = hq + f * (2 * 8 * 4096);
            float4 h1[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) h1[g] = ld_h(hf + 4096 * (8 + g));
#pragma unroll
            for (int g = PF; g < 8; ++g) h0[g] = ld_h(hf + 4096 * g);
            float2 v[16];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                v[2 * g] = cmul(xc[2 * g], make_float2(h0[g].x, h0[g].y));
                v[2 * g + 1] = cmul(xc[2 * g + 1], make_float2(h0[g].z, h0[g].w));
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float2 a = xp[2 * g], c = xp[2 * g + 1];
                v[2 * g].x = fmaf(a.x, h1[g].x, fmaf(-a.y, h1[g].y, v[2 * g].x));
                v[2 * g].y = fmaf(a.x, h1[g].y, fmaf(a.y, h1[g].x, v[2 * g].y));
                v[2 * g + 1].x = fmaf(c.x, h1[g].z, fmaf(-c.y, h1[g].w, v[2 * g + 1].x));
                v[2 * g + 1].y = fmaf(c.x, h1[g].w, fmaf(c.y, h1[g].z, v[2 * g + 1].y));
            }
            const int hn = hq + min(f + 1, p.n_filt - 1) * (2 * 8 * 4096);
            ifft4096_wi(
                v, tw, buf, tw2p, tid, [&](int) {},
                [&](int g) {
                    if (2 * g < PF) {
                        h0[2 * g] = ld_h(hn + 4096 * (2 * g));
                        h0[2 * g + 1] = ld_h(hn + 4096 * (2 * g + 1));
                    }
                });
            float* __restrict__ ya = p.y + ((int64_t)f * p.n_ch + ca) * p.ld_y;
            const __amdgpu_buffer_rsrc_t oa = __builtin_amdgcn_make_buffer_rsrc(ya, 0, (int)sig_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(ya + p.ld_y, 0, vb ? (int)sig_bytes : 0, 0x00020000);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[8 + m].x), oa, out_off + 1024 * m, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[8 + m].y), ob, out_off + 1024 * m, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) xp[s] = xc[s];
    }
}

// Byte offsets are formed in 32-bit signed arithmetic (4 * sample index) and a NEGATIVE offset is how the zeros
// in front of the signal are read (the range check sees a huge unsigned number): signals below 2^29 samples.
// (A wrapped negative register offset whose IMMEDIATE part carries it back over zero inside a wave was seen to
// return samples from in front of the row -- profiles/r04_fir_16s_blocks.txt; here hop and immediate offsets are
// multiples of 1024 samples, so a load's total offset is negative exactly when its register part is.)
inline bool fits(int64_t n_samples) { return n_samples > 0 && n_samples < ((int64_t)1 << 29) - 8192; }

}  // namespace fir4k
