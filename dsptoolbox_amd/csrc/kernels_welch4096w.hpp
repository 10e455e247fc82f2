// Welch nfft 4096, 50 % overlap, ONE input channel: the three-workgroups-per-CU form of the
// headline kernel (kernels_welch4096.hpp has the algebra: pair transforms, T += conj(W) Z,
// P += |Z|^2 over all 4096 bins, one fold k <-> N-k per chunk).  gfx950.
//
// What differs from welch4096::k_y (two workgroups per CU, 222 VGPRs, 72 KB of LDS):
//   * the second exchange is WAVE-LOCAL.  Pass 2 thread (k1u, n3) holds k2 = 0..15; pass 3 wants
//     thread (k1u, k2) with n3 = 0..15: a 16 x 16 transpose inside the 16 lanes that share k1u.
//     Those lanes are the only readers of row k1u of the pass-1 image, so the row itself
//     (272 complex = 16 x 17) is their transpose area: no second buffer, no workgroup barrier,
//     LDS traffic of one wave executes in program order.
//   * ONE exchange buffer (34 KB) + the window in LDS (16 KB) + W256 table = 52 KB -> three
//     workgroups per CU; two workgroup barriers per transform, back to back around the pass-1
//     image stores, so the waves of a workgroup meet once per transform.
//   * <= 168 VGPRs: window values come from LDS, the W4096 twiddles are rebuilt from six base
//     values (W^t, W^2t, W^3t, W^4t, W^8t, W^12t: one extra complex product for nine of the
//     fifteen), and with hop = N/2 the last half block of a pair is the first of the next, so
//     only 16 new samples per thread and pair are loaded (8 carried in registers).
//   * thread tid ends with bins bt + 256 k3, bt = 16 (tid & 15) + (tid >> 4); k_x3 stores the
//     input spectra thread-major in exactly that mapping.
#pragma once
#include <utility>

#include "kernels_welch4096.hpp"

namespace welch4096 {

// dev only (-DW4_TIMING=1): per-phase s_memtime stamps of workgroup 0 / lane 0 -> w3_timing[]
#if W4_TIMING
__device__ unsigned long long w3_timing[16];
__device__ unsigned long long w3_life[4096][8];  // per workgroup: memtime start/end, memrealtime start/end, loop start/end, HW_ID, XCC_ID
struct Stamp {
    unsigned long long ph[12] = {}, prev = 0;
    __device__ __forceinline__ void operator()(int i) {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (i > 0) ph[i - 1] += t - prev;
        prev = t;
    }
};
#else
struct Stamp {
    __device__ __forceinline__ void operator()(int) {}
};
#endif

constexpr int L3S = 17;                                       // transposed row stride (complex)
constexpr int LDS3_BYTES = 16 * L1S * 8 + 256 * 8 + N * 4;    // exchange + W256 + window = 53248

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef W4_OCC
#define W4_OCC 3  // workgroups per CU the kernel is built for (2: twiddles and window in registers)
#endif
#ifndef W4_WIN_ROT
#define W4_WIN_ROT 1
#endif
#ifndef W4_TW6
#define W4_TW6 (W4_OCC == 3)
#endif
#ifndef W4_NO_READ2
#define W4_NO_READ2 1
#endif
#ifndef W4_TW2_REG
#define W4_TW2_REG 0  // experiment (round 5): the W256 twiddles of pass 2 rebuilt from six per-thread values instead of 15 LDS reads
#endif
#ifndef W4_WIN_GLOBAL
#define W4_WIN_GLOBAL 0  // experiment (tools/exp): window values by buffer loads (L1-resident 16 KB) instead of the LDS copy
#endif
constexpr int LDS3G_BYTES = 16 * L1S * 8 + 256 * 8;  // ... which leaves exchange + W256 = 36864 bytes: four workgroups per CU

struct Tw6 {
#if W4_TW6
    float2 a[3];  // W4096^(t k1), k1 = 1, 2, 3
    float2 b[3];  // W4096^(t k1), k1 = 4, 8, 12
#else
    float2 w[15];
#endif
#if W4_TW2_REG
    float2 c[3];  // W256^(n3 k2), k2 = 1, 2, 3
    float2 d[3];  // W256^(n3 k2), k2 = 4, 8, 12
#endif
};

__device__ __forceinline__ void load_tw6(Tw6& tw, const float2* __restrict__ twt, int tid) {
#if W4_TW6
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        tw.a[j] = twt[j * 256 + tid];
        tw.b[j] = twt[(4 * (j + 1) - 1) * 256 + tid];
    }
#else
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) tw.w[k1 - 1] = twt[(k1 - 1) * 256 + tid];
#endif
#if W4_TW2_REG
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        tw.c[j] = twt[15 * 256 + (j + 1) * 16 + (tid & 15)];        // tw2[k2 * 16 + n3]
        tw.d[j] = twt[15 * 256 + 4 * (j + 1) * 16 + (tid & 15)];
    }
#endif
}

__device__ __forceinline__ void apply_tw6(float2 (&v)[16], const Tw6& tw) {
#if W4_TW6
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const int lo = k1 & 3, hi = k1 >> 2;
        float2 z = v[pos16(k1)];
        if (lo) z = cmul(z, tw.a[lo - 1]);
        if (hi) z = cmul(z, tw.b[hi - 1]);
        v[pos16(k1)] = z;
    }
#else
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) v[pos16(k1)] = cmul(v[pos16(k1)], tw.w[k1 - 1]);
#endif
}

// One channel as a raw buffer: the hardware range check returns 0 for every sample at or past
// n_samples, which IS the reference's zero padding of the last frames (helpers/other.py:207-209)
// -- no ragged-tail code path.  The whole offset travels in the VGPR/immediate part (the part
// the range check sees); n_samples < 2^30 - 2^13 (the host routes longer signals elsewhere).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t channel_rsrc(const float* ch, int64_t n_samples) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ch), 0, (int)(uint32_t)(n_samples * 4), 0x00020000);
}
__device__ __forceinline__ float ld_sample(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// 8-byte LDS reads that stay ds_read_b64: hipcc pairs neighbouring reads into ds_read2_b64, which
// occupies the LDS array for 8 cycles where two ds_read_b64 take 2 + 2 (MI355X LDS table; rocprof
// SQ_LDS_IDX_ACTIVE of the paired build agreed with that accounting to 4 %).  Inline asm, so the
// compiler does not see the pending result: every use is behind an explicit counted
// s_waitcnt lgkmcnt (LDS operations of a wave return in order) followed by a scheduling barrier.
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }
template <int BYTE_OFF>
__device__ __forceinline__ void lds_rd64(float2& d, uint32_t addr) {
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 65536 && BYTE_OFF % 8 == 0, "ds_read_b64 offset");
    v2f_t r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(BYTE_OFF) : "memory");
    d = make_float2(r.x, r.y);
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// bins held by thread tid after fft4096_w: bt + 256 k3 (v[pos16(k3)])
__device__ __forceinline__ int bin_thread(int tid) { return ((tid & 15) << 4) | (tid >> 4); }
// padded position of bin k in the fold image (stride-16 bins of neighbouring lanes -> 17)
__device__ __forceinline__ int fold_pos(int k) { return k + (k >> 4); }

// v[n1] = z[tid + 256 n1]  ->  Z[bt + 256 k3] in v[pos16(k3)].  `buf`: the workgroup's ONE exchange
// image (16 rows of L1S complex).  Two workgroup barriers, both around the pass-1 stores.
template <typename Hook = NoHook, typename Hook2 = NoHook>
__device__ __forceinline__ void fft4096_w(float2 (&v)[16], const Tw6& tw, float2* __restrict__ buf,
                                          const float2* __restrict__ tw2, int tid,
                                          Hook behind_ex1 = Hook(), Hook2 behind_ex2 = Hook2()) {
    dft16(v);
    apply_tw6(v, tw);
    const int k1u = tid >> 4, n3 = tid & 15;
    __syncthreads();  // every wave has read its rows (pass 3 of the previous transform)
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) buf[k1 * L1S + tid] = v[pos16(k1)];
    float2 w2[15];
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) w2[k2 - 1] = tw2[k2 * 16 + n3];
    behind_ex1();
    __syncthreads();
    float2* __restrict__ row = buf + k1u * L1S;
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = row[16 * n2 + n3];
    dft16(v);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) v[pos16(k2)] = cmul(v[pos16(k2)], w2[k2 - 1]);
    // 16 x 16 transpose (n3 <-> k2) among the 16 lanes of this row, in the row itself
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) row[n3 * L3S + k2] = v[pos16(k2)];
    behind_ex2();
    wave_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = row[j * L3S + n3];  // lane now plays k2 = n3
    dft16(v);
}

// Progress-based wave priority.  All workgroups of the grid are resident at once and run the same
// loop; the CU arbitrates VALU issue by priority, then AGE, so at equal priority the oldest of
// the three workgroups of a CU wins every conflict, finishes its chunk at 57 us and leaves the CU
// under-occupied while the youngest needs 92 us (per-workgroup s_memrealtime stamps).  Here a
// workgroup's priority falls as its own work gets done (16 levels: 4 hardware levels, dithered
// over consecutive iterations), so whoever is behind wins the arbitration and the three
// finish together.  One update per iteration (the immediate operand costs a scalar branch chain).
#ifndef W4_PRIO
#define W4_PRIO 1
#endif
#ifndef W4_XS_EARLY
#define W4_XS_EARLY 0
#endif
#ifndef W4_AB
#define W4_AB 0  // timing-only ablations (wrong results): 1 no xs loads, 2 no sample loads, 4 no LDS traffic, 8 no barriers,
                 // 16 no internal W16 twiddles, 32 no window (no LDS window reads), 64 single twiddle products (no rebuild)
#endif
#define W4_SYNC()                          \
    do {                                   \
        if (!(W4_AB & 8)) __syncthreads(); \
    } while (0)
__device__ __forceinline__ void set_prio(int level16, int dither) {
#if W4_PRIO
    const int pr = min(3, (level16 + dither) >> 2);
    if (pr == 0)
        __builtin_amdgcn_s_setprio(0);
    else if (pr == 1)
        __builtin_amdgcn_s_setprio(1);
    else if (pr == 2)
        __builtin_amdgcn_s_setprio(2);
    else
        __builtin_amdgcn_s_setprio(3);
#endif
}

// fft4096_w with the exchange traffic spread between the butterflies.
//   ld_a(g), g = 0..3: call-outs of pass 1's stage A (the caller's sample loads go there: the
//                      registers of the previous samples have just been consumed by the window,
//                      and a load issued here has a whole iteration to arrive -- vmcnt retires in
//                      order, so the wait for the input spectrum at the accumulation also waits
//                      for every older sample load)
//   ld_b(g), g = 0..3: call-outs of pass 2's stage B, next to the transpose stores (input spectrum)
// The barrier that protects the pass-1 image of the NEXT transform sits right behind this
// transform's last LDS read, so the next transform's pass-1 stores can start while its own
// butterflies are still running.
//   mid():            called by every wave right behind the mid-transform barrier (before the first
//                      ld_b): the fused kernel waits there, once, for its chunk's input spectra
template <typename LA, typename LB, typename MID = NoHook>
__device__ __forceinline__ void fft4096_wi(float2 (&v)[16], const Tw6& tw, float2* __restrict__ buf,
                                           const float2* __restrict__ tw2, int tid, LA ld_a, LB ld_b,
                                           Stamp& ts, int level16, MID mid = MID()) {
    const int k1u = tid >> 4, n3 = tid & 15;
    float2* __restrict__ col = buf + tid;
    dft16_h(
        v, NoHookI(),
        [&](int g) {
            W4_PIN();
            ld_a(g);
            W4_PIN();
        },
        [&](int g) {
        // X[k1 = g + 4 j] -> twiddle -> pass-1 image row k1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float2 z = v[4 * g + j];
#if W4_TW6
            if (g) z = cmul(z, tw.a[g - 1]);
            if (j && !((W4_AB & 64) && g)) z = cmul(z, tw.b[j - 1]);
#else
            if (g + 4 * j) z = cmul(z, tw.w[g + 4 * j - 1]);
#endif
            W4_PIN();
            if (W4_AB & 4)
                asm volatile("" ::"v"(z.x), "v"(z.y));
            else
                col[(g + 4 * j) * L1S] = z;
            W4_PIN();
        }
    });
    ts(2);
    W4_SYNC();
    mid();
    ts(3);
    float2* __restrict__ row = buf + k1u * L1S;
    const uint32_t a_row = lds_addr(row + n3), a_tw2 = lds_addr(tw2 + n3);
    // pass-2 inputs v[n2] = row[16 n2 + n3], requested in the order the butterflies consume them
    W4_PIN();
    static_for<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value, n2 = 4 * (i & 3) + (i >> 2);
        if (W4_AB & 4)
            asm volatile("" : "=v"(v[n2].x), "=v"(v[n2].y));
        else
            lds_rd64<16 * n2 * 8>(v[n2], a_row);
    });
    ts(4);
    float2 w2[16];
    dft16_h(
        v,
        [&](auto gc) {
            // 16 reads issued, then 3 + 4 + 4 table reads behind the first three butterflies:
            // butterfly g needs the first 4 (g + 1) of the 16
            constexpr int g = decltype(gc)::value;
#if W4_TW2_REG
            lgkm_wait<12 - 4 * g>();
#else
            lgkm_wait<(g == 0 ? 12 : 11)>();
#endif
        },
        [&](int g) {
            W4_PIN();
#if !W4_TW2_REG
            static_for<4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                // (the compiler folds g; the offset must be a constant for the asm operand)
                if (g == 0 && j > 0) lds_rd64<(0 + j) * 16 * 8>(w2[0 + j], a_tw2);
                if (g == 1) lds_rd64<(4 + j) * 16 * 8>(w2[4 + j], a_tw2);
                if (g == 2) lds_rd64<(8 + j) * 16 * 8>(w2[8 + j], a_tw2);
                if (g == 3) lds_rd64<(12 + j) * 16 * 8>(w2[12 + j], a_tw2);
            });
#endif
#if W4_XS_EARLY
            ld_b(g);
#endif
            if (g == 3) {
                lgkm_wait<0>();  // every table value is in before stage B multiplies
                ts(5);
            }
            W4_PIN();
        },
        [&](int g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k2 = g + 4 * j;
                float2 z = v[4 * g + j];
#if W4_TW2_REG
                if (g) z = cmul(z, tw.c[g - 1]);
                if (j) z = cmul(z, tw.d[j - 1]);
#else
                if (k2) z = cmul(z, w2[k2]);
#endif
                W4_PIN();
                if (W4_AB & 4)
                    asm volatile("" ::"v"(z.x), "v"(z.y));
                else
                    row[n3 * L3S + k2] = z;
                W4_PIN();
            }
#if !W4_XS_EARLY
            W4_PIN();
            ld_b(g);
            W4_PIN();
#endif
        });
    ts(6);
    wave_sync();
    W4_PIN();
    static_for<16>([&](auto ic) {  // lane now plays k2 = n3
        constexpr int i = decltype(ic)::value, j = 4 * (i & 3) + (i >> 2);
        if (W4_AB & 4)
            asm volatile("" : "=v"(v[j].x), "=v"(v[j].y));
        else
            lds_rd64<j * L3S * 8>(v[j], a_row);
    });
    ts(7);
    dft16_h(
        v,
        [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            lgkm_wait<12 - 4 * g>();
            // all reads of the image are back: the next transform's pass-1 stores may begin
            if (g == 3) {
                ts(8);
                W4_SYNC();
                ts(9);
            }
        },
        NoHookI(), NoHookI());
}

// ---- input spectra -----------------------------------------------------------
__global__ __launch_bounds__(NT) void k_x3(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * L1S;
    const int tid = threadIdx.x;
    const int cx = (int)blockIdx.x / p.n_pairs, pr = (int)blockIdx.x - cx * p.n_pairs;  // one input channel: cx = 0
    Tw6 tw;
    float2 v[16];
    {
        Raw<true> raw;
        const __amdgpu_buffer_rsrc_t rs = channel_rsrc(p.sig + (int64_t)cx * p.ld, p.n_samples);
        const int off0 = 4 * (2 * pr * 2048 + tid);
#pragma unroll
        for (int m = 0; m < 24; ++m) raw.s[m] = ld_sample(rs, off0 + 1024 * m);
        float win[16];
        load_tw6(tw, p.twt, tid);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) win[n1] = p.window[tid + 256 * n1];
        tw2[tid] = p.twt[15 * 256 + tid];
        window_pair<true>(v, raw, win);
        if (needs_drop(p, pr)) drop_second(v);
    }
    fft4096_w(v, tw, buf, tw2, tid);
    if (p.detrend && tid == 0) v[pos16(0)] = make_float2(0.f, 0.f);
    float4* xo = reinterpret_cast<float4*>(p.xs + ((int64_t)cx * p.n_pairs + pr) * N) + tid;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float2 z0 = v[pos16(2 * g)], z1 = v[pos16(2 * g + 1)];
        xo[256 * g] = make_float4(z0.x, z0.y, z1.x, z1.y);
    }
    float* pw = reinterpret_cast<float*>(buf);
    const int bt = bin_thread(tid);
    __syncthreads();
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
        float2 z = v[pos16(k3)];
        pw[fold_pos(bt + 256 * k3)] = z.x * z.x + z.y * z.y;
    }
    __syncthreads();
    float* po = p.px + ((int64_t)cx * p.n_pairs + pr) * NB;
    for (int k = tid; k < NB; k += NT) po[k] = 0.5f * (pw[fold_pos(k)] + pw[fold_pos((N - k) & (N - 1))]);
}

// pairs [p0, p1) of chunk q: n_pairs / n_chunks each, one more where the host set the chunk's bit
// (place_remainder), else the even split
__device__ __forceinline__ void chunk_range(const Args& p, int q, int& p0, int& p1) {
    if (p.use_plus) {
        auto below = [&](int qq) {  // number of set bits of p.plus below bit qq
            int n = 0;
            for (int w = 0; w < (qq >> 5); ++w) n += __popc(p.plus[w]);
            if (qq & 31) n += __popc(p.plus[qq >> 5] & ((1u << (qq & 31)) - 1u));
            return n;
        };
        const int base = p.n_pairs / p.n_chunks;
        p0 = q * base + below(q);
        p1 = p0 + base + (int)((p.plus[q >> 5] >> (q & 31)) & 1u);
    } else {
        p0 = (int)((int64_t)q * p.n_pairs / p.n_chunks);
        p1 = (int)((int64_t)(q + 1) * p.n_pairs / p.n_chunks);
    }
}

// paired inputs: px rows of every input channel summed over the pairs of each chunk (fp64);
// grid = (n_chunks, n_cx).  (With ONE input channel k_y3's workgroups share this sum instead.)
__global__ __launch_bounds__(256) void k_px_sum(Args p) {
    const int q = blockIdx.x, cx = blockIdx.y;
    int p0, p1;
    chunk_range(p, q, p0, p1);
    for (int k = threadIdx.x; k < NB; k += 256) {
        double sum = 0.0;
        for (int pr = p0; pr < p1; ++pr) sum += (double)p.px[((int64_t)cx * p.n_pairs + pr) * NB + k];
        p.psx[((int64_t)q * p.n_cx + cx) * NB + k] = (float)sum;
    }
}

// ---- output channels ---------------------------------------------------------
template <bool AUTO = false>
__global__ __launch_bounds__(NT, W4_OCC) void k_y3(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * L1S;
    float* winl = reinterpret_cast<float*>(lds + 16 * L1S + 256);
    const int tid = threadIdx.x;
    // XCD-aware decode: blocks b, b + 8, ... share an XCD (and its L2).  Each XCD takes a contiguous
    // run of (chunk, channel) units in chunk-major order, so the input spectra it re-reads for every
    // channel are those of one or two chunks and stay in its 4 MB L2 (grid = a multiple of 8).
    int q, c;
    {
        const int b = blockIdx.x, total = p.n_chunks * p.n_ch;
        const int u = (total & 7) == 0 ? (b & 7) * (total >> 3) + (b >> 3) : b;
        q = u / p.n_ch;
        c = u - q * p.n_ch;
    }
#if W4_TIMING
    const unsigned long long life_t0 = __builtin_amdgcn_s_memtime(), life_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    Tw6 tw;
    load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
#if W4_WIN_GLOBAL
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.window), 0, N * 4, 0x00020000);
    (void)winl;
#elif W4_OCC == 3
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) winl[tid + 256 * n1] = p.window[tid + 256 * n1];
#else
    float winr[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) winr[n1] = p.window[tid + 256 * n1];
#endif
    const float* ch = p.sig + (int64_t)c * p.ld;
    int p0, p1;
    chunk_range(p, q, p0, p1);
    const int cx = p.n_cx > 1 ? c : 0;  // paired inputs: this channel's own input spectra
    if (!AUTO && p.n_cx <= 1) {
        // Input auto-spectrum of this chunk: every workgroup of the chunk sums a slice of the bins
        // over the chunk's px rows (fp64).  The slice (33 bins for 64 channels) is covered in ONE
        // sweep -- `width` bins x 256 / width row groups -- so the loads of a workgroup are one
        // round trip, not two.
        double* red = reinterpret_cast<double*>(lds);  // [256 / width][width], before the first transform
        const int bpc = (NB + p.n_ch - 1) / p.n_ch;
        const int b0 = c * bpc, b1 = min(b0 + bpc, NB);
        const int lw = bpc <= 32 ? 5 : (bpc <= 64 ? 6 : (bpc <= 128 ? 7 : 8)), width = 1 << lw, rows = NT >> lw;
        const int rg = tid >> lw, kl = tid & (width - 1);
        for (int kb = b0; kb < b1; kb += width) {
            const int k = kb + kl;
            double sum = 0.0;
            if (k < b1)
                for (int pr = p0 + rg; pr < p1; pr += rows) sum += (double)p.px[(int64_t)pr * NB + k];
            red[rg * width + kl] = sum;
            __syncthreads();
            if (rg == 0 && k < b1) {
                double t = 0.0;
                for (int j = 0; j < rows; ++j) t += red[j * width + kl];
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
    // samples of the current pair: half block 0 (carried from the previous pair) and half
    // blocks 1, 2 (s[m] = ch[start + tid + 256 m], m = 8..23)
    float carry[8], nx[16];
    const __amdgpu_buffer_rsrc_t rs = channel_rsrc(ch, p.n_samples);
    // the chunk's input spectra as a raw buffer (32-bit offsets, one address register)
    const __amdgpu_buffer_rsrc_t xrs =
        AUTO ? rs
             : __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.xs + ((int64_t)cx * p.n_pairs + p0) * N), 0,
                                                 (int)(uint32_t)((p1 - p0) * (N * 8)), 0x00020000);
    if (p0 < p1) {
        const int off0 = 4 * (2 * p0 * 2048 + tid);
#pragma unroll
        for (int j = 0; j < 8; ++j) carry[j] = ld_sample(rs, off0 + 1024 * j);
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = ld_sample(rs, off0 + 1024 * (8 + j));
    }
    Stamp ts;
#if W4_TIMING
    const unsigned long long life_r1 = __builtin_amdgcn_s_memrealtime();
#endif
#if W4_WIN_GLOBAL
    float winr[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) winr[n1] = ld_sample(wrs, 4 * (tid + 256 * n1));
#elif W4_OCC == 3 && W4_WIN_ROT
    // the window values of an iteration are requested from LDS at the end of the previous one (into
    // registers the accumulation has just freed) instead of in front of their first use
    float winr[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) winr[n1] = winl[tid + 256 * n1];
#endif
    for (int pr = p0; pr < p1; ++pr) {
        float2 v[16];
        ts(0);
        const int level16 = ((p1 - pr - 1) * 16) / (p1 - p0);  // 15 ... 0 as the chunk gets done
        set_prio(level16, (pr * 5) & 3);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
#if W4_OCC == 3 && !W4_WIN_ROT && !W4_WIN_GLOBAL
            const float w = winl[tid + 256 * n1];
#else
            const float w = winr[n1];
#endif
            const float a = n1 < 8 ? carry[n1] : nx[n1 - 8];
            v[n1] = (W4_AB & 32) ? make_float2(a, nx[n1]) : make_float2(a * w, nx[n1] * w);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) carry[j] = nx[8 + j];
        if (needs_drop(p, pr)) drop_second(v);
        ts(1);
        float2 xw[16];
        const int off1 = 4 * ((2 * pr + 2) * 2048 + tid) + 1024 * 8;
        const int xoff = (pr - p0) * (N * 8) + tid * 16;  // bytes into this chunk's input spectra
        fft4096_wi(
            v, tw, buf, tw2, tid,
            [&](int g) {  // samples of the next pair, four per call-out (past the chunk's last pair
                          // they are simply not used; past the signal the range check gives 0)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (W4_AB & 2)
                        asm volatile("" : "=v"(nx[4 * g + j]));
                    else
                        nx[4 * g + j] = ld_sample(rs, off1 + 1024 * (4 * g + j));
                }
            },
            [&](int g) {  // input spectrum of this pair, two 16-byte loads per call-out
#if W4_WIN_GLOBAL
                // the next pair's window values (their registers are free since the windowing above)
#pragma unroll
                for (int j = 0; j < 4; ++j) winr[4 * g + j] = ld_sample(wrs, 4 * (tid + 256 * (4 * g + j)));
#endif
                if (!AUTO) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float4 q4;
                        if (W4_AB & 1)
                            asm volatile("" : "=v"(q4.x), "=v"(q4.y), "=v"(q4.z), "=v"(q4.w));
                        else
                            q4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff + 4096 * (2 * g + j), 0, 0));
                        xw[2 * (2 * g + j)] = make_float2(q4.x, q4.y);
                        xw[2 * (2 * g + j) + 1] = make_float2(q4.z, q4.w);
                    }
                }
            },
            ts, level16);
        ts(10);
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) {
            float2 z = v[pos16(k3)];
            if (!AUTO) {
                float2 w = xw[k3];
                T[k3].x = fmaf(w.x, z.x, fmaf(w.y, z.y, T[k3].x));
                T[k3].y = fmaf(w.x, z.y, fmaf(-w.y, z.x, T[k3].y));
            }
            P[k3] = fmaf(z.x, z.x, fmaf(z.y, z.y, P[k3]));
        }
#if W4_OCC == 3 && W4_WIN_ROT && !W4_WIN_GLOBAL
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1)
            if (!(W4_AB & 32)) winr[n1] = winl[tid + 256 * n1];
        __builtin_amdgcn_sched_barrier(0);
#endif
        ts(11);
    }
#if W4_TIMING
    const unsigned long long life_r2 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 11; ++i) atomicAdd(&w3_timing[i], ts.ph[i]);
        atomicAdd(&w3_timing[15], (unsigned long long)(p1 - p0));
    }
#endif
    if (p.detrend && tid == 0) P[0] = 0.f;
    // fold k <-> N-k once per chunk, through LDS (padded image: 4096 + 256 = 16 x 272)
    const int bt = bin_thread(tid);
    __syncthreads();
    const int64_t so = ((int64_t)q * p.n_ch + c) * NB;
    if (!AUTO) {
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) buf[fold_pos(bt + 256 * k3)] = T[k3];
        __syncthreads();
        for (int k = tid; k < NB; k += NT) {
            float2 a = buf[fold_pos(k)], b = buf[fold_pos((N - k) & (N - 1))];
            p.pxy[so + k] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
        }
        __syncthreads();
    }
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) pw[fold_pos(bt + 256 * k3)] = P[k3];
    __syncthreads();
    for (int k = tid; k < NB; k += NT) p.pyy[so + k] = 0.5f * (pw[fold_pos(k)] + pw[fold_pos((N - k) & (N - 1))]);
#if W4_TIMING
    if (tid == 0 && blockIdx.x < 4096) {
        w3_life[blockIdx.x][0] = life_t0;
        w3_life[blockIdx.x][1] = __builtin_amdgcn_s_memtime();
        w3_life[blockIdx.x][2] = life_r0;
        w3_life[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
        w3_life[blockIdx.x][4] = life_r1;
        w3_life[blockIdx.x][5] = life_r2;
        w3_life[blockIdx.x][6] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        w3_life[blockIdx.x][7] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
#endif
}

// ---- host side -----------------------------------------------------------------
// three workgroups per CU, all resident at once: 768 (chunk, channel) units when there is enough
// work; fp32 accumulation chains stay <= 64 pairs; grid a multiple of 8 where possible so the
// XCD-aware decode of k_y3 applies
inline int chunks_for3(int n_pairs, int n_ch, int want = 0) {  // want: the caller's chunk count, 0 = choose here
    if (want <= 0) {
        want = (768 + n_ch - 1) / n_ch;
        const int by_len = (n_pairs + 63) / 64;
        if (want < by_len) want = by_len;
        // prefer a grid that is a multiple of 8 (whole units per XCD)
        for (int w = want; w < want + 8; ++w)
            if (((int64_t)w * n_ch) % 8 == 0) {
                want = w;
                break;
            }
    }
    if (want > n_pairs) want = n_pairs;
    if (want < 1) want = 1;
    return want;
}
inline Plan plan3(int n_frames, int n_cy, int want_chunks = 0) {
    Plan pl;
    pl.n_pairs = (n_frames + 1) / 2;
    pl.n_chunks = chunks_for3(pl.n_pairs, n_cy, want_chunks);
    pl.ppc = (pl.n_pairs + pl.n_chunks - 1) / pl.n_chunks;
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    pl.bytes = pad(sizeof(float2) * (size_t)pl.n_pairs * N) + pad(sizeof(float) * (size_t)pl.n_pairs * NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * NB) + pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * NB) +
               pad(sizeof(float) * (size_t)pl.n_chunks * n_cy * NB);
    return pl;
}
// Where the n_pairs % n_chunks longer chunks go.  k_y3 hands XCD x the units [x U, (x + 1) U) of
// the chunk-major (chunk, channel) list, U = n_chunks n_ch / 8; with 12 chunks of 64 channels an
// XCD owns one whole chunk and half of a shared one, and an even split (21, 21, 22, 21, ...) gives
// the XCDs 2016 or 2080 transforms (3 % apart, seen as 89 vs 93 us mean workgroup lifetime).
// Greedy: each extra pair goes to the chunk that keeps the busiest XCD lowest.
inline void place_remainder(Args& a, int n_ch) {
    a.use_plus = 0;
    for (auto& w : a.plus) w = 0;
    const int nc = a.n_chunks, total = nc * n_ch;
    if (nc > 24 * 32 || (total & 7) != 0 || nc <= 0) return;
    const int base = a.n_pairs / nc, rem = a.n_pairs - base * nc;
    a.use_plus = 1;
    if (rem == 0) return;
    const int U = total / 8;
    std::vector<int> ov((size_t)8 * nc, 0);  // units of chunk q on XCD x
    for (int u = 0; u < total; ++u) ov[(size_t)(u / U) * nc + u / n_ch] += 1;
    std::vector<int64_t> load(8, 0);
    for (int x = 0; x < 8; ++x)
        for (int q = 0; q < nc; ++q) load[x] += (int64_t)ov[(size_t)x * nc + q] * base;
    std::vector<char> taken(nc, 0);
    for (int r = 0; r < rem; ++r) {
        int best = -1;
        int64_t best_max = 0, best_aff = 0;
        for (int q = 0; q < nc; ++q) {
            if (taken[q]) continue;
            int64_t mx = 0, aff = 0;
            for (int x = 0; x < 8; ++x) {
                const int64_t l = load[x] + ov[(size_t)x * nc + q];
                mx = std::max(mx, l);
                if (ov[(size_t)x * nc + q]) aff = std::max(aff, l);
            }
            if (best < 0 || mx < best_max || (mx == best_max && aff < best_aff)) {
                best = q;
                best_max = mx;
                best_aff = aff;
            }
        }
        taken[best] = 1;
        a.plus[best >> 5] |= 1u << (best & 31);
        for (int x = 0; x < 8; ++x) load[x] += ov[(size_t)x * nc + best];
    }
}

// the raw-buffer loads carry byte offsets in 32 bits
inline bool fits3(int64_t n_samples, int n_frames) {
    return n_samples < ((int64_t)1 << 30) - 8192 && (int64_t)(n_frames + 2) * 2048 < ((int64_t)1 << 30) - 8192;
}

}  // namespace welch4096
