// Generic power-of-two complex FFT held in LDS by one team of threads (gfx950).
//
// Stockham autosort with radix-16 passes (then one radix-4 and/or radix-2 pass for the
// leftover bits), in place in ONE LDS buffer of N float2: every thread first reads all the
// butterfly inputs it owns into registers, the team barriers, then the outputs go back to
// the (different) autosort positions.  N = 16384 takes 4 passes (16,16,16,4) instead of
// the 7 of a radix-4 scheme: 4/7 of the LDS traffic and barriers.
//
// Register interfaces.  A pass of radix R owns, per thread, the butterflies
// j = tid + i*NT (j < N/R) with inputs x[j + t*N/R], t < R, kept in v[i*R + t]:
//   * FIRST_FROM_REG: the caller has loaded those values (coalesced over j) for the first
//     pass -- radix first_radix<N, REVERSED>();
//   * LAST_TO_REG: the last pass (radix last_radix<N, REVERSED>(), sub-transform length
//     N/R) leaves X[j + t*N/R] in v[i*R + t].
// REVERSED runs the radix sequence backwards (4,16,16,16): the output layout of a forward
// transform (last radix 4) is then the input layout of the following reversed transform,
// which lets the FIR kernel go forward -> multiply -> inverse without touching LDS.
//
// LDS addressing: logical index i lives at lidx(i) = i + (i >> 4).  Without the extra slot per
// 16 elements the first pass (sub-transform length 1) scatters its outputs with a stride of
// 16 complex = 32 banks: every lane of a wave on the same two banks (rocprof: 67 % of the LDS
// cycles of the 1024-point STFT kernel were bank conflicts).  Every access to a transform
// buffer -- inside the passes and by the kernels that read the natural-order result -- goes
// through lidx(); a buffer holds lds_len<N>() float2.
//
// Twiddles: one table per pass of exp(-2 pi i t k / (NS R)) (t < R, k < NS), stored as columns
// [t][k] (for one t the lanes read consecutive entries) while the table fits the 32 KB vector L1,
// and as rows [k][t] for the big radix-16 tables (a thread's 15 twiddles = one 128-byte row, a
// wave = 64 consecutive rows, each line fetched once): no scattered gathers either way.
// Measured both ways: all-columns FIR 4.0 ms / deconv 58 us, all-rows 3.2 ms / 70 us.  tw_offset<N, REVERSED>(pass) locates a pass's table inside the
// per-length blob built in fp64 on the host (tw_table_len<N>() entries, forward sequence
// first, then the reversed one), cached in the context.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace dsfft {

__host__ __device__ constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
__host__ __device__ constexpr int lidx(int i) { return i + (i >> 4); }
template <int N>
__host__ __device__ constexpr int lds_len() { return N + N / 16 + 2; }
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }
__host__ __device__ constexpr int cmin(int a, int b) { return a < b ? a : b; }

// threads cooperating on one length-N complex FFT
__host__ __device__ constexpr int threads_for(int n) { return cmax(64, cmin(512, n / 16)); }

// radix sequence: as many 16s as fit, then 4 and/or 2
template <int N>
struct Plan {
    static constexpr int LOGN = ilog2(N);
    static constexpr int N16 = LOGN / 4;
    static constexpr int REM = LOGN % 4;  // 0, 1 (->2), 2 (->4), 3 (->4,2)
    static constexpr int NPASS = N16 + (REM == 0 ? 0 : (REM == 3 ? 2 : 1));
    static constexpr int NT = threads_for(N);
    __host__ __device__ static constexpr int radix(int p) {  // forward order
        return p < N16 ? 16 : ((REM == 1) ? 2 : ((REM == 2) ? 4 : (p == N16 ? 4 : 2)));
    }
    __host__ __device__ static constexpr int bpt(int r) { return (N / r + NT - 1) / NT; }
    // registers per thread: the widest pass
    static constexpr int VMAX = cmax(cmax(N16 > 0 ? bpt(16) * 16 : 0, (REM >= 2) ? bpt(4) * 4 : 0),
                                     (REM == 1 || REM == 3) ? bpt(2) * 2 : 0);
    static constexpr int LDS_BYTES = lds_len<N>() * 8;
};

template <int N, bool REVERSED>
__host__ __device__ constexpr int first_radix() {
    return REVERSED ? Plan<N>::radix(Plan<N>::NPASS - 1) : Plan<N>::radix(0);
}
template <int N, bool REVERSED>
__host__ __device__ constexpr int last_radix() {
    return REVERSED ? Plan<N>::radix(0) : Plan<N>::radix(Plan<N>::NPASS - 1);
}

// Backwards-compatible names used by the kernels
template <int N>
struct Cfg {
    static constexpr int NT = Plan<N>::NT;
    static constexpr int LDS_BYTES = Plan<N>::LDS_BYTES;
    static constexpr int VMAX = Plan<N>::VMAX;
    static constexpr int R1 = first_radix<N, false>();      // radix of the register-fed first pass
    static constexpr int BPT1 = Plan<N>::bpt(R1);           // butterflies per thread in it
    static constexpr int NB1 = N / R1;                      // butterflies in it (= input stride)
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {  // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// Barrier between the threads of ONE transform.  A transform of <= 1024 points is owned by a
// single wave (NT == 64): its LDS operations execute in program order, so no s_barrier is needed
// (6 workgroup barriers per transform removed from the 1024-point STFT).  Kernels that let OTHER
// teams touch a transform's buffer (cooperative loads / stores) place their own __syncthreads().
template <int N>
__device__ __forceinline__ void team_barrier() {
    if constexpr (Plan<N>::NT > 64) __syncthreads();
}

template <bool INV>
__device__ __forceinline__ void bfly2(float2& a, float2& b) {
    float2 s = cadd(a, b), d = csub(a, b);
    a = s;
    b = d;
}

// radix-4 butterfly in registers, natural output order; INV selects exp(+i...) kernels
template <bool INV>
__device__ __forceinline__ void bfly4(float2& x0, float2& x1, float2& x2, float2& x3) {
    float2 a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = csub(x1, x3);
    float2 jd = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);  // (+/-) i * d
    x0 = cadd(a, c);
    x2 = csub(a, c);
    x1 = cadd(b, jd);
    x3 = csub(b, jd);
}

// 16-point DFT in registers: input v[n] (n = n0 + 4 n1), output X[k] in v[pos16(k)]
__host__ __device__ constexpr int pos16(int k) { return 4 * (k & 3) + (k >> 2); }
template <bool INV>
__device__ __forceinline__ void dft16(float2* v) {
    constexpr float C8 = 0.92387953251128673848f, S8 = 0.38268343236508978178f;
    constexpr float R2 = 0.70710678118654752440f;
    constexpr float SG = INV ? -1.f : 1.f;  // sign of the sine terms
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) bfly4<INV>(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    // position n0 + 4 k1 holds Y[n0][k1]; multiply by W16^(+- n0 k1) = c -+ i s
    auto mulw = [](float2 z, float c, float s) {  // z * (c - i s)
        return make_float2(fmaf(z.x, c, z.y * s), fmaf(z.y, c, -z.x * s));
    };
    v[5] = mulw(v[5], C8, SG * S8);
    v[9] = mulw(v[9], R2, SG * R2);
    v[13] = mulw(v[13], S8, SG * C8);
    v[6] = mulw(v[6], R2, SG * R2);
    v[10] = INV ? make_float2(-v[10].y, v[10].x) : make_float2(v[10].y, -v[10].x);
    v[14] = mulw(v[14], -R2, SG * R2);
    v[7] = mulw(v[7], S8, SG * C8);
    v[11] = mulw(v[11], -R2, SG * R2);
    v[15] = mulw(v[15], -C8, -SG * S8);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) bfly4<INV>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

// layout of a pass's twiddle table: rows [k][t] for the big radix-16 tables (> 32 KB, beyond the
// vector L1), columns [t][k] otherwise
__host__ __device__ constexpr bool tw_rows(int r, int ns) { return r == 16 && ns * r * 8 > 32 * 1024; }

template <bool INV>
__device__ __forceinline__ float2 twid(const float2* __restrict__ tw, int idx) {
    float2 w = tw[idx];
    if (INV) w.y = -w.y;
    return w;
}

template <int N, bool REVERSED>
__host__ __device__ constexpr int pass_radix(int p) {
    return REVERSED ? Plan<N>::radix(Plan<N>::NPASS - 1 - p) : Plan<N>::radix(p);
}
template <int N, bool REVERSED>
__host__ __device__ constexpr int pass_ns(int p) {  // product of the radices before pass p
    int ns = 1;
    for (int q = 0; q < p; ++q) ns *= pass_radix<N, REVERSED>(q);
    return ns;
}
// entries of the table of pass p (NS * R; the NS == 1 pass has no twiddles)
template <int N, bool REVERSED>
__host__ __device__ constexpr int tw_pass_len(int p) {
    return pass_ns<N, REVERSED>(p) == 1 ? 0 : pass_ns<N, REVERSED>(p) * pass_radix<N, REVERSED>(p);
}
template <int N, bool REVERSED>
__host__ __device__ constexpr int tw_seq_len() {
    int n = 0;
    for (int p = 0; p < Plan<N>::NPASS; ++p) n += tw_pass_len<N, REVERSED>(p);
    return n;
}
template <int N, bool REVERSED>
__host__ __device__ constexpr int tw_offset(int p) {
    int n = REVERSED ? tw_seq_len<N, false>() : 0;
    for (int q = 0; q < p; ++q) n += tw_pass_len<N, REVERSED>(q);
    return n;
}
template <int N>
__host__ __device__ constexpr int tw_table_len() { return tw_seq_len<N, false>() + tw_seq_len<N, true>(); }

// One Stockham pass of radix R with sub-transform length NS (both compile time).
template <int N, int R, int NS, int TWOFF, bool INV, bool FROM_REG, bool TO_REG>
__device__ __forceinline__ void pass(float2 (&v)[Plan<N>::VMAX], float2* __restrict__ buf,
                                     const float2* __restrict__ tw, int tid) {
    using P = Plan<N>;
    constexpr int NBF = N / R, BPT = P::bpt(R);
    constexpr bool FULL = (NBF % P::NT) == 0;  // every thread owns BPT butterflies
    // lidx(a + c) = lidx(a) + c + c/16 when c is a multiple of 16: one shift-add per butterfly,
    // the rest are immediates
    if (!FROM_REG) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            int j = tid + i * P::NT;
            if (FULL || j < NBF) {
                if constexpr (NBF % 16 == 0) {
                    const int b = lidx(j);
#pragma unroll
                    for (int t = 0; t < R; ++t) v[i * R + t] = buf[b + t * (NBF + NBF / 16)];
                } else {
#pragma unroll
                    for (int t = 0; t < R; ++t) v[i * R + t] = buf[lidx(j + t * NBF)];
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        int j = tid + i * P::NT;
        if (FULL || j < NBF) {
            float2* x = &v[i * R];
            if (NS > 1) {
                if constexpr (tw_rows(R, NS)) {
                    // rows [k][t]: a thread's 15 twiddles are one 128-byte row, a wave reads 64
                    // consecutive rows (each line is fetched once and reused by the 15 loads)
                    const float2* __restrict__ row = tw + TWOFF + (j & (NS - 1)) * R;
#pragma unroll
                    for (int t = 1; t < R; ++t) x[t] = cmul(x[t], twid<INV>(row, t));
                } else {
                    // columns [t][k]: for one t the lanes (consecutive k) read consecutive entries
                    const float2* __restrict__ col = tw + TWOFF + (j & (NS - 1));
#pragma unroll
                    for (int t = 1; t < R; ++t) x[t] = cmul(x[t], twid<INV>(col, t * NS));
                }
            }
            if (R == 16) {
                dft16<INV>(x);
                float2 o[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) o[t] = x[pos16(t)];
#pragma unroll
                for (int t = 0; t < 16; ++t) x[t] = o[t];
            } else if (R == 4) {
                bfly4<INV>(x[0], x[1], x[2], x[3]);
            } else {
                bfly2<INV>(x[0], x[1]);
            }
        }
    }
    if (!TO_REG) {
        if (!FROM_REG) team_barrier<N>();  // every read of this pass done before any write
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            int j = tid + i * P::NT;
            if (FULL || j < NBF) {
                int k = j & (NS - 1);
                int o = (j - k) * R + k;
                if constexpr (NS % 16 == 0) {
                    const int b = lidx(o);
#pragma unroll
                    for (int t = 0; t < R; ++t) buf[b + t * (NS + NS / 16)] = v[i * R + t];
                } else if constexpr (NS == 1 && R == 16) {
                    const int b = 17 * j;  // lidx(16 j + t) = 17 j + t
#pragma unroll
                    for (int t = 0; t < R; ++t) buf[b + t] = v[i * R + t];
                } else {
#pragma unroll
                    for (int t = 0; t < R; ++t) buf[lidx(o + t * NS)] = v[i * R + t];
                }
            }
        }
        team_barrier<N>();
    }
}

template <int N, int PI, bool INV, bool REVERSED, bool FIRST_FROM_REG, bool LAST_TO_REG>
struct Passes {
    static __device__ __forceinline__ void run(float2 (&v)[Plan<N>::VMAX], float2* buf,
                                               const float2* tw, int tid) {
        using P = Plan<N>;
        if constexpr (PI < P::NPASS) {
            constexpr int R = pass_radix<N, REVERSED>(PI);
            constexpr int NS = pass_ns<N, REVERSED>(PI);
            constexpr int OFF = tw_offset<N, REVERSED>(PI);
            pass<N, R, NS, OFF, INV, (PI == 0) && FIRST_FROM_REG, (PI == P::NPASS - 1) && LAST_TO_REG>(
                v, buf, tw, tid);
            Passes<N, PI + 1, INV, REVERSED, FIRST_FROM_REG, LAST_TO_REG>::run(v, buf, tw, tid);
        }
    }
};

// Full transform of one team (blockDim may hold several teams: every barrier is a
// __syncthreads of the whole workgroup, so all teams must call this together).
// FIRST_FROM_REG: the caller filled v for the first pass (see header); otherwise the data
// is taken from buf (caller barriers after filling it).  Result: natural order in buf
// (after a barrier), or -- LAST_TO_REG -- X[j + t*N/R] in v[i*R + t], buf undefined.
// When FIRST_FROM_REG the caller guarantees nobody still reads buf.
template <int N, bool INV, bool FIRST_FROM_REG, bool LAST_TO_REG, bool REVERSED = false>
__device__ __forceinline__ void fft(float2 (&v)[Plan<N>::VMAX], float2* buf, const float2* tw,
                                    int tid) {
    Passes<N, 0, INV, REVERSED, FIRST_FROM_REG, LAST_TO_REG>::run(v, buf, tw, tid);
}

// iterate the register set of a radix-R pass: f(i*R + t, n) with n = (tid + i*NT) + t*N/R
template <int N, int R, typename F>
__device__ __forceinline__ void for_each_reg(int tid, F f) {
    using P = Plan<N>;
    constexpr int NBF = N / R, BPT = P::bpt(R);
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        int j = tid + i * P::NT;
        if ((NBF % P::NT) == 0 || j < NBF) {
#pragma unroll
            for (int t = 0; t < R; ++t) f(i * R + t, j + t * NBF);
        }
    }
}

// host: fill the per-length twiddle blob (tw_table_len<N>() entries)
template <int N, bool REVERSED, typename F2>
inline void fill_tw_seq(F2* out) {
    for (int p = 0; p < Plan<N>::NPASS; ++p) {
        const int ns = pass_ns<N, REVERSED>(p), r = pass_radix<N, REVERSED>(p);
        if (ns == 1) continue;
        F2* t = out + tw_offset<N, REVERSED>(p);
        for (int k = 0; k < ns; ++k)
            for (int q = 0; q < r; ++q) {
                double a = -2.0 * 3.14159265358979323846 * (double)q * (double)k / ((double)ns * r);
                F2& e = tw_rows(r, ns) ? t[k * r + q] : t[q * ns + k];
                e.x = (float)cos(a);
                e.y = (float)sin(a);
            }
    }
}
template <int N, typename F2>
inline void fill_tw_table(F2* out) {
    fill_tw_seq<N, false>(out);
    fill_tw_seq<N, true>(out);
}

}  // namespace dsfft
