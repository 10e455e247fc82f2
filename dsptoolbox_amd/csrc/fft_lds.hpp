// Generic power-of-two complex FFT held in LDS by one workgroup (gfx950).
//
// Stockham autosort, radix-4 passes (+ one radix-2 pass when log2 N is odd),
// in place in ONE LDS buffer of N float2: every thread first reads all the
// butterfly inputs it owns into registers, the workgroup barriers, then the
// outputs go back to the (different) autosort positions.  The first pass can
// take its inputs straight from registers (the caller loads x[j + t*N/4] from
// global memory, coalesced over j), and the last pass can leave its outputs in
// registers: thread j then owns X[j + t*N/4], which is again the input set of
// a following first pass (used by the FIR kernel: forward -> multiply ->
// inverse without touching LDS in between).
//
// Twiddles come from a table tw[m] = exp(-2 pi i m / N), m < N, computed in
// fp64 on the host (one per length, cached in the context).
#pragma once
#include <hip/hip_runtime.h>

namespace dsfft {

__host__ __device__ constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

// threads cooperating on one length-N complex FFT
__host__ __device__ constexpr int threads_for(int n) {
    return n >= 4096 ? 512 : (n >= 2048 ? 256 : (n >= 1024 ? 128 : 64));
}

template <int N>
struct Cfg {
    static constexpr int LOGN = ilog2(N);
    static constexpr int NT = threads_for(N);
    static constexpr int NB4 = N / 4;                              // radix-4 butterflies per pass
    static constexpr int BPT = (NB4 + NT - 1) / NT;                // per thread
    static constexpr int NPASS4 = LOGN / 2;
    static constexpr bool ODD = (LOGN & 1) != 0;
    static constexpr int NB2 = N / 2;
    static constexpr int BPT2 = (NB2 + NT - 1) / NT;
    static constexpr int LDS_BYTES = N * 8;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {  // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// radix-4 butterfly in registers; INV selects exp(+i...) kernels
template <bool INV>
__device__ __forceinline__ void bfly4(float2& x0, float2& x1, float2& x2, float2& x3) {
    float2 a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = csub(x1, x3);
    float2 jd = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);  // (+/-) i * d
    x0 = cadd(a, c);
    x2 = csub(a, c);
    x1 = cadd(b, jd);
    x3 = csub(b, jd);
}

template <bool INV>
__device__ __forceinline__ float2 twid(const float2* __restrict__ tw, int idx) {
    float2 w = tw[idx];
    if (INV) w.y = -w.y;
    return w;
}

// One radix-4 Stockham pass with sub-transform length NS (compile time).
//   v[i][t]: in = x[j + t*N/4], j = tid + i*NT; out -> position (j-k)*4 + k + t*NS, k = j % NS
// FROM_REG: inputs already in v (first pass);  TO_REG: keep outputs in v (last pass)
template <int N, int NS, bool INV, bool FROM_REG, bool TO_REG>
__device__ __forceinline__ void pass4(float2 (&v)[Cfg<N>::BPT][4], float2* __restrict__ buf,
                                      const float2* __restrict__ tw, int tid) {
    using C = Cfg<N>;
    if (!FROM_REG) {
#pragma unroll
        for (int i = 0; i < C::BPT; ++i) {
            int j = tid + i * C::NT;
            if (C::NB4 >= C::NT || j < C::NB4) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[i][t] = buf[j + t * C::NB4];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < C::BPT; ++i) {
        int j = tid + i * C::NT;
        if (C::NB4 >= C::NT || j < C::NB4) {
            if (NS > 1) {
                int k = j & (NS - 1);
                constexpr int STEP = N / (NS * 4);
                v[i][1] = cmul(v[i][1], twid<INV>(tw, k * STEP));
                v[i][2] = cmul(v[i][2], twid<INV>(tw, 2 * k * STEP));
                v[i][3] = cmul(v[i][3], twid<INV>(tw, 3 * k * STEP));
            }
            bfly4<INV>(v[i][0], v[i][1], v[i][2], v[i][3]);
        }
    }
    if (!TO_REG) {
        if (!FROM_REG) __syncthreads();  // every read of this pass done before any write
#pragma unroll
        for (int i = 0; i < C::BPT; ++i) {
            int j = tid + i * C::NT;
            if (C::NB4 >= C::NT || j < C::NB4) {
                int k = j & (NS - 1);
                int o = ((j - k) << 2) + k;
#pragma unroll
                for (int t = 0; t < 4; ++t) buf[o + t * NS] = v[i][t];
            }
        }
        __syncthreads();
    }
}

// final radix-2 pass (log2 N odd), NS = N/2: in x[j], x[j+N/2]; out j, j+N/2
template <int N, bool INV>
__device__ __forceinline__ void pass2_last(float2* __restrict__ buf, const float2* __restrict__ tw,
                                           int tid) {
    using C = Cfg<N>;
    float2 a[C::BPT2], b[C::BPT2];
#pragma unroll
    for (int i = 0; i < C::BPT2; ++i) {
        int j = tid + i * C::NT;
        if (C::NB2 >= C::NT || j < C::NB2) {
            a[i] = buf[j];
            b[i] = cmul(buf[j + C::NB2], twid<INV>(tw, j));  // k = j, STEP = 1
        }
    }
    // in == out positions per thread: no barrier needed between read and write
#pragma unroll
    for (int i = 0; i < C::BPT2; ++i) {
        int j = tid + i * C::NT;
        if (C::NB2 >= C::NT || j < C::NB2) {
            buf[j] = cadd(a[i], b[i]);
            buf[j + C::NB2] = csub(a[i], b[i]);
        }
    }
    __syncthreads();
}

template <int N, int P, bool INV, bool FIRST_FROM_REG, bool LAST_TO_REG>
struct Passes {
    static __device__ __forceinline__ void run(float2 (&v)[Cfg<N>::BPT][4], float2* buf,
                                               const float2* tw, int tid) {
        using C = Cfg<N>;
        constexpr int NS = 1 << (2 * P);
        constexpr bool last = (P == C::NPASS4 - 1) && !C::ODD;
        if constexpr (P < C::NPASS4) {
            pass4<N, NS, INV, (P == 0) && FIRST_FROM_REG, last && LAST_TO_REG>(v, buf, tw, tid);
            Passes<N, P + 1, INV, FIRST_FROM_REG, LAST_TO_REG>::run(v, buf, tw, tid);
        }
    }
};

// Full transform.  If FIRST_FROM_REG the caller has filled v[i][t] = x[tid + i*NT + t*N/4].
// Otherwise the data is taken from buf (caller barriers after filling it).
// Result: natural order in buf (after a barrier), or -- LAST_TO_REG, even log2 N
// only -- X[tid + i*NT + t*N/4] in v[i][t] with buf left undefined.
template <int N, bool INV, bool FIRST_FROM_REG, bool LAST_TO_REG>
__device__ __forceinline__ void fft(float2 (&v)[Cfg<N>::BPT][4], float2* buf, const float2* tw,
                                    int tid) {
    using C = Cfg<N>;
    static_assert(!(LAST_TO_REG && C::ODD), "register output needs an even log2 N");
    Passes<N, 0, INV, FIRST_FROM_REG, LAST_TO_REG>::run(v, buf, tw, tid);
    if constexpr (C::ODD) pass2_last<N, INV>(buf, tw, tid);
}

}  // namespace dsfft
