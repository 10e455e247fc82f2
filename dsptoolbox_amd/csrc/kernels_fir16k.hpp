// FIR block convolution with 16384-point blocks (2049 .. 8193 taps): overlap-save kernel built on
// the register-resident 4096-point transform of kernels_welch4096.hpp.
//
//   16384 = 4 x 4096, 1024 threads = 4 groups of 256.  Group q owns the sub-spectrum Z[4k' + q]:
//   forward (once per input block and channel pair), decimation in frequency:
//       b_q[n'] = ( sum_j z[n' + 4096 j] W4^(jq) ) W16384^(n' q) ,   Z[4k'+q] = FFT4096(b_q)[k']
//   per filter, decimation in time:
//       g_q = IFFT4096( Z[4k'+q] H[4k'+q] ) ,   y[n' + 4096 j] = sum_q W4^(-jq) W16384^(-n' q) g_q[n']
//   The 4096-point transforms run in registers (16 values per thread, W4096 twiddles in registers,
//   two LDS exchanges, welch4096::fft4096 and its mirror image ifft4096 below); the spectrum of
//   the input block stays in registers across all filters; the tap spectra are stored in exactly
//   the register layout (k_permute), so a filter costs 16 coalesced loads per thread; the radix-4
//   recombination across the groups goes through LDS once.  Every output sample is produced and
//   stored once (overlap-save), two channels ride one complex transform.
//
//   The generic kernel (kernels_generic.hpp: k_fir<16384>, 512 threads x 32 values) spills, reads
//   344 KB of twiddle tables from L2 per transform and runs one lockstep workgroup per CU at two
//   waves per SIMD; this one keeps 16 waves per CU busy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <vector>

#include "kernels_welch4096.hpp"

namespace fir16k {

namespace w4 = welch4096;
constexpr int NBIG = 16384, M = 4096, NTB = 1024;
constexpr int LDS_BYTES = (4 * w4::BUF_C + 256) * 8;  // four exchange buffers + W256 table: 149 504 B

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}

// inverse radix-4 butterfly in place: (a,b,c,d) <- DFT4 with exp(+i...) kernels
__device__ __forceinline__ void r4i(float2& a, float2& b, float2& c, float2& d) {
    float2 s0 = make_float2(a.x + c.x, a.y + c.y), d0 = make_float2(a.x - c.x, a.y - c.y);
    float2 s1 = make_float2(b.x + d.x, b.y + d.y), d1 = make_float2(b.x - d.x, b.y - d.y);
    a = make_float2(s0.x + s1.x, s0.y + s1.y);
    c = make_float2(s0.x - s1.x, s0.y - s1.y);
    b = make_float2(d0.x - d1.y, d0.y + d1.x);  // d0 + i d1
    d = make_float2(d0.x + d1.y, d0.y - d1.x);  // d0 - i d1
}

// Inverse of welch4096::dft16: input X[k] in v[4*(k&3) + (k>>2)], output x[n] in v[n] (x16).
//   x[n1 + 4 n0] = sum_k0 W4^(-n0 k0) [ W16^(-n1 k0) sum_k1 X[k0 + 4 k1] W4^(-n1 k1) ]
__device__ __forceinline__ void idft16(float2 (&v)[16]) {
    constexpr float C8 = 0.92387953251128673848f, S8 = 0.38268343236508978178f;
    constexpr float R2 = 0.70710678118654752440f;
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) r4i(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
    // position 4 k0 + n1 times W16^(-n1 k0) = c + i s
    auto mulw = [](float2 z, float c, float s) {  // z * (c + i s)
        return make_float2(fmaf(z.x, c, -z.y * s), fmaf(z.y, c, z.x * s));
    };
    v[4 + 1] = mulw(v[4 + 1], C8, S8);                                           // W16^-1
    v[4 + 2] = make_float2((v[6].x - v[6].y) * R2, (v[6].x + v[6].y) * R2);      // W16^-2
    v[4 + 3] = mulw(v[4 + 3], S8, C8);                                           // W16^-3
    v[8 + 1] = make_float2((v[9].x - v[9].y) * R2, (v[9].x + v[9].y) * R2);      // W16^-2
    v[8 + 2] = make_float2(-v[10].y, v[10].x);                                   // W16^-4 = +i
    v[8 + 3] = make_float2(-(v[11].x + v[11].y) * R2, (v[11].x - v[11].y) * R2);  // W16^-6
    v[12 + 1] = mulw(v[12 + 1], S8, C8);                                         // W16^-3
    v[12 + 2] = make_float2(-(v[14].x + v[14].y) * R2, (v[14].x - v[14].y) * R2);  // W16^-6
    v[12 + 3] = mulw(v[12 + 3], -C8, -S8);                                       // W16^-9 = -W16^-1
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) r4i(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
}

// Mirror image of welch4096::fft4096 (single exchange buffer): v[pos16(k3)] = Y[t + 256 k3] on
// entry, v[n1] = sum_k Y[k] W4096^(-k (t + 256 n1)) on return.  The caller guarantees nobody
// still reads buf when this is entered.
// twt: the W4096^(t k1) table in global memory ([15][256]); this thread's 15 values are fetched
// into the registers the data has just left, behind the second exchange barrier (keeping them
// resident costs 30 registers the 128-register budget of a 1024-thread workgroup does not have).
// FOLD (kernels_fir16k.hpp's own use): twt is a [16][256] table W16384^(t (4 k1 + q)) of the calling
// group q -- the last pass's twiddle W4096^(t k1) times the group's output twiddle W16384^(t q), which
// is the same for every output of a thread and therefore can ride on the 16 inputs of the last
// 16-point transform (k1 = 0 included): the caller's recombination is left with the wave-uniform
// factor W64^(n1 q) only -- 60 vector instructions less per thread and filter.
template <bool FOLD = false>
__device__ __forceinline__ void ifft4096(float2 (&v)[16], const float2* __restrict__ twt, float2* __restrict__ buf,
                                         const float2* __restrict__ tw2, int t) {
    idft16(v);  // v[n3]
    float4* row = reinterpret_cast<float4*>(buf + t * w4::L2S);
#pragma unroll
    for (int j = 0; j < 8; ++j) row[j] = make_float4(v[2 * j].x, v[2 * j].y, v[2 * j + 1].x, v[2 * j + 1].y);
    const int k1u = t >> 4, n3 = t & 15;
    float2 w2[15];  // W256 twiddles: fetched behind the barrier into the registers v just left
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) w2[k2 - 1] = tw2[k2 * 16 + n3];
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) v[w4::pos16(k2)] = buf[(16 * k2 + k1u) * w4::L2S + n3];
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) v[w4::pos16(k2)] = cmulc(v[w4::pos16(k2)], w2[k2 - 1]);
    idft16(v);  // v[n2]
    __syncthreads();  // every read of the row image is done before it is overwritten
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) buf[k1u * w4::L1S + 16 * n2 + n3] = v[n2];
    float2 w1[FOLD ? 16 : 15];
#pragma unroll
    for (int k1 = FOLD ? 0 : 1; k1 < 16; ++k1) w1[FOLD ? k1 : k1 - 1] = twt[(FOLD ? k1 : k1 - 1) * 256 + t];
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[w4::pos16(k1)] = buf[k1 * w4::L1S + t];
#pragma unroll
    for (int k1 = FOLD ? 0 : 1; k1 < 16; ++k1) v[w4::pos16(k1)] = cmulc(v[w4::pos16(k1)], w1[FOLD ? k1 : k1 - 1]);
    idft16(v);  // v[n1]
}

// twn: [4][256] W16384^(t q) then [4][16] W64^(n1 q), then the folded last-pass table
// [4][16][256] W16384^(t (4 k1 + q))  (fp64-computed)
constexpr int TWN_FOLD = 4 * 256 + 4 * 16;
constexpr int TWN_LEN = TWN_FOLD + 4 * 16 * 256;
inline void host_tables(std::vector<float2>& t) {
    t.resize(TWN_LEN);
    for (int q = 0; q < 4; ++q)
        for (int k1 = 0; k1 < 16; ++k1)
            for (int tt = 0; tt < 256; ++tt) {
                double a = -2.0 * M_PI * (double)(tt * (4 * k1 + q)) / 16384.0;
                t[TWN_FOLD + (q * 16 + k1) * 256 + tt] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
    for (int q = 0; q < 4; ++q) {
        for (int tt = 0; tt < 256; ++tt) {
            double a = -2.0 * M_PI * (double)(tt * q) / 16384.0;
            t[q * 256 + tt] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        for (int n1 = 0; n1 < 16; ++n1) {
            double a = -2.0 * M_PI * (double)(n1 * q) / 64.0;
            t[4 * 256 + q * 16 + n1] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    }
}

// hperm[(((k*4 + q)*8 + s/2)*256 + t)*2 + (s&1)] = hs[k*16384 + 4*(t + 256*k3(s)) + q], s = pos16(k3):
// register slots (2j, 2j+1) of a thread are adjacent -> one 16-byte load per two slots
struct PermArgs {
    const float2* hs;
    int n_filt;
    float2* hperm;
};
__global__ void k_permute(PermArgs p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.n_filt * NBIG) return;
    const int s = (int)(i & 1) + 2 * (int)((i >> 9) & 7), t = (int)((i >> 1) & 255), q = (int)((i >> 12) & 3);
    const int64_t k = i >> 14;
    const int k3 = (s >> 2) + 4 * (s & 3);  // inverse of pos16
    p.hperm[i] = p.hs[k * NBIG + 4 * (t + 256 * k3) + q];
}

struct Args {
    const float* x;
    int64_t n_samples, ldx, ld_y;
    int n_ch, n_filt, n_taps;
    const float2* twt;    // welch4096::host_tables()
    const float2* twn;    // host_tables() above
    const float2* hperm;  // [n_filt][4][8][256][2], 1/N folded in
    float* y;             // [(k*n_ch + c)*ld_y + n]
    int block0;           // first block of this launch
};

// grid = (n_blocks, ceil(n_ch/2), filter slices); block j covers outputs [j L, (j+1) L), L = 16384 - (n_taps-1).
// PLAIN: the number of discarded samples T1 = n_taps - 1 is a multiple of 4, whole block inside the
// signal, 16-byte aligned rows: every kept group of four samples is stored with one 16-byte store
// behind one compare per quarter (4097 taps: quarters 1..3 whole), no test per element (the host launches
// the interior blocks with PLAIN and the rest without: keeping both store paths in one kernel costs
// registers the 128-VGPR budget does not have).
template <bool PLAIN>
__global__ __launch_bounds__(NTB) void k_fir(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    const int tid = threadIdx.x, t = tid & 255;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 8);  // group = sub-spectrum, wave-uniform
    float2* buf = lds + q * w4::BUF_C;
    float2* tw2 = lds + 4 * w4::BUF_C;
    float2* comb = lds;  // [4][4096] recombination image, overlays the exchange buffers
    const int T1 = p.n_taps - 1;
    const int L = NBIG - T1;
    const int64_t out0 = (int64_t)(blockIdx.x + p.block0) * L;
    const int ca = 2 * blockIdx.y, cb = ca + 1;
    const bool vb = cb < p.n_ch;
    const float* xa = p.x + (int64_t)ca * p.ldx;
    const float* xb = vb ? p.x + (int64_t)cb * p.ldx : xa;
    const float mb = vb ? 1.f : 0.f;

    if (tid < 256) tw2[tid] = p.twt[15 * 256 + tid];
    const float2 wt = p.twn[q * 256 + t];      // W16384^(t q)
    const float2* c64 = p.twn + 4 * 256 + q * 16;  // W64^(n1 q), wave-uniform

    // ---- forward: radix-4 across the quarters, twiddle, 4096-point transform of this group
    float2 v[16];
    {
        const int64_t s0 = out0 - T1;
        const int64_t last = p.n_samples - 1;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            float2 z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t gidx = s0 + t + 256 * n1 + (int64_t)M * j;
                const int64_t cg = gidx < 0 ? 0 : (gidx > last ? last : gidx);  // clamp + select
                const float a = xa[cg], b = xb[cg];
                const float ok = (gidx == cg) ? 1.f : 0.f;
                z[j] = make_float2(a * ok, b * (ok * mb));
            }
            float2 b;
            if (q == 0)
                b = make_float2(z[0].x + z[1].x + z[2].x + z[3].x, z[0].y + z[1].y + z[2].y + z[3].y);
            else if (q == 1)  // z0 - i z1 - z2 + i z3
                b = make_float2(z[0].x + z[1].y - z[2].x - z[3].y, z[0].y - z[1].x - z[2].y + z[3].x);
            else if (q == 2)
                b = make_float2(z[0].x - z[1].x + z[2].x - z[3].x, z[0].y - z[1].y + z[2].y - z[3].y);
            else  // z0 + i z1 - z2 - i z3
                b = make_float2(z[0].x - z[1].y - z[2].x + z[3].y, z[0].y + z[1].x - z[2].y - z[3].x);
            v[n1] = cmul(b, cmul(wt, c64[n1]));
        }
    }
    __syncthreads();  // W256 table written
    {
        w4::Tw tw;  // only the forward transform keeps its W4096 twiddles in registers
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) tw.w[k1 - 1] = p.twt[(k1 - 1) * 256 + t];
        w4::fft4096_plain<false>(v, tw, buf, tw2, t);
    }
    float2 z[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) z[s] = v[s];

    // ---- per filter: multiply, inverse, recombine, store
    const float4* hq = reinterpret_cast<const float4*>(p.hperm) + (int64_t)q * 8 * 256 + t;
    auto load_taps = [&](const float4* h) {  // tap spectrum of one filter in register layout
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 r = h[j * 256];
            v[2 * j] = make_float2(r.x, r.y);
            v[2 * j + 1] = make_float2(r.z, r.w);
        }
    };
    // gridDim.z workgroups share a block: each applies a slice of the filters (the forward transform
    // is repeated per slice; the host picks the split that fills the last round of workgroups best)
    const int k0 = (int)((int64_t)blockIdx.z * p.n_filt / gridDim.z);
    const int k1 = (int)((int64_t)(blockIdx.z + 1) * p.n_filt / gridDim.z);
    load_taps(hq + (int64_t)k0 * (NBIG / 2));
    for (int k = k0; k < k1; ++k) {
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = cmul(z[s], v[s]);
        __syncthreads();  // the previous filter's recombination reads are done
        const float2* twq = p.twn + TWN_FOLD + q * 16 * 256;
        asm volatile("" : "+s"(twq));  // not loop invariant for the compiler: no hoisting into 32 live registers
        ifft4096<true>(v, twq, buf, tw2, t);
        __syncthreads();  // every group has read its last exchange image: the buffers become comb
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) comb[q * M + t + 256 * n1] = cmulc(v[n1], c64[n1]);  // W16384^(-t q) is folded in
        // the registers are free: fetch the next filter's tap spectrum behind the recombination
        if (k + 1 < k1) load_taps(hq + (int64_t)(k + 1) * (NBIG / 2));
        __syncthreads();
        float* oa = p.y + ((int64_t)k * p.n_ch + ca) * p.ld_y + (out0 - T1);
        float* ob = oa + p.ld_y;
        // each thread recombines four consecutive samples n0 .. n0+3 of every quarter: 16-byte
        // LDS reads and 16-byte stores.  Interior blocks with exactly 4096 discarded samples
        // (4097 taps) store quarters 1..3 whole, without a test per store.
        const int n0 = 4 * (q * 256 + t);
        float2 u[4][4];  // [sub-spectrum][sample]
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float4* src = reinterpret_cast<const float4*>(comb + qq * M + n0);
            const float4 a = src[0], b = src[1];
            u[qq][0] = make_float2(a.x, a.y);
            u[qq][1] = make_float2(a.z, a.w);
            u[qq][2] = make_float2(b.x, b.y);
            u[qq][3] = make_float2(b.z, b.w);
        }
        // y[n + 4096 j] = sum_q u_q[n] W4^(-jq), one quarter j at a time (few live registers)
        auto quarter = [&](int j, int i) {
            const float2 u0 = u[0][i], u1 = u[1][i], u2 = u[2][i], u3 = u[3][i];
            if (j == 0) return make_float2(u0.x + u1.x + u2.x + u3.x, u0.y + u1.y + u2.y + u3.y);
            if (j == 1) return make_float2(u0.x - u1.y - u2.x + u3.y, u0.y + u1.x - u2.y - u3.x);  // u0 + i u1 - u2 - i u3
            if (j == 2) return make_float2(u0.x - u1.x + u2.x - u3.x, u0.y - u1.y + u2.y - u3.y);
            return make_float2(u0.x + u1.y - u2.x - u3.y, u0.y - u1.x - u2.y + u3.x);  // u0 - i u1 - u2 + i u3
        };
        if (PLAIN) {
            // T1 is a multiple of 4: a thread's four consecutive samples of a quarter are kept or
            // discarded together (one compare per quarter, no test per element, no bounds test)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n0 + M * j >= T1) {
                    const float2 y0 = quarter(j, 0), y1 = quarter(j, 1), y2 = quarter(j, 2), y3 = quarter(j, 3);
                    *reinterpret_cast<float4*>(oa + n0 + M * j) = make_float4(y0.x, y1.x, y2.x, y3.x);
                    if (vb) *reinterpret_cast<float4*>(ob + n0 + M * j) = make_float4(y0.y, y1.y, y2.y, y3.y);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int nn = n0 + i + M * j;
                    const float2 yy = quarter(j, i);
                    if (nn >= T1 && out0 + (nn - T1) < p.n_samples) {
                        oa[nn] = yy.x;
                        if (vb) ob[nn] = yy.y;
                    }
                }
        }
    }
}

}  // namespace fir16k
