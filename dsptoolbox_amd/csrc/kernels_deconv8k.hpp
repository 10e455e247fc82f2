// Spectral division for 8192-point transforms with ONE inverse spectrum shared by all channels
// (the batched config: many responses against one sweep): ir = irfft(rfft(y) R), two channels per
// complex transform, on the register-resident 4096-point transform of kernels_welch4096.hpp.
//
//   8192 = 2 x 4096, 512 threads = 2 groups of 256; group q owns the sub-spectrum Z[2k' + q]:
//       b_q[n'] = (z[n'] + (-1)^q z[n' + 4096]) W8192^(n' q) ,  Z[2k'+q] = FFT4096(b_q)[k']
//       g_q = IFFT4096( Z[2k'+q] Rf[2k'+q] ) ,  y[n' + 4096 j] = sum_q (-1)^(jq) W8192^(-n' q) g_q[n']
//   z = ya + i yb.  Because both channels see the same real impulse response, its full
//   Hermitian spectrum Rf multiplies the PACKED spectrum directly (no separation of the two
//   channels): Rf[k] = R[k], Rf[N-k] = conj R[k], and the real parts of R[0], R[N/2] (numpy's irfft
//   ignores the imaginary parts there).  74 KB of LDS: two workgroups per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <vector>

#include "kernels_fir16k.hpp"

namespace deconv8k {

// W32^n1 = exp(-2 pi i n1 / 32), n1 < 16, as compile-time constants (the float values of host_tables()'s second part).
// kernels_welch8192.hpp uses them since round 4: read from the table in global memory they are wave-uniform, but hipcc
// issues VECTOR loads for them, and a vector load in the middle of a loop's sample loads makes each a separate wait.
// (The deconvolution kernels below keep the table reads: with the constants k_deconv3q's loads and stores flow freely
// -- no s_waitcnt vmcnt(0) per group of four any more -- but it spills 33 ... 46 registers instead of 5 at its 128 and
// measured 47-49 us against 46.5, profiles/r04_deconv_constants.txt.)
__device__ __forceinline__ float2 w32(int n1) {
    constexpr float C[16] = {1.0f, 0.9807852506637573f, 0.9238795042037964f, 0.8314695954322815f, 0.7071067690849304f,
                             0.5555702447891235f, 0.3826834261417389f, 0.19509032368659973f, 6.123234262925839e-17f,
                             -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f,
                             -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f};
    constexpr float S[16] = {-0.0f, -0.19509032368659973f, -0.3826834261417389f, -0.5555702447891235f, -0.7071067690849304f,
                             -0.8314695954322815f, -0.9238795042037964f, -0.9807852506637573f, -1.0f, -0.9807852506637573f,
                             -0.9238795042037964f, -0.8314695954322815f, -0.7071067690849304f, -0.5555702447891235f,
                             -0.3826834261417389f, -0.19509032368659973f};
    return make_float2(C[n1], S[n1]);
}

namespace w4 = welch4096;
using fir16k::cmul;
using fir16k::cmulc;
constexpr int N = 8192, M = 4096, NTB = 512;
constexpr int LDS_BYTES = (2 * w4::BUF_C + 256) * 8;  // 75 776 B

// twn: [256] W8192^t then [16] W32^n1  (fp64-computed)
constexpr int TWN_LEN = 256 + 16;
inline void host_tables(std::vector<float2>& t) {
    t.resize(TWN_LEN);
    for (int tt = 0; tt < 256; ++tt) {
        double a = -2.0 * M_PI * (double)tt / 8192.0;
        t[tt] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int n1 = 0; n1 < 16; ++n1) {
        double a = -2.0 * M_PI * (double)n1 / 32.0;
        t[256 + n1] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
}

struct Args {
    const float* y;
    int64_t n_samples, ld, n_out, ld_out;
    int n_ch;
    const float2* twt;    // welch4096::host_tables()
    const float2* twn;    // host_tables() above
    const float2* r;      // [N/2+1] shared inverse spectrum
    float* ir;
};

// grid = (ceil(n_ch/2), n_items)
__global__ __launch_bounds__(NTB, 4) void k_deconv(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    const int tid = threadIdx.x, t = tid & 255;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 8);
    float2* buf = lds + q * w4::BUF_C;
    float2* tw2 = lds + 2 * w4::BUF_C;
    float2* comb = lds;  // [2][4096] recombination image, overlays the exchange buffers
    const int ca = 2 * blockIdx.x, cb = ca + 1;
    const bool vb = cb < p.n_ch;
    const int64_t item = blockIdx.y;
    const float* ya = p.y + (item * p.n_ch + ca) * p.ld;
    const float* yb = vb ? ya + p.ld : ya;
    const float mb = vb ? 1.f : 0.f;
    if (tid < 256) tw2[tid] = p.twt[15 * 256 + tid];
    const float2 wt = p.twn[t];
    const float2* c32 = p.twn + 256;

    // The shared inverse spectrum in the register layout: slot s = pos16(k3) holds bin
    // k = 2 (t + 256 k3) + q, gathered straight from the one-sided r (32 KB, L2 resident; a
    // separate permutation launch would cost 6 us of a 50 us step) before anything else, so the
    // loads are in flight during the forward transform.  Rf[k] = R[k], Rf[N-k] = conj R[k], real
    // at k = 0 and N/2; 1/N folded in.  Branch-free index math.
    float2 rf[16];
    {
        const float inv = 1.0f / (float)N;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k3 = (s >> 2) + 4 * (s & 3);
            const int k = 2 * (t + 256 * k3) + q;
            const float2 r = p.r[min(k, N - k)];
            const float sg = (k == 0 || k == N / 2) ? 0.f : (k < N / 2 ? inv : -inv);
            rf[s] = make_float2(r.x * inv, r.y * sg);
        }
    }

    // ---- forward
    float2 v[16];
    {
        const int last = (int)(p.n_samples < (int64_t)N ? p.n_samples : (int64_t)N) - 1;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            float2 z[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int g = t + 256 * n1 + M * j;
                const int cg = g > last ? last : g;  // clamp + select: no branch per load
                const float a = ya[cg], b = yb[cg];
                const float ok = (g == cg) ? 1.f : 0.f;
                z[j] = make_float2(a * ok, b * (ok * mb));
            }
            if (q == 0)
                v[n1] = make_float2(z[0].x + z[1].x, z[0].y + z[1].y);
            else
                v[n1] = cmul(make_float2(z[0].x - z[1].x, z[0].y - z[1].y), cmul(wt, c32[n1]));
        }
    }
    __syncthreads();  // W256 table written
    {
        w4::Tw tw;
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) tw.w[k1 - 1] = p.twt[(k1 - 1) * 256 + t];
        w4::fft4096_plain<false>(v, tw, buf, tw2, t);
    }
    // ---- multiply by the shared inverse spectrum (fetched at the top of the kernel)
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = cmul(v[s], rf[s]);
    __syncthreads();  // the forward transform's last exchange image has been read
    fir16k::ifft4096(v, p.twt, buf, tw2, t);
    __syncthreads();  // both groups have read their last exchange image: the buffers become comb
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1)
        comb[q * M + t + 256 * n1] = (q == 0) ? v[n1] : cmulc(v[n1], cmul(wt, c32[n1]));
    __syncthreads();
    // ---- y[n'] = u0 + u1, y[n' + 4096] = u0 - u1; four consecutive samples per thread and chunk
    float* oa = p.ir + (item * p.n_ch + ca) * p.ld_out;
    float* ob = oa + p.ld_out;
    const bool plain = p.n_out >= N && (p.ld_out & 3) == 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n0 = 2048 * c + 4 * tid;
        const float4* s0 = reinterpret_cast<const float4*>(comb + n0);
        const float4* s1 = reinterpret_cast<const float4*>(comb + M + n0);
        const float4 a0 = s0[0], a1 = s0[1], b0 = s1[0], b1 = s1[1];
        const float4 lo_x = make_float4(a0.x + b0.x, a0.z + b0.z, a1.x + b1.x, a1.z + b1.z);
        const float4 lo_y = make_float4(a0.y + b0.y, a0.w + b0.w, a1.y + b1.y, a1.w + b1.w);
        const float4 hi_x = make_float4(a0.x - b0.x, a0.z - b0.z, a1.x - b1.x, a1.z - b1.z);
        const float4 hi_y = make_float4(a0.y - b0.y, a0.w - b0.w, a1.y - b1.y, a1.w - b1.w);
        if (plain) {
            *reinterpret_cast<float4*>(oa + n0) = lo_x;
            *reinterpret_cast<float4*>(oa + n0 + M) = hi_x;
            if (vb) {
                *reinterpret_cast<float4*>(ob + n0) = lo_y;
                *reinterpret_cast<float4*>(ob + n0 + M) = hi_y;
            }
        } else {
            const float lx[4] = {lo_x.x, lo_x.y, lo_x.z, lo_x.w}, ly[4] = {lo_y.x, lo_y.y, lo_y.z, lo_y.w};
            const float hx[4] = {hi_x.x, hi_x.y, hi_x.z, hi_x.w}, hy[4] = {hi_y.x, hi_y.y, hi_y.z, hi_y.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (n0 + i < p.n_out) {
                    oa[n0 + i] = lx[i];
                    if (vb) ob[n0 + i] = ly[i];
                }
                if (n0 + i + M < p.n_out) {
                    oa[n0 + i + M] = hx[i];
                    if (vb) ob[n0 + i + M] = hy[i];
                }
            }
        }
    }
}

// ---- the same division in ONE 256-thread group per channel pair (round 3) ----------------------------
// k_deconv above runs the two sub-spectra side by side in a 512-thread workgroup: two workgroups per
// CU, all eight waves of a workgroup in step through six barriers, 45 us for 1024 stereo items (0.37
// of the HBM roofline, 46 % of the wave cycles waiting).  Here one group of 256 threads runs the two
// 4096-point sub-problems ONE AFTER THE OTHER on the headline kernel's transform pair
// (welch4096::fft4096_w, natural order in -> permuted bins; fir4k::ifft4096_w, its mirror image): the
// second half block waits in 32 registers while the first is transformed, multiplied and transformed
// back, then the first result waits for the second.  36 KB of LDS and <= 168 registers: THREE
// independent workgroups per CU, no cross-group recombination through LDS (both halves end in the same
// thread), every sample loaded and stored through range-checked buffer accesses (no edge code).
}  // namespace deconv8k
#include "kernels_fir4k.hpp"
namespace deconv8k {

constexpr int LDS_BYTES_3 = fir4k::LDS_BYTES;

// grid = ceil(n_ch / 2) * n_items
__global__ __launch_bounds__(256, 3) void k_deconv3(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    const int tid = threadIdx.x;
    const int pairs = (p.n_ch + 1) / 2;
    const int64_t item = (int)blockIdx.x / pairs;
    const int ca = 2 * ((int)blockIdx.x - (int)item * pairs), cb = ca + 1;
    const bool vb = cb < p.n_ch;
    const uint32_t in_bytes = (uint32_t)(p.n_samples < (int64_t)N ? p.n_samples : (int64_t)N) * 4u;
    const float* ya = p.y + (item * p.n_ch + ca) * p.ld;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ya), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ya + (vb ? p.ld : 0)), 0, vb ? (int)in_bytes : 0, 0x00020000);
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    const float2 wt = p.twn[tid];           // W8192^tid
    const float2* c32 = p.twn + 256;        // W32^n1 (wave-uniform)
    // b_0 = z[n'] + z[n' + 4096] ,  b_1 = (z[n'] - z[n' + 4096]) W8192^n' ,  n' = tid + 256 n1
    float2 v[16], b1[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int off = 4 * (tid + 256 * n1);
        const float2 z0 = make_float2(w4::ld_sample(ra, off), w4::ld_sample(rb, off));
        const float2 z1 = make_float2(w4::ld_sample(ra, off + 4 * M), w4::ld_sample(rb, off + 4 * M));
        v[n1] = make_float2(z0.x + z1.x, z0.y + z1.y);
        b1[n1] = cmul(make_float2(z0.x - z1.x, z0.y - z1.y), cmul(wt, c32[n1]));
    }
    // The shared inverse spectrum of sub-spectrum q in the transform's register layout: slot
    // s = pos16(k3) <-> bin k = 2 (bt + 256 k3) + q;  Rf[k] = R[k], Rf[N - k] = conj R[k], real at 0 and
    // N / 2 (numpy's irfft ignores the imaginary parts there); 1 / N folded in.  Branch-free.
    const int bt = w4::bin_thread(tid);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.r), 0, (N / 2 + 1) * 8, 0x00020000);
    auto gather = [&](float2 (&rf)[16], int q) {
        const float inv = 1.0f / (float)N;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k3 = (s >> 2) + 4 * (s & 3);
            const int k = 2 * (bt + 256 * k3) + q;
            const float2 r = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rr, 8 * min(k, N - k), 0, 0));
            const float sg = (k == 0 || k == N / 2) ? 0.f : (k < N / 2 ? inv : -inv);
            rf[s] = make_float2(r.x * inv, r.y * sg);
        }
    };
    // (the transforms with their LDS traffic spread between the butterflies: the plain forms, whose
    // exchanges hipcc issues in bursts, made this kernel SLOWER than the 512-thread one -- 58 against 45 us)
    float2* tw2p = lds + 16 * w4::L1S + 256;
    fir4k::fill_tw2p(tw2p, p.twt, tid);
    float2 rf[16];
    gather(rf, 0);
    w4::Stamp ts;
    auto none = [](int) {};
    __syncthreads();  // the tables
    w4::fft4096_wi(v, tw, buf, tw2, tid, none, none, ts, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = cmul(v[s], rf[s]);
    gather(rf, 1);  // in flight during the transform back
    fir4k::ifft4096_wi(v, tw, buf, tw2p, tid, none, none);
    float2 g0[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        g0[n1] = v[n1];
        v[n1] = b1[n1];
    }
    __syncthreads();  // the column reads of the transform back: the forward transform stores into the image at once
    w4::fft4096_wi(v, tw, buf, tw2, tid, none, none, ts, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = cmul(v[s], rf[s]);
    fir4k::ifft4096_wi(v, tw, buf, tw2p, tid, none, none);
    // y[n'] = g_0 + W8192^(-n') g_1 ,  y[n' + 4096] = g_0 - W8192^(-n') g_1
    float* oa = p.ir + (item * p.n_ch + ca) * p.ld_out;
    const uint32_t out_bytes = (uint32_t)(p.n_out < (int64_t)N ? p.n_out : (int64_t)N) * 4u;
    const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(oa, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(oa + (vb ? p.ld_out : 0), 0, vb ? (int)out_bytes : 0, 0x00020000);
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const float2 u = cmulc(v[n1], cmul(wt, c32[n1]));
        const int off = 4 * (tid + 256 * n1);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].x + u.x), sa, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].y + u.y), sb, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].x - u.x), sa, off + 4 * M, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].y - u.y), sb, off + 4 * M, 0, 0);
    }
}

// The same with FOUR workgroups per CU (<= 128 registers: the shared spectrum of a sub-problem is gathered behind its
// forward transform instead of being held through it, the second half block waits as 32 finished values): all 1024
// channel pairs of the benchmark are resident at once, as independent 256-thread workgroups (four channel pairs as
// four TEAMS of one 1024-thread workgroup were slower: 54.5 against 46.7 us).
// grid = ceil(n_ch / 2) * n_items
__global__ __launch_bounds__(256, 4) void k_deconv3q(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    const int tid = threadIdx.x;
    const int pairs = (p.n_ch + 1) / 2;
    const int64_t item = (int)blockIdx.x / pairs;
    const int ca = 2 * ((int)blockIdx.x - (int)item * pairs), cb = ca + 1;
    const bool vb = cb < p.n_ch;
    const uint32_t in_bytes = (uint32_t)(p.n_samples < (int64_t)N ? p.n_samples : (int64_t)N) * 4u;
    const float* ya = p.y + (item * p.n_ch + ca) * p.ld;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ya), 0, (int)in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ya + (vb ? p.ld : 0)), 0, vb ? (int)in_bytes : 0, 0x00020000);
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    const float2 wt = p.twn[tid];           // W8192^tid
    const float2* c32 = p.twn + 256;        // W32^n1 (wave-uniform)
    // b_0 = z[n'] + z[n' + 4096] ,  b_1 = (z[n'] - z[n' + 4096]) W8192^n' ,  n' = tid + 256 n1
    float2 v[16], b1[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int off = 4 * (tid + 256 * n1);
        const float2 z0 = make_float2(w4::ld_sample(ra, off), w4::ld_sample(rb, off));
        const float2 z1 = make_float2(w4::ld_sample(ra, off + 4 * M), w4::ld_sample(rb, off + 4 * M));
        v[n1] = make_float2(z0.x + z1.x, z0.y + z1.y);
        float2 y1 = cmul(make_float2(z0.x - z1.x, z0.y - z1.y), cmul(wt, c32[n1]));
        asm volatile("" : "+v"(y1.x), "+v"(y1.y));  // parked as the product, not as its factors
        b1[n1] = y1;
    }
    // The shared inverse spectrum of sub-spectrum q in the transform's register layout: slot
    // s = pos16(k3) <-> bin k = 2 (bt + 256 k3) + q;  Rf[k] = R[k], Rf[N - k] = conj R[k], real at 0 and
    // N / 2 (numpy's irfft ignores the imaginary parts there); 1 / N folded in.  Branch-free.
    const int bt = w4::bin_thread(tid);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.r), 0, (N / 2 + 1) * 8, 0x00020000);
    auto gather = [&](float2 (&rf)[16], int q) {
        const float inv = 1.0f / (float)N;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k3 = (s >> 2) + 4 * (s & 3);
            const int k = 2 * (bt + 256 * k3) + q;
            const float2 r = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rr, 8 * min(k, N - k), 0, 0));
            const float sg = (k == 0 || k == N / 2) ? 0.f : (k < N / 2 ? inv : -inv);
            rf[s] = make_float2(r.x * inv, r.y * sg);
        }
    };
    // (the transforms with their LDS traffic spread between the butterflies: the plain forms, whose
    // exchanges hipcc issues in bursts, made this kernel SLOWER than the 512-thread one -- 58 against 45 us)
    float2* tw2p = lds + 16 * w4::L1S + 256;
    fir4k::fill_tw2p(tw2p, p.twt, tid);
    float2 rf[16];
    w4::Stamp ts;
    auto none = [](int) {};
    __syncthreads();  // the tables
    w4::fft4096_wi(v, tw, buf, tw2, tid, none, none, ts, 0);
    gather(rf, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = cmul(v[s], rf[s]);
    fir4k::ifft4096_wi(v, tw, buf, tw2p, tid, none, none);
    float2 g0[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        g0[n1] = v[n1];
        v[n1] = b1[n1];
    }
    __syncthreads();  // the column reads of the transform back: the forward transform stores into the image at once
    w4::fft4096_wi(v, tw, buf, tw2, tid, none, none, ts, 0);
    gather(rf, 1);
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = cmul(v[s], rf[s]);
    fir4k::ifft4096_wi(v, tw, buf, tw2p, tid, none, none);
    // y[n'] = g_0 + W8192^(-n') g_1 ,  y[n' + 4096] = g_0 - W8192^(-n') g_1
    float* oa = p.ir + (item * p.n_ch + ca) * p.ld_out;
    const uint32_t out_bytes = (uint32_t)(p.n_out < (int64_t)N ? p.n_out : (int64_t)N) * 4u;
    const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(oa, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(oa + (vb ? p.ld_out : 0), 0, vb ? (int)out_bytes : 0, 0x00020000);
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const float2 u = cmulc(v[n1], cmul(wt, c32[n1]));
        const int off = 4 * (tid + 256 * n1);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].x + u.x), sa, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].y + u.y), sb, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].x - u.x), sa, off + 4 * M, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].y - u.y), sb, off + 4 * M, 0, 0);
    }
}

// ---- round 5: persistent, pipelined ---------------------------------------------------------------------------------
// What the three kernels above have in common: ONE unit (channel pair of an item) per workgroup and every workgroup of the
// grid resident at once, so the whole chip loads (67 MB), then the whole chip transforms, then the whole chip stores:
// HBM idles while the vector pipe works and the other way round (VALU utilisation 0.23, 1.05 x the algorithmic traffic and
// still 0.37 of the roofline).  And inside the 128 registers of four workgroups per CU the inverse spectrum's gather
// (16 eight-byte loads per sub-problem, 64 different cache lines per wave instruction: neighbouring lanes sit 32 bins apart)
// and the W32 table reads serialize into ~80 memory round trips per wave.
// Here: TWO workgroups per CU (256 registers each) stay for the whole launch and take units u = block, block + grid, ...;
//   * the 64 samples of the NEXT unit are requested (one burst, raw-buffer loads into 64 registers) as soon as the current
//     unit's samples have been folded into its two sub-problems, and arrive during its four transforms;
//   * the 64 results are stored in one burst that drains during the next unit's transforms;
//   * the shared inverse spectrum is read from a PERMUTED copy (k_rperm: the transform's register layout, scale and the
//     real-bin rule folded in; 64 KB, written once per call): 16 coalesced 8-byte loads per sub-problem through one
//     descriptor, no index arithmetic, no 64-line gathers;
//   * W32^n1 as compile-time constants.
// No scratch.  A unit past the end is a descriptor of zero records (loads return 0, no branch around loads).
constexpr int RPERM_LEN = 2 * 16 * 256;  // float2: [q][slot s][tid]

// rperm[(q 16 + s) 256 + tid] = Rf[2 (bin_thread(tid) + 256 k3(s)) + q] / N, Rf the full Hermitian spectrum of the real
// impulse response r describes (real at bins 0 and N / 2: numpy's irfft ignores the imaginary parts there).  grid = 32 x 256.
struct RpArgs {
    const float2* r;  // [N / 2 + 1]
    float2* rperm;    // [RPERM_LEN]
};
__global__ __launch_bounds__(256) void k_rperm(RpArgs a) {
    const float2* __restrict__ r = a.r;
    float2* __restrict__ rperm = a.rperm;
    const int tid = threadIdx.x, q = (int)blockIdx.x >> 4, s = (int)blockIdx.x & 15;
    const int k3 = (s >> 2) + 4 * (s & 3);
    const int k = 2 * (w4::bin_thread(tid) + 256 * k3) + q;
    const float2 v = r[min(k, N - k)];
    const float inv = 1.0f / (float)N;
    const float sg = (k == 0 || k == N / 2) ? 0.f : (k < N / 2 ? inv : -inv);
    rperm[(int)blockIdx.x * 256 + tid] = make_float2(v.x * inv, v.y * sg);
}

struct PArgs {
    Args a;
    const float2* rperm;
    int n_units;  // ceil(n_ch / 2) * n_items
};

__global__ __launch_bounds__(256, 2) void k_deconv_p(PArgs pp) {
    const Args& p = pp.a;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    float2* tw2p = lds + 16 * w4::L1S + 256;
    const int tid = threadIdx.x;
    const int pairs = (p.n_ch + 1) / 2;
    const uint32_t in_bytes = (uint32_t)(p.n_samples < (int64_t)N ? p.n_samples : (int64_t)N) * 4u;
    const uint32_t out_bytes = (uint32_t)(p.n_out < (int64_t)N ? p.n_out : (int64_t)N) * 4u;
    const int off0 = 4 * tid;
    float za[32], zb[32];  // raw samples of a unit: [n1 + 16 j], half block j
    // every sample of unit u (wave-uniform descriptors; past the last unit: zero records)
    auto request = [&](int u) {
        const bool live = u < pp.n_units;
        const int uu = live ? u : 0;
        const int64_t item = uu / pairs;
        const int ca = 2 * (uu - (int)item * pairs);
        const bool vb = ca + 1 < p.n_ch;
        const float* ya = p.y + (item * p.n_ch + ca) * p.ld;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ya), 0, live ? (int)in_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ya + (vb ? p.ld : 0)), 0, (live && vb) ? (int)in_bytes : 0, 0x00020000);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            za[n1] = w4::ld_sample(ra, off0 + 1024 * n1);
            zb[n1] = w4::ld_sample(rb, off0 + 1024 * n1);
            za[16 + n1] = w4::ld_sample(ra, off0 + 1024 * n1 + 4 * M);
            zb[16 + n1] = w4::ld_sample(rb, off0 + 1024 * n1 + 4 * M);
        }
    };
    int u = blockIdx.x;
    request(u);
    __builtin_amdgcn_sched_barrier(0);
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    fir4k::fill_tw2p(tw2p, p.twt, tid);
    const float2 wt = p.twn[tid];  // W8192^tid
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.rperm), 0, RPERM_LEN * 8, 0x00020000);
    auto spectrum = [&](float2 (&rf)[16], int q) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
            rf[s] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rr, ((q * 16 + s) * 256) * 8 + 8 * tid, 0, 0));
    };
    w4::Stamp ts;
    auto none = [](int) {};
    __syncthreads();  // the tables
    for (; u < pp.n_units; u += gridDim.x) {
        // b_0 = z[n'] + z[n' + 4096] ,  b_1 = (z[n'] - z[n' + 4096]) W8192^n' ,  n' = tid + 256 n1
        float2 v[16], b1[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            v[n1] = make_float2(za[n1] + za[16 + n1], zb[n1] + zb[16 + n1]);
            b1[n1] = cmul(make_float2(za[n1] - za[16 + n1], zb[n1] - zb[16 + n1]), cmul(wt, w32(n1)));
        }
        __builtin_amdgcn_sched_barrier(0);
        request(u + (int)gridDim.x);  // the next unit's samples: in flight during this unit's transforms
        __builtin_amdgcn_sched_barrier(0);
        float2 rf[16];
        w4::fft4096_wi(v, tw, buf, tw2, tid, none, none, ts, 0);
        spectrum(rf, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = cmul(v[s], rf[s]);
        fir4k::ifft4096_wi(v, tw, buf, tw2p, tid, none, none);
        float2 g0[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            g0[n1] = v[n1];
            v[n1] = b1[n1];
        }
        __syncthreads();  // the column reads of the transform back: the forward transform stores into the image at once
        w4::fft4096_wi(v, tw, buf, tw2, tid, none, none, ts, 0);
        spectrum(rf, 1);
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = cmul(v[s], rf[s]);
        fir4k::ifft4096_wi(v, tw, buf, tw2p, tid, none, none);
        // y[n'] = g_0 + W8192^(-n') g_1 ,  y[n' + 4096] = g_0 - W8192^(-n') g_1
        const int64_t item = u / pairs;
        const int ca = 2 * (u - (int)item * pairs);
        const bool vb = ca + 1 < p.n_ch;
        float* oa = p.ir + (item * p.n_ch + ca) * p.ld_out;
        const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(oa, 0, (int)out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t sb = __builtin_amdgcn_make_buffer_rsrc(oa + (vb ? p.ld_out : 0), 0, vb ? (int)out_bytes : 0, 0x00020000);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const float2 w = cmulc(v[n1], cmul(wt, w32(n1)));
            const int off = 4 * (tid + 256 * n1);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].x + w.x), sa, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].y + w.y), sb, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].x - w.x), sa, off + 4 * M, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g0[n1].y - w.y), sb, off + 4 * M, 0, 0);
        }
        __syncthreads();  // the last column reads of this unit's transform back, before the next unit's first image stores
    }
}

}  // namespace deconv8k
