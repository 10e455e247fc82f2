// EXPERIMENT (round 4): Welch H1 / H2 / H3, nfft 4096, 50 % overlap, one input
// channel, as TWO launches with per-unit LAST-ARRIVER finishes and no wait anywhere:
//
//   k_x3la : k_x3 (one workgroup per input frame pair) + the chunk's last arriver sums the chunk's
//            px rows -> psx[q]  (replaces the px slice sums in k_y3's prologue)
//   k_y3la : k_y3's pair loop; behind it a workgroup stores its folded partials write-through
//            (16-byte sc1 stores), every wave drains, barrier, ONE agent-scope add on done[c]; the
//            workgroup whose add returns n_chunks - 1 folds the channel's n_chunks partials (16-byte
//            sc1 loads) and finishes its 2049 bins (dsk::tf_from_sums).  Nobody spins; k_welch_finish
//            and one kernel boundary disappear.
//
// Hand-off form: MI355X_MICROARCH.md, 'Hand-offs measured with sc1 loads in place of the acquire',
// first row (one lane per storing workgroup adds to ONE unsharded counter behind every wave's
// vmcnt(0) and the workgroup barrier; the last adder loads after its add has returned, the other
// waves behind a barrier it joins; 16-byte sc1 stores and loads).  Partial rows are padded to
// NBP = 2080 elements so that no 128-byte line is shared by two workgroups' rows.
//
// reference: transfer_functions/transfer_functions.py:476-534 (per-channel _welch loop + H / coherence)
#pragma once
#include "../../dsptoolbox_amd/csrc/kernels_welch4096w.hpp"

namespace welch4096 {

constexpr int NBP = 2080;     // padded partial row (elements): 2080 * 4 = 65 * 128 bytes
constexpr int NITEM = 513;    // items of four consecutive bins: 4 * 513 = 2052 >= NB
constexpr int LA_MAX = 1024;  // counters: [0, LA_MAX) chunk arrivals (k_x3la), [LA_MAX, 2 LA_MAX) channel arrivals

#ifndef W4LA_STAMPS
#define W4LA_STAMPS 0  // dev only: per-workgroup s_memrealtime stamps -> LaArgs::stamps[block][8]
#endif
#if W4LA_STAMPS
#define LA_STAMP(i)                                                                                       \
    do {                                                                                                  \
        if (threadIdx.x == 0) la.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define LA_STAMP(i)
#endif

struct LaArgs {
    Args a;          // px, psx, pxy, pyy rows have stride NBP
    unsigned* cnt;   // 2 * LA_MAX words, zero before the first launch; the kernels leave them zero
    int mode;
    dsk::FinishPar fin;
    float2* tf;      // [NB][n_ch]
    float* coh;      // [NB][n_ch]
#if W4LA_STAMPS
    unsigned long long* stamps;  // [grid][8]: 0 start, 1 loop done, 2 stores issued, 3 drained, 4 ticket, 5 finish done
#endif
};

__device__ __forceinline__ void la_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t la_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
typedef unsigned la_v4u __attribute__((ext_vector_type(4)));
// aux bit 4 = sc1: write-through store / L1-bypassing, agent-coherent load
__device__ __forceinline__ void la_st128(float4 v, __amdgpu_buffer_rsrc_t r, int byte_off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(la_v4u, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ float4 la_ld128(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}

// ---- input spectra + per-chunk px sums -------------------------------------------------------
__global__ __launch_bounds__(NT) void k_x3la(LaArgs la) {
    const Args& p = la.a;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * L1S;
    const int tid = threadIdx.x;
    const int pr = (int)blockIdx.x;  // one input channel
    Tw6 tw;
    float2 v[16];
    {
        Raw<true> raw;
        const __amdgpu_buffer_rsrc_t rs = channel_rsrc(p.sig, p.n_samples);
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
    float4* xo = reinterpret_cast<float4*>(p.xs + (int64_t)pr * N) + tid;
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
    const __amdgpu_buffer_rsrc_t po = la_rsrc(p.px + (int64_t)pr * NBP, NBP * 4);
    for (int i = tid; i < NITEM; i += NT) {
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * i + j, kk = min(k, N / 2);
            r[j] = k < NB ? 0.5f * (pw[fold_pos(kk)] + pw[fold_pos((N - kk) & (N - 1))]) : 0.f;
        }
        la_st128(make_float4(r[0], r[1], r[2], r[3]), po, 16 * i);
    }
    // which chunk does this pair belong to (n_chunks <= 768: a short scalar scan)
    int q = 0, p0 = 0, p1 = 0;
    for (; q < p.n_chunks; ++q) {
        chunk_range(p, q, p0, p1);
        if (pr < p1) break;
    }
    la_drain();
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(tw2);
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(&la.cnt[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag[0] != (unsigned)(p1 - p0 - 1)) return;
    // last arriver of chunk q: its px rows, fp64 -> psx[q] (read by the NEXT launch: plain stores)
    if (tid == 0) __hip_atomic_store(&la.cnt[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const __amdgpu_buffer_rsrc_t pin = la_rsrc(p.px + (int64_t)p0 * NBP, (uint32_t)((p1 - p0) * NBP * 4));
    float4* so = reinterpret_cast<float4*>(p.psx + (int64_t)q * NBP);
    for (int i = tid; i < NITEM; i += NT) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int r0 = 0; r0 < p1 - p0; r0 += 8) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = la_ld128(pin, (min(r0 + j, p1 - p0 - 1) * NBP + 4 * i) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (r0 + j < p1 - p0) {
                    s[0] += (double)t[j].x;
                    s[1] += (double)t[j].y;
                    s[2] += (double)t[j].z;
                    s[3] += (double)t[j].w;
                }
        }
        so[i] = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
    }
}

// ---- output channels + per-channel last-arriver finish ----------------------------------------
__device__ __forceinline__ void la_tail(const LaArgs& la, float2* lds, float2 (&T)[16], float (&P)[16], int q, int c, int tid);

__global__ __launch_bounds__(NT, 3) void k_y3la(LaArgs la) {
    const Args& p = la.a;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * L1S;
    float* winl = reinterpret_cast<float*>(lds + 16 * L1S + 256);
    const int tid = threadIdx.x;
    int q, c;
    {
        const int b = blockIdx.x, total = p.n_chunks * p.n_ch;
        const int u = (total & 7) == 0 ? (b & 7) * (total >> 3) + (b >> 3) : b;
        q = u / p.n_ch;
        c = u - q * p.n_ch;
    }
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    LA_STAMP(0);
    Tw6 tw;
    load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) winl[tid + 256 * n1] = p.window[tid + 256 * n1];
    const float* ch = p.sig + (int64_t)c * p.ld;
    int p0, p1;
    chunk_range(p, q, p0, p1);
    float2 T[16];
    float P[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        T[j] = make_float2(0.f, 0.f);
        P[j] = 0.f;
    }
    float carry[8], nx[16];
    const __amdgpu_buffer_rsrc_t rs = channel_rsrc(ch, p.n_samples);
    const __amdgpu_buffer_rsrc_t xrs = la_rsrc(p.xs + (int64_t)p0 * N, (uint32_t)((p1 - p0) * (N * 8)));
    if (p0 < p1) {
        const int off0 = 4 * (2 * p0 * 2048 + tid);
#pragma unroll
        for (int j = 0; j < 8; ++j) carry[j] = ld_sample(rs, off0 + 1024 * j);
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = ld_sample(rs, off0 + 1024 * (8 + j));
    }
    Stamp ts;
    float winr[16];  // (a thread reads back the window values it wrote itself: no barrier)
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) winr[n1] = winl[tid + 256 * n1];
    for (int pr = p0; pr < p1; ++pr) {
        float2 v[16];
        const int level16 = ((p1 - pr - 1) * 16) / (p1 - p0);
        set_prio(level16, (pr * 5) & 3);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const float w = winr[n1];
            const float a = n1 < 8 ? carry[n1] : nx[n1 - 8];
            v[n1] = make_float2(a * w, nx[n1] * w);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) carry[j] = nx[8 + j];
        if (needs_drop(p, pr)) drop_second(v);
        float2 xw[16];
        const int off1 = 4 * ((2 * pr + 2) * 2048 + tid) + 1024 * 8;
        const int xoff = (pr - p0) * (N * 8) + tid * 16;
        fft4096_wi(
            v, tw, buf, tw2, tid,
            [&](int g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nx[4 * g + j] = ld_sample(rs, off1 + 1024 * (4 * g + j));
            },
            [&](int g) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 q4 = __builtin_bit_cast(
                        float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff + 4096 * (2 * g + j), 0, 0));
                    xw[2 * (2 * g + j)] = make_float2(q4.x, q4.y);
                    xw[2 * (2 * g + j) + 1] = make_float2(q4.z, q4.w);
                }
            },
            ts, level16);
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) {
            const float2 z = v[pos16(k3)];
            const float2 w = xw[k3];
            T[k3].x = fmaf(w.x, z.x, fmaf(w.y, z.y, T[k3].x));
            T[k3].y = fmaf(w.x, z.y, fmaf(-w.y, z.x, T[k3].y));
            P[k3] = fmaf(z.x, z.x, fmaf(z.y, z.y, P[k3]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) winr[n1] = winl[tid + 256 * n1];
        __builtin_amdgcn_sched_barrier(0);
    }
    LA_STAMP(1);
    __builtin_amdgcn_s_setprio(3);  // what follows is the tail of the launch
    // (the thread index re-formed from the wave index kept in a scalar register: nothing the tail
    // needs is hoisted above the pair loop as a vector register)
    return la_tail(la, lds, T, P, q, c,
                   (wave_s << 6) | (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
}

__device__ __forceinline__ void la_tail(const LaArgs& la, float2* lds, float2 (&T)[16], float (&P)[16], int q, int c, int tid) {
    const Args& p = la.a;
    float2* buf = lds;
    if (p.detrend && tid == 0) P[0] = 0.f;
    const int bt = bin_thread(tid);
    const int64_t row = (int64_t)q * p.n_ch + c;
    const __amdgpu_buffer_rsrc_t rxy = la_rsrc(p.pxy + row * NBP, NBP * 8), ryy = la_rsrc(p.pyy + row * NBP, NBP * 4);
    __syncthreads();
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) buf[fold_pos(bt + 256 * k3)] = T[k3];
    __syncthreads();
    for (int i = tid; i < NITEM; i += NT) {
        float2 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * i + j, kk = min(k, N / 2);
            const float2 a = buf[fold_pos(kk)], b = buf[fold_pos((N - kk) & (N - 1))];
            r[j] = k < NB ? make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y)) : make_float2(0.f, 0.f);
        }
        la_st128(make_float4(r[0].x, r[0].y, r[1].x, r[1].y), rxy, 32 * i);
        la_st128(make_float4(r[2].x, r[2].y, r[3].x, r[3].y), rxy, 32 * i + 16);
    }
    __syncthreads();
    float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) pw[fold_pos(bt + 256 * k3)] = P[k3];
    __syncthreads();
    for (int i = tid; i < NITEM; i += NT) {
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * i + j, kk = min(k, N / 2);
            r[j] = k < NB ? 0.5f * (pw[fold_pos(kk)] + pw[fold_pos((N - kk) & (N - 1))]) : 0.f;
        }
        la_st128(make_float4(r[0], r[1], r[2], r[3]), ryy, 16 * i);
    }
    LA_STAMP(2);
    la_drain();
    __syncthreads();
    LA_STAMP(3);
    unsigned* flag = reinterpret_cast<unsigned*>(lds + 16 * L1S);  // the W256 table is dead
    if (tid == 0)
        flag[0] = __hip_atomic_fetch_add(&la.cnt[LA_MAX + c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    LA_STAMP(4);
    if (flag[0] != (unsigned)(p.n_chunks - 1)) return;

    // ---- last arriver of channel c: chunk sums in fp64, H and coherence ---------------------------
    if (tid == 0) __hip_atomic_store(&la.cnt[LA_MAX + c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t slab = (int64_t)p.n_ch * NBP;  // elements between the rows of consecutive chunks
    const __amdgpu_buffer_rsrc_t axy = la_rsrc(p.pxy + (int64_t)c * NBP, (uint32_t)(((int64_t)(p.n_chunks - 1) * slab + NBP) * 8));
    const __amdgpu_buffer_rsrc_t ayy = la_rsrc(p.pyy + (int64_t)c * NBP, (uint32_t)(((int64_t)(p.n_chunks - 1) * slab + NBP) * 4));
    const float4* __restrict__ psx = reinterpret_cast<const float4*>(p.psx);
    for (int i = tid; i < NITEM; i += NT) {
        double sxx[4] = {}, syy[4] = {}, sr[4] = {}, si[4] = {};
        for (int q0 = 0; q0 < p.n_chunks; q0 += 6) {
            float4 a[6], b[6], y[6], x[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int qq = min(q0 + j, p.n_chunks - 1);
                a[j] = la_ld128(axy, (int)((qq * slab + 4 * i) * 8));
                b[j] = la_ld128(axy, (int)((qq * slab + 4 * i) * 8 + 16));
                y[j] = la_ld128(ayy, (int)((qq * slab + 4 * i) * 4));
                x[j] = psx[(int64_t)qq * (NBP / 4) + i];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (q0 + j < p.n_chunks) {
                    sr[0] += (double)a[j].x, si[0] += (double)a[j].y, sr[1] += (double)a[j].z, si[1] += (double)a[j].w;
                    sr[2] += (double)b[j].x, si[2] += (double)b[j].y, sr[3] += (double)b[j].z, si[3] += (double)b[j].w;
                    syy[0] += (double)y[j].x, syy[1] += (double)y[j].y, syy[2] += (double)y[j].z, syy[3] += (double)y[j].w;
                    sxx[0] += (double)x[j].x, sxx[1] += (double)x[j].y, sxx[2] += (double)x[j].z, sxx[3] += (double)x[j].w;
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * i + j;
            if (k < NB) {
                dsk::cd sxy{sr[j], si[j] + 0.0};  // + 0.0: a sum of -0 partials becomes +0 like the reference's mean
                dsk::tf_from_sums(sxx[j], sxy, syy[j], k, la.mode, la.fin, la.tf[(int64_t)k * p.n_ch + c],
                                  la.coh[(int64_t)k * p.n_ch + c]);
            }
        }
    }
    LA_STAMP(5);
}

}  // namespace welch4096
