// Welch H1 / H2 / H3, nfft 4096, 50 % overlap, ONE input channel: the whole estimate in ONE launch.
// gfx950.  NOT part of the library: tools/exp only (see the note at the end).  (compute_transfer_function, transfer_functions/transfer_functions.py:476-534: the
// per-channel _welch loop, _spectral_methods.py:10-173, and the H / coherence lines :525-534.)
//
// kernels_welch4096w.hpp runs the step as three launches: k_x3 (input spectra) -> k_y3 (output
// channels, chunk partials) -> k_welch_finish (chunk sums, H, coherence).  The two small kernels
// and the two dependent launch boundaries were 20 % of the step.  Here the same grid of
// (chunk, channel) workgroups -- all resident at once, three per CU -- does all of it:
//
//   1. PRODUCE  workgroup (q, c) with c < pairs(q) first transforms pair p0(q) + c of the INPUT
//               channel: spectrum -> xs (16-byte write-through stores), folded power -> px row
//               (write-through), every wave drains its stores, barrier, ONE lane adds the number of
//               pairs produced to xready[q] (agent-scope atomic).
//   2. MAIN     k_y3's pair loop on the output channel.  The input spectra are first needed two
//               thirds into the first transform: there, once, one lane polls xready[q] (relaxed
//               agent-scope loads, s_sleep), does ONE agent-scope acquire, and the workgroup's
//               barrier releases the other waves -- by then the producers (which started at the
//               same time and had one transform to do) are normally done.
//   3. PARTIALS every workgroup sums its slice of the chunk's px rows (fp64) -> psx[q], folds its
//               T / P accumulators -> pxy / pyy[q][c]; all write-through; drain, barrier, one lane
//               adds 1 to `published`.
//   4. FINISH   every workgroup waits for published == grid (one lane polls; the input auto spectrum
//               of a bin is the work of another channel's workgroups, so the wait is grid-wide),
//               acquires, and finishes slice q of channel c's bins: chunk sums in fp64, then H and the
//               coherence (dsk::tf_from_sums, the finish kernel's own code) -- all 768 workgroups
//               share the finish, one round of loads each.  (A first version let the last arriver of
//               a channel finish it alone: 46 us of dependent slab reads in one workgroup.)  The last
//               workgroup through resets the counters for the next launch.
//
// Inter-workgroup visibility follows cdna_hip_programming.md Guideline 16: payload stored
// write-through (sc1) and drained by EVERY storing wave before the workgroup's barrier, one
// agent-scope atomic as the signal, ONE relaxed poll + ONE agent acquire + vmcnt(0) + barrier on the
// consumer, then plain loads.  Nothing is read before it is published (xs, px, psx, pxy, pyy are
// only ever read behind a matched poll / ticket), so no cache can hold a stale copy; placement
// (which XCD, which CU) only changes speed.  Every spin is bounded (s_memrealtime): on a timeout
// the kernel flags sync[0] and runs to its end with garbage -- the host checks the flag at its
// next synchronisation and reports an error instead of hanging.
//
// The host launches this kernel only when the whole grid is resident at once (occupancy query x
// CUs); otherwise, and for paired inputs / other hops, the three-launch path stays.
#pragma once
#include "../../dsptoolbox_amd/csrc/kernels_welch4096w.hpp"

namespace welch4096 {

constexpr int F_MAX_UNITS = 768;                     // chunks <= 768, channels <= 768
constexpr int F_SYNC_WORDS = 16 + F_MAX_UNITS;       // [0] timeout code, [1] published, [2] through, [16 + q] xready
constexpr unsigned long long F_SPIN_TICKS = 30000000ull;  // 0.3 s of the 100 MHz s_memrealtime clock

#ifndef W4F_STAMPS
#define W4F_STAMPS 0  // dev only: per-workgroup s_memrealtime stamps -> FusedArgs::stamps[block][8]
#endif
#if W4F_STAMPS
#define W4F_STAMP(i)                                                                                   \
    do {                                                                                               \
        if (threadIdx.x == 0) fa.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define W4F_STAMP(i)
#endif

struct FusedArgs {
    Args a;
    unsigned* sync;  // F_SYNC_WORDS, zero before the first launch; the kernel leaves it zero
    int mode;        // DS_TF_H1 .. H3
    dsk::FinishPar fin;
    float2* tf;      // [NB][n_ch]
    float* coh;      // [NB][n_ch]
#if W4F_STAMPS
    unsigned long long* stamps;
#endif
};

__device__ __forceinline__ unsigned ld_agent(const unsigned* w) {
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one lane: wait until *w >= target (counters only grow inside a launch); false on a timeout.
// Hundreds of workgroups poll the same few words: the pause between two polls grows with the
// distance to the target (a first version polled every 0.1 us from 500-768 lanes at once and the
// memory channel of those words backed up -- the producers it was waiting for took 20-30 us
// instead of 6, and the last workgroups of the launch slowed down by as much).
__device__ __forceinline__ bool spin_until(const unsigned* w, unsigned target, unsigned* err, unsigned code) {
    unsigned seen = ld_agent(w);
    if (seen >= target) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned left = target - seen;
        if (left > 16)
            __builtin_amdgcn_s_sleep(48);  // ~1.5 us
        else if (left > 2)
            __builtin_amdgcn_s_sleep(16);  // ~0.5 us
        else
            __builtin_amdgcn_s_sleep(6);
        seen = ld_agent(w);
        if (seen >= target) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > F_SPIN_TICKS) {
            __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}
// write-through (sc1) stores: aux bit 4
__device__ __forceinline__ void st_wt_b128(float4 v, __amdgpu_buffer_rsrc_t r, int byte_off) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ void st_wt_b64(float2 v, __amdgpu_buffer_rsrc_t r, int byte_off) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ void st_wt_b32(float v, __amdgpu_buffer_rsrc_t r, int byte_off) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void tail(const FusedArgs& fa, float2* lds, float2 (&T)[16], float (&P)[16], int q, int c,
                                     int p0, int p1, int bpc, int tid);

__global__ __launch_bounds__(NT, 3) void k_h1f(FusedArgs fa) {
    const Args& p = fa.a;
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * L1S;
    float* winl = reinterpret_cast<float*>(lds + 16 * L1S + 256);
    const int tid = threadIdx.x;
    unsigned* const sy = fa.sync;
    int q, c;
    {
        const int b = blockIdx.x, total = p.n_chunks * p.n_ch;
        const int u = (total & 7) == 0 ? (b & 7) * (total >> 3) + (b >> 3) : b;
        q = u / p.n_ch;
        c = u - q * p.n_ch;
    }
    W4F_STAMP(0);
    // the pair loop runs at priority 3 ... 0 as a chunk gets done; the input-spectrum transforms in
    // front of it must not start below their neighbours' loops (at priority 0 they took 20-30 us)
    __builtin_amdgcn_s_setprio(3);
    Tw6 tw;
    load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) winl[tid + 256 * n1] = p.window[tid + 256 * n1];
    int p0, p1;
    chunk_range(p, q, p0, p1);
    // (scalars the tail needs, formed here: their division would otherwise be hoisted above the pair
    // loop as a VECTOR register and spilled across it; the same for the thread index, re-formed behind
    // the loop from the wave's index kept in a scalar register)
    const int bpc = __builtin_amdgcn_readfirstlane((NB + p.n_ch - 1) / p.n_ch);
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- 1. PRODUCE: input spectra of this chunk, one pair per workgroup c < pairs(q) ----------
    {
        int made = 0;
        for (int pr = p0 + c; pr < p1; pr += p.n_ch, ++made) {
            float2 v[16];
            {
                const __amdgpu_buffer_rsrc_t rs = channel_rsrc(p.xsig, p.n_samples);
                const int off0 = 4 * (2 * pr * 2048 + tid);
                float s[24];
#pragma unroll
                for (int m = 0; m < 24; ++m) s[m] = ld_sample(rs, off0 + 1024 * m);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const float w = winl[tid + 256 * n1];  // (written by this very thread above)
                    v[n1] = make_float2(s[n1] * w, s[n1 + 8] * w);
                }
                if (needs_drop(p, pr)) drop_second(v);
            }
            W4F_STAMP(8);
            fft4096_w(v, tw, buf, tw2, tid);
            W4F_STAMP(9);
            if (p.detrend && tid == 0) v[pos16(0)] = make_float2(0.f, 0.f);
            const __amdgpu_buffer_rsrc_t xo = rsrc_of(p.xs + (int64_t)pr * N, N * 8);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float2 z0 = v[pos16(2 * g)], z1 = v[pos16(2 * g + 1)];
                st_wt_b128(make_float4(z0.x, z0.y, z1.x, z1.y), xo, 16 * (tid + 256 * g));
            }
            float* pw = reinterpret_cast<float*>(buf);
            const int bt = bin_thread(tid);
            __syncthreads();
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) {
                const float2 z = v[pos16(k3)];
                pw[fold_pos(bt + 256 * k3)] = z.x * z.x + z.y * z.y;
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t po = rsrc_of(p.px + (int64_t)pr * NB, NB * 4);
            for (int k = tid; k < NB; k += NT)
                st_wt_b32(0.5f * (pw[fold_pos(k)] + pw[fold_pos((N - k) & (N - 1))]), po, 4 * k);
        }
        if (made) {  // (workgroup-uniform)
            W4F_STAMP(10);
            drain_stores();
            __syncthreads();
            W4F_STAMP(11);
            if (tid == 0)
                __hip_atomic_fetch_add(&sy[16 + q], (unsigned)made, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    W4F_STAMP(1);
    // ---- 2. MAIN: the pair loop of k_y3 -----------------------------------------------------------
    float2 T[16];
    float P[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        T[j] = make_float2(0.f, 0.f);
        P[j] = 0.f;
    }
    float carry[8], nx[16];
    const float* ch = p.sig + (int64_t)c * p.ld;
    const __amdgpu_buffer_rsrc_t rs = channel_rsrc(ch, p.n_samples);
    const __amdgpu_buffer_rsrc_t xrs = rsrc_of(p.xs + (int64_t)p0 * N, (uint32_t)((p1 - p0) * (N * 8)));
    if (p0 < p1) {
        const int off0 = 4 * (2 * p0 * 2048 + tid);
#pragma unroll
        for (int j = 0; j < 8; ++j) carry[j] = ld_sample(rs, off0 + 1024 * j);
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = ld_sample(rs, off0 + 1024 * (8 + j));
    }
    Stamp ts;
    float winr[16];
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
            ts, level16,
            [&]() {
                // the chunk's input spectra: polled once, in front of their first use
                if (pr == p0) {
                    W4F_STAMP(2);
                    if (tid == 0) {
                        spin_until(&sy[16 + q], (unsigned)(p1 - p0), &sy[0], 1u);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __syncthreads();
                    W4F_STAMP(3);
                }
            });
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
    W4F_STAMP(4);
    __builtin_amdgcn_s_setprio(3);  // what follows is the tail of the launch
    return tail(fa, lds, T, P, q, c, p0, p1, bpc,
                (wave_s << 6) | (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
}

// steps 3 and 4 of k_h1f (its own function only to give `tid` a new name: see wave_s above)
__device__ __forceinline__ void tail(const FusedArgs& fa, float2* lds, float2 (&T)[16], float (&P)[16], int q, int c,
                                     int p0, int p1, int bpc, int tid) {
    const Args& p = fa.a;
    float2* buf = lds;
    unsigned* const sy = fa.sync;
    if (p.detrend && tid == 0) P[0] = 0.f;

    // ---- 3. PARTIALS ----------------------------------------------------------------------------
    __syncthreads();
    {
        // input auto spectrum of this chunk: this workgroup's slice of the bins over the chunk's px
        // rows, fp64, one sweep of `width` bins x 256 / width row groups
        double* red = reinterpret_cast<double*>(lds);
        const int b0 = c * bpc, b1 = min(b0 + bpc, NB);
        const int lw = bpc <= 32 ? 5 : (bpc <= 64 ? 6 : (bpc <= 128 ? 7 : 8)), width = 1 << lw, rows = NT >> lw;
        const int rg = tid >> lw, kl = tid & (width - 1);
        const __amdgpu_buffer_rsrc_t so = rsrc_of(p.psx + (int64_t)q * NB, NB * 4);
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
                st_wt_b32((float)t, so, 4 * k);
            }
            __syncthreads();
        }
    }
    const int bt = bin_thread(tid);
    const int64_t so = ((int64_t)q * p.n_ch + c) * NB;
    {
        const __amdgpu_buffer_rsrc_t rxy = rsrc_of(p.pxy + so, NB * 8), ryy = rsrc_of(p.pyy + so, NB * 4);
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) buf[fold_pos(bt + 256 * k3)] = T[k3];
        __syncthreads();
        for (int k = tid; k < NB; k += NT) {
            const float2 a = buf[fold_pos(k)], b = buf[fold_pos((N - k) & (N - 1))];
            st_wt_b64(make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y)), rxy, 8 * k);
        }
        __syncthreads();
        float* pw = reinterpret_cast<float*>(buf);
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) pw[fold_pos(bt + 256 * k3)] = P[k3];
        __syncthreads();
        for (int k = tid; k < NB; k += NT)
            st_wt_b32(0.5f * (pw[fold_pos(k)] + pw[fold_pos((N - k) & (N - 1))]), ryy, 4 * k);
    }
    drain_stores();
    __syncthreads();
    W4F_STAMP(5);
    // ---- 4. FINISH: slice q of channel c's bins, once every workgroup of the grid has published -------
    // (the input auto spectrum of a bin comes from the workgroups of ANOTHER channel index, so the
    // wait is for the whole grid: one counter; nobody can end before the slowest workgroup anyway, and
    // behind the wait all 768 workgroups share the finish -- one round of loads each)
    const unsigned grid = (unsigned)(p.n_chunks * p.n_ch);
    if (tid == 0) {
        __hip_atomic_fetch_add(&sy[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(&sy[1], grid, &sy[0], 2u);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    W4F_STAMP(6);
    {
        const int bps = (NB + p.n_chunks - 1) / p.n_chunks;       // bins per slice
        const int k0 = q * bps, k1 = min(k0 + bps, NB);
        int lw = 0;                                                 // 2^lw bins side by side, 256 >> lw chunk groups
        while ((1 << lw) < bps && lw < 8) ++lw;
        const int width = 1 << lw, groups = NT >> lw;
        const int kl = tid & (width - 1), g = tid >> lw;
        const int64_t sq = (int64_t)p.n_ch * NB;
        const float2* __restrict__ pxy = p.pxy + (int64_t)c * NB;
        const float* __restrict__ pyy = p.pyy + (int64_t)c * NB;
        const float* __restrict__ psx = p.psx;
        double* red = reinterpret_cast<double*>(lds);  // [4][256]
        for (int kb = k0; kb < k1; kb += width) {
            const int k = kb + kl;
            const bool live = k < k1;
            double sxx = 0.0, syy = 0.0, sxr = 0.0, sxi = 0.0;
            if (live) {
                // up to 16 chunks (48 loads) in flight per thread: one round trip for the usual grids
                for (int base = g; base < p.n_chunks; base += 16 * groups) {
                    float2 t[16];
                    float yy[16], xx[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int qq = min(base + j * groups, p.n_chunks - 1);
                        t[j] = pxy[qq * sq + k];
                        yy[j] = pyy[qq * sq + k];
                        xx[j] = psx[(int64_t)qq * NB + k];
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (base + j * groups < p.n_chunks) {
                            sxr += (double)t[j].x;
                            sxi += (double)t[j].y;
                            syy += (double)yy[j];
                            sxx += (double)xx[j];
                        }
                    }
                }
            }
            if (groups > 1) {
                red[tid] = sxx;
                red[256 + tid] = sxr;
                red[512 + tid] = sxi;
                red[768 + tid] = syy;
                __syncthreads();
                if (g == 0) {
                    for (int j = 1; j < groups; ++j) {
                        sxx += red[j * width + kl];
                        sxr += red[256 + j * width + kl];
                        sxi += red[512 + j * width + kl];
                        syy += red[768 + j * width + kl];
                    }
                }
                __syncthreads();
            }
            if (live && g == 0) {
                dsk::cd sxy{sxr, sxi + 0.0};  // + 0.0: a sum of -0 partials becomes +0 like the reference's mean
                dsk::tf_from_sums(sxx, sxy, syy, k, fa.mode, fa.fin, fa.tf[(int64_t)k * p.n_ch + c],
                                  fa.coh[(int64_t)k * p.n_ch + c]);
            }
        }
    }
    W4F_STAMP(7);
    // the last workgroup through: everybody has seen the counters -> back to zero for the next launch
    unsigned* const flag = reinterpret_cast<unsigned*>(lds) + 4096;
    __syncthreads();
    if (tid == 0) flag[0] = __hip_atomic_fetch_add(&sy[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag[0] != grid - 1) return;
    for (int i = tid; i < p.n_chunks; i += NT)
        __hip_atomic_store(&sy[16 + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        __hip_atomic_store(&sy[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sy[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// EXPERIMENT ONLY since round 4 (it lived behind DSPTOOLBOX_AMD_W4_ONE_LAUNCH in the library in round 3).
// Measured on MI355X, 64 x 2^20 samples (tools/exp/exp_fused.hip, per-workgroup s_memrealtime stamps,
// profiles/r03_welch_one_launch_stamps.txt): the one launch is correct (bit-identical to the three
// launches, also beside a competing kernel and over repeated launches) but SLOWER, 165-190 us against
// 107-110 us, and a kernel whose workgroups wait for each other stalls behind any other stream that
// holds CUs -- so it is not shipped.  The harness needs every workgroup resident at once.

}  // namespace welch4096
