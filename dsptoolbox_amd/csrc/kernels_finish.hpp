// Epilogue kernels: chunk partials -> the reference's finish() (scaling, edge
// halving, principal square root; standard/_spectral_methods.py:151-171) in
// fp64, transfer functions and coherence
// (transfer_functions/transfer_functions.py:525-534), and the channel x channel
// cross-spectral-matrix GEMM on the fp32 MFMA pipe.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "size_guards.hpp"

namespace dsk {

struct cd {
    double x, y;
};

__device__ __forceinline__ cd csqrt_principal(cd z) {
    double r = sqrt(z.x * z.x + z.y * z.y);  // (no overflow protection needed at these magnitudes)
    if (r == 0.0) return cd{0.0, z.y};
    if (z.x >= 0.0) {
        double t = sqrt(0.5 * (r + z.x));
        return cd{t, z.y / (2.0 * t)};
    }
    double t = sqrt(0.5 * (r - z.x));
    return cd{fabs(z.y) / (2.0 * t), copysign(t, z.y)};
}

struct FinishPar {
    double inv;     // norm_scale / n_frames
    double factor;  // physical-unit factor (applied when halve_edges)
    int halve_edges, amp_sqrt, nb;
};

__device__ __forceinline__ double finish_real(double s, int b, const FinishPar& f) {
    s *= f.inv;
    if (f.halve_edges) {
        s *= f.factor;
        if (b == 0 || b == f.nb - 1) s *= 0.5;
    }
    return f.amp_sqrt ? sqrt(s) : s;
}
__device__ __forceinline__ cd finish_cplx(cd s, int b, const FinishPar& f) {
    s.x *= f.inv;
    s.y *= f.inv;
    if (f.halve_edges) {
        double e = (b == 0 || b == f.nb - 1) ? 0.5 * f.factor : f.factor;
        s.x *= e;
        s.y *= e;
    }
    return f.amp_sqrt ? csqrt_principal(s) : s;
}

// frame sums -> H1 / H2 / H3 and coherence (transfer_functions.py:525-534)
__device__ __forceinline__ void tf_from_sums(double sxx, cd sxy, double syy, int b, int mode,
                                             const FinishPar& fin, float2& tf, float& coh) {
    cd gxy = finish_cplx(sxy, b, fin);
    double gxx = finish_real(sxx, b, fin), gyy = finish_real(syy, b, fin);
    double axy2 = gxy.x * gxy.x + gxy.y * gxy.y;
    cd h;
    if (mode == 1) {  // H1 = Gxy / Gxx
        h = cd{gxy.x / gxx, gxy.y / gxx};
    } else if (mode == 2) {  // H2 = Gyy / Gyx, Gyx = finish(conj(Sxy)) = conj(Gxy) ...
        // ... except where Sxy is exactly real and negative (DC / Nyquist bins): the
        // reference takes the principal root of (-a + 0j) for Gxy AND for Gyx.
        cd gyx = (sxy.y == 0.0) ? gxy : cd{gxy.x, -gxy.y};
        h = cd{gyy * gyx.x / axy2, -gyy * gyx.y / axy2};  // Gyy / Gyx = Gyy conj(Gyx)/|Gyx|^2
    } else {  // H3 = Gxy/|Gxy| * sqrt(Gyy/Gxx)
        double s = sqrt(gyy / gxx) / sqrt(axy2);
        h = cd{gxy.x * s, gxy.y * s};
    }
    tf = make_float2((float)h.x, (float)h.y);
    coh = (float)(axy2 / gxx / gyy);
}

// kind: 0 = transfer function + coherence, 1 = auto spectra (psd), 2 = cross spectra (csd)
struct WelchFinArgs {
    const float* pxx;   // [q][n_cx][nb]
    const float2* pxy;  // [q][n_cy][nb]
    const float* pyy;   // [q][n_cy][nb]
    int n_chunks, n_chunks_x, n_cx, n_cy, kind, mode;
    FinishPar fin;
    float2* tf;  // [nb][n_cy]   (kind 0: tf, kind 2: csd)
    float* coh;  // [nb][n_cy]   (kind 0: coherence, kind 1: psd [nb][n_cx])
    // windows shorter than the transform that ran (128 / 64 / 32 samples on the 256-point kernels): the partial rows hold
    // in_nb bins of which every in_step-th is one of the fin.nb output bins (0: rows of fin.nb bins, every one)
    int in_nb = 0, in_step = 1;
    int force_wide = 0;  // 64-bit loads whatever the slab size (launch_finish, api.hip)
};

// block = 256 threads = 64 output values x 4 waves; wave s sums the chunks q = s (mod 4) in fp64,
// the four partial sums are combined through LDS and wave 0 finishes.  (One thread per output
// value over all chunks is a latency chain on a quarter of the CUs: 33 us for 48 chunks of the
// 1024-point path, against 13 us this way.)  grid = ceil(nb * nc / 64).
__global__ __launch_bounds__(256) void k_welch_finish(WelchFinArgs p) {
    __shared__ double red[3][4][64];
    const int nb = p.fin.nb;
    const int nc = p.kind == 1 ? p.n_cx : p.n_cy;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    // bins vary fastest across lanes: the partial slabs [q][c][b] are read coalesced
    // (the (b, c)-ordered outputs are 64x smaller than the slabs)
    const int64_t tix = (int64_t)blockIdx.x * 64 + lane;
    const bool live = tix < (int64_t)nb * nc;
    const int c = live ? (int)(tix / nb) : 0, b = live ? (int)(tix % nb) : 0;
    const int64_t idx = (int64_t)b * nc + c;
    const int cx = p.n_cx == 1 ? 0 : c;
    double sxx = 0.0, syy = 0.0;
    cd sxy{0.0, 0.0};
    const float* __restrict__ pxx = p.pxx;
    const float2* __restrict__ pxy = p.pxy;
    const float* __restrict__ pyy = p.pyy;
    const int inb = p.in_nb > 0 ? p.in_nb : nb;  // bins per partial row
    const int64_t bi = (int64_t)b * p.in_step;    // this output bin within it
    // A wave's chunks q = sub, sub + 4, ... in batches of four, every load of a batch requested before the first sum
    // (raw-buffer loads: a chunk past the last one, or a dead lane, is an offset behind the end and reads as zero).
    // As a rolled loop of plain loads this was one memory round trip per chunk -- three in a row for the headline's
    // twelve chunks, most of the kernel's 8 us.
    {
        constexpr int U = 4;
        const int64_t sx = (int64_t)p.n_cx * inb, ox = (int64_t)cx * inb + bi;
        const int64_t sy = (int64_t)p.n_cy * inb, oy = (int64_t)c * inb + bi;
        const bool want_yy = p.kind == 0;
        // one descriptor per chunk slab (the chunk index is wave-uniform): 32-bit offsets stay inside ONE slab of
        // channels x bins, whatever the number of chunks
        const int subu = __builtin_amdgcn_readfirstlane(sub);
        const uint32_t ex = live ? (uint32_t)ox : 0x3ffffffcu, ey = live ? (uint32_t)oy : 0x1ffffffeu;
        const int n_max = p.n_chunks_x > p.n_chunks ? p.n_chunks_x : p.n_chunks;
        // The descriptors above hold a slab's byte size in 32 bits and the lanes' offsets are 32-bit byte offsets: a slab
        // of 4 GiB or more (windows of 2^23 / 2^24 samples with 128 / 64 channels on the four-step path) would wrap and
        // read zeros.  Those slabs take plain 64-bit loads, one chunk at a time (welch_finish_wide_slab in tests/host_san).
        if (p.force_wide || welch_finish_wide_slab(sx, sy)) {
            for (int q = subu; q < n_max; q += 4) {
                if (!live) continue;
                if (p.kind != 2 && q < p.n_chunks_x) sxx += (double)pxx[(int64_t)q * sx + ox];
                if (p.kind != 1 && q < p.n_chunks) {
                    const float2 v = pxy[(int64_t)q * sy + oy];
                    sxy.x += (double)v.x;
                    sxy.y += (double)v.y;
                    if (want_yy) syy += (double)pyy[(int64_t)q * sy + oy];
                }
            }
        } else
        for (int q0 = subu; q0 < n_max; q0 += 4 * U) {
            float vx[U], vy[U];
            float2 vxy[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + 4 * u;
                const bool hx = p.kind != 2 && q < p.n_chunks_x, hy = p.kind != 1 && q < p.n_chunks;
                const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(pxx) + (hx ? q * sx : 0), 0, hx ? (int)(uint32_t)(sx * 4) : 0, 0x00020000);
                const __amdgpu_buffer_rsrc_t rxy = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float2*>(pxy) + (hy ? q * sy : 0), 0, hy ? (int)(uint32_t)(sy * 8) : 0, 0x00020000);
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(pyy) + ((hy && want_yy) ? q * sy : 0), 0, (hy && want_yy) ? (int)(uint32_t)(sy * 4) : 0, 0x00020000);
                vx[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(ex * 4u), 0, 0));
                vxy[u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rxy, (int)(ey * 8u), 0, 0));
                vy[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, (int)(ey * 4u), 0, 0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                sxx += (double)vx[u];
                sxy.x += (double)vxy[u].x;
                sxy.y += (double)vxy[u].y;
                syy += (double)vy[u];
            }
        }
    }
    red[0][sub][lane] = sxx;
    red[1][sub][lane] = sxy.x;
    red[2][sub][lane] = sxy.y;
    __syncthreads();
    const double sxy_x = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
    // + 0.0: a sum of -0 partials becomes +0 like the reference's mean
    const double sxy_y = red[2][0][lane] + red[2][1][lane] + red[2][2][lane] + red[2][3][lane] + 0.0;
    sxx = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    __syncthreads();
    red[0][sub][lane] = syy;
    __syncthreads();
    if (sub != 0 || !live) return;
    syy = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    sxy = cd{sxy_x, sxy_y};
    if (p.kind == 1) {
        p.coh[idx] = (float)finish_real(sxx, b, p.fin);
        return;
    }
    if (p.kind == 2) {
        cd gxy = finish_cplx(sxy, b, p.fin);
        p.tf[idx] = make_float2((float)gxy.x, (float)gxy.y);
        return;
    }
    tf_from_sums(sxx, sxy, syy, b, p.mode, p.fin, p.tf[idx], p.coh[idx]);
}

// Median over frames (average="median", standard/_spectral_methods.py:153-162): per (channel, bin)
// the median of the per-frame auto power |X_f|^2, or of the real and imaginary parts of the cross
// power conj(X_f) Y_f, from the stored frame spectra xs[cx][f][b], ys[c][f][b].  One block = one
// channel x 8 bins; the series sit in LDS and are sorted there (one bitonic network over all series of the
// workgroup).  Results go into the
// one-chunk partial layout k_welch_finish reads.
struct MedianArgs {
    const float2* xs;  // [n_cx][F][nb]
    const float2* ys;  // [n_cy][F][nb] or nullptr (auto spectra of xs only)
    int n_cx, n_cy, n_frames, nb, kind;  // kind as in WelchFinArgs
    int bpb;      // bins per workgroup: 8, 4, 2 or 1 (what fits the LDS for this frame count)
    float* pxx;   // [n_cx][nb]
    float2* pxy;  // [n_cy][nb]
    float* pyy;   // [n_cy][nb]
};

// All `nser` series of the workgroup (stride P = the frame count rounded up to a power of two, the tail filled with
// +inf) are sorted at once by one bitonic network in LDS: P log2(P) (log2(P) + 1) / 4 compare-exchanges per series
// instead of the F^2 comparisons of ranking every element against all others (2048 frames: 68 k against 4.2 M).
// The two middle elements of the F sorted values go to out2[2 q], out2[2 q + 1].
__host__ __device__ inline int median_stride(int F) {
    int P = 1;
    while (P < F) P <<= 1;
    return P < 2 ? 2 : P;
}
__device__ __forceinline__ void sort_series(float* ser, int nser, int P, int tid, int nt) {
    const int half = P >> 1, lh = __ffs(half) - 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = tid; idx < nser * half; idx += nt) {
                const int q = idx >> lh, t = idx & (half - 1);
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                float* s = ser + (size_t)q * P;
                const float a = s[i], b = s[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    s[i] = b;
                    s[l] = a;
                }
            }
            __syncthreads();
        }
    }
}
__device__ __forceinline__ void middle_of_sorted(const float* ser, int nser, int P, int F, int tid, float* out2) {
    if (tid < nser) {
        out2[2 * tid] = ser[(size_t)tid * P + (F - 1) / 2];
        out2[2 * tid + 1] = ser[(size_t)tid * P + F / 2];
    }
}

__global__ __launch_bounds__(256) void k_welch_median(MedianArgs p) {
    extern __shared__ float ser[];  // [bpb bins][3 series][P]  (+ bpb*3*2 results)
    const int F = p.n_frames, nb = p.nb, P = median_stride(F);
    const int BP = p.bpb, lb = __ffs(BP) - 1;
    const int b0 = blockIdx.x * BP, c = blockIdx.y;
    const int tid = threadIdx.x;
    const bool have_y = p.ys != nullptr;
    const int cx = p.n_cx == 1 ? 0 : c;
    float* res = ser + (size_t)BP * 3 * P;  // [bpb][3][2]
    // series 0: |X|^2 (kind 0: of xs[cx]; kind 1: of xs[c]); 1: Re conj(X) Y; 2: Im conj(X) Y; for kind 0
    // |Y|^2 replaces series 0 in a second sweep below
    const float2* X = p.xs + (size_t)(p.kind == 1 ? c : cx) * F * nb;
    const float2* Y = have_y ? p.ys + (size_t)c * F * nb : nullptr;
    const float inf = __builtin_inff();
    for (int i = tid; i < BP * P; i += 256) {
        const int bl = i & (BP - 1), f = i >> lb, b = b0 + bl;
        float s0 = inf, s1 = inf, s2 = inf;
        if (f < F) {
            float2 xv = make_float2(0.f, 0.f), yv = make_float2(0.f, 0.f);
            if (b < nb) {
                xv = X[(size_t)f * nb + b];
                if (have_y) yv = Y[(size_t)f * nb + b];
            }
            s0 = xv.x * xv.x + xv.y * xv.y;
            s1 = xv.x * yv.x + xv.y * yv.y;  // Re conj(x) y
            s2 = xv.x * yv.y - xv.y * yv.x;  // Im conj(x) y
        }
        ser[(bl * 3 + 0) * P + f] = s0;
        ser[(bl * 3 + 1) * P + f] = s1;
        ser[(bl * 3 + 2) * P + f] = s2;
    }
    __syncthreads();
    // (without an output channel only the first of the three series of a bin is used: sorting all three keeps the
    // layout uniform and the other two are constant)
    sort_series(ser, BP * 3, P, tid, 256);
    middle_of_sorted(ser, BP * 3, P, F, tid, res);
    __syncthreads();
    if (tid < BP && b0 + tid < nb) {
        const int b = b0 + tid;
        const float* r = res + tid * 6;
        const float mxx = 0.5f * (r[0] + r[1]);
        if (p.kind == 1) {
            p.pxx[(size_t)c * nb + b] = mxx;
        } else {
            if (p.n_cx == 1 ? c == 0 : true) p.pxx[(size_t)cx * nb + b] = mxx;
            p.pxy[(size_t)c * nb + b] = make_float2(0.5f * (r[2] + r[3]), 0.5f * (r[4] + r[5]));
        }
    }
    if (p.kind == 0) {  // |Y|^2 series
        __syncthreads();
        for (int i = tid; i < BP * P; i += 256) {  // BP bins per workgroup here too (LDS holds BP series)
            const int bl = i & (BP - 1), f = i >> lb, b = b0 + bl;
            float s0 = inf;
            if (f < F) {
                const float2 yv = b < nb ? Y[(size_t)f * nb + b] : make_float2(0.f, 0.f);
                s0 = yv.x * yv.x + yv.y * yv.y;
            }
            ser[(size_t)bl * P + f] = s0;
        }
        __syncthreads();
        sort_series(ser, BP, P, tid, 256);
        middle_of_sorted(ser, BP, P, F, tid, res);
        __syncthreads();
        if (tid < BP && b0 + tid < nb) p.pyy[(size_t)c * nb + b0 + tid] = 0.5f * (res[tid * 2] + res[tid * 2 + 1]);
    }
}

struct CsmMedianArgs {
    const float2* xs;
    int n_ch, n_frames;
    int bpb;        // bins per workgroup: 8, 4, 2 or 1
    FinishPar fin;  // fin.scale already holds norm_scale * n_bias
    float2* csm;    // [nb][C][C]
};
__global__ __launch_bounds__(256) void k_csm_median(CsmMedianArgs p) {
    extern __shared__ float ser[];  // [bpb bins][2 series][P] (+ bpb*2*2 results)
    const int F = p.n_frames, nb = p.fin.nb, C = p.n_ch, P = median_stride(F);
    const int BP = p.bpb, lb = __ffs(BP) - 1;
    const int b0 = blockIdx.x * BP, tid = threadIdx.x;
    // triangular decode: pair index -> (i1 <= i2), rows of length C, C-1, ...
    int i1 = 0, rem = blockIdx.y;
    while (rem >= C - i1) {
        rem -= C - i1;
        ++i1;
    }
    const int i2 = i1 + rem;
    float* res = ser + (size_t)BP * 2 * P;
    const float2* X = p.xs + (size_t)i1 * F * nb;
    const float2* Y = p.xs + (size_t)i2 * F * nb;
    const float inf = __builtin_inff();
    for (int i = tid; i < BP * P; i += 256) {
        const int bl = i & (BP - 1), f = i >> lb, b = b0 + bl;
        float s0 = inf, s1 = inf;
        if (f < F) {
            float2 xv = make_float2(0.f, 0.f), yv = make_float2(0.f, 0.f);
            if (b < nb) {
                xv = X[(size_t)f * nb + b];
                yv = Y[(size_t)f * nb + b];
            }
            s0 = xv.x * yv.x + xv.y * yv.y;
            s1 = xv.x * yv.y - xv.y * yv.x;
        }
        ser[(bl * 2 + 0) * P + f] = s0;
        ser[(bl * 2 + 1) * P + f] = s1;
    }
    __syncthreads();
    sort_series(ser, 2 * BP, P, tid, 256);
    middle_of_sorted(ser, 2 * BP, P, F, tid, res);
    __syncthreads();
    if (tid < BP && b0 + tid < nb) {
        const int b = b0 + tid;
        const float* r = res + tid * 4;
        cd m{0.5 * ((double)r[0] + (double)r[1]), 0.5 * ((double)r[2] + (double)r[3]) + 0.0};
        cd g = finish_cplx(m, b, p.fin);
        float2* o = p.csm + (size_t)b * C * C;
        if (i1 == i2) {
            o[(size_t)i1 * C + i1] = make_float2((float)g.x, 0.f);
        } else {
            o[(size_t)i2 * C + i1] = make_float2((float)g.x, (float)g.y);
            o[(size_t)i1 * C + i2] = make_float2((float)g.x, (float)-g.y);
        }
    }
}

// r[c][b] = eps ? conj(X)/(|X|^2 + eps[b]) : 1/X ; xspec is [b][c]
__global__ void k_deconv_inverse(const float2* xspec, int n_ch, int nb, const float* eps,
                                 float2* r) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nb * n_ch) return;
    int c = (int)(idx / nb), b = (int)(idx % nb);
    float2 x = xspec[(int64_t)b * n_ch + c];
    double xr = x.x, xi = x.y;
    double den = xr * xr + xi * xi + (eps ? (double)eps[b] : 0.0);
    r[idx] = make_float2((float)(xr / den), (float)(-xi / den));
}

// ---------------------------------------------------------------- CSM GEMM (fp32 MFMA)
// X[b][f][c] (the STFT layout).  grid = (nb, n_tile_pairs), 256 threads.
// Tile pair (I >= J) of 32 x 32 channels: G[i][j] = sum_f X_i conj(X_j) with
//   Re += Xr_i Xr_j + Xi_i Xi_j ,  Im += Xi_i Xr_j - Xr_i Xi_j
// as four v_mfma_f32_32x32x2_f32 per two frames; the 4 waves split the frames
// and are combined in fp64 through LDS, followed by finish() and the
// reference's mirror  csm[b][i2][i1] = g, csm[b][i1][i2] = conj(g)  (i2 >= i1).
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct CsmArgs {
    const float2* X;
    int n_ch, n_frames;
    FinishPar fin;
    float2* csm;  // matrix of bin b0 first
    int b0;       // first bin of this call (generic kernel: a bin range for the multi-GPU split)
    int n_groups; // k_csm_group_b3 / k_csm_offdiag_b3: groups of 64 channels (0: not used)
    int n_groups_bins;  // k_csm_offdiag_b3: bins of this call (its grid is padded to whole XCD rounds)
    // Frame chunks (round 5: the transform of chunk k + 1 runs beside the product of chunk k on a second stream): the
    // raw fp32 sums of the lower triangle, [bin][i][j] with the matrix's own strides.  part_in: added to this call's
    // sums; part_out: this call stores its sums there INSTEAD of finishing them (the last chunk has part_out == null).
    const float2* part_in = nullptr;
    float2* part_out = nullptr;
};

// Combine the four waves' partial 32x32 tiles (fp64, through LDS), apply the Welch finish and
// store the tile and its conjugate mirror.  All 256 threads call these together.
//   diag_m (diagonal tiles, I == J, of the one-workgroup-per-bin kernels): `im` holds only
//   M = Xi Xr^T; the imaginary part of the Hermitian tile is M - M^T (Im = Xi Xr^T - Xr Xi^T),
//   formed here from the element and its mirror -- one matrix product per k-step less than
//   accumulating both.
//   ILV (k_csm_gemm64_b3): tile T holds the channels 2 r + T (even / odd) instead of 32 T + r; the
//   off-diagonal tile then has elements on both sides of the diagonal, and an upper one is stored
//   as the conjugate of the lower element it mirrors (finish() is applied to the lower one, as
//   everywhere).
// Code size matters here: the fp64 finish (division, complex square root) is some hundred
// instructions per element, and a kernel that inlines it per element, tile and bin kind (round 2's
// first form: 17 000 instructions, 135 KB against a 64 KB instruction cache) spends more time
// fetching its epilogue than running its matrix instructions.  So: tile kind as run-time
// arguments, one call site per tile and kernel, and the finish written once, branch-free, for the
// thread's four elements (csm_tile_reduce_store).
// the four waves' partial tiles [wave][re / im][register][lane]; rows of 65 so that the transposed
// reads of the diagonal tiles (register index varying across the lanes) fall into different banks
typedef float CsmRed[4][2][16][65];

__device__ __forceinline__ void csm_tile_put(CsmRed& red, const f32x16& re, const f32x16& im) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[w][0][r][l] = re[r];
        red[w][1][r][l] = im[r];
    }
}

// STAGED: the finished elements go into an LDS image G of the C x C matrix (row stride C + 1, so
// that the mirror writes of a wave fall into different banks) and leave as whole rows afterwards:
// stored straight from here, the mirror elements of a wave are 64 separate 8-byte writes C * 8 bytes
// apart, and the 3 M such requests of the 64-channel shape keep the L2 channels busy for 16 us.
template <bool ILV = false, bool STAGED = false>
__device__ __forceinline__ void csm_tile_reduce_store(CsmRed& red, int I, int J, bool diag_m, int b,
                                                      const CsmArgs& p, float2* G = nullptr, int c_local = -1) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int C = c_local < 0 ? p.n_ch : c_local, F = p.n_frames;  // c_local: channels of the group staged in G
    const int ldo = STAGED ? C + 1 : C;
    float2* out = STAGED ? G : p.csm + (int64_t)(b - p.b0) * C * C;
    // The chain LDS read -> sum -> fp64 scale -> complex square root -> store is ~800 cycles of
    // dependent latency per element and there are only two waves per SIMD to hide it, so the four
    // elements of a thread run side by side in branch-free code (selects instead of the branches of
    // finish_real / finish_cplx / csqrt_principal, same expressions, same results).
    float gx[4], gy[4];
    int gi[4], gj[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = w * 4 + rr;
        // the four partial sums (250 frames each, accumulated in fp32 by the matrix instructions) are
        // combined in fp32 as well: two more roundings of 6e-8; the finish below stays in fp64
        gx[rr] = (red[0][0][r][l] + red[1][0][r][l]) + (red[2][0][r][l] + red[3][0][r][l]);
        gy[rr] = (red[0][1][r][l] + red[1][1][r][l]) + (red[2][1][r][l] + red[3][1][r][l]);
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
        if (diag_m) {  // element (j, i) of the tile sits in register rm, lane lm (uniform branch)
            const int rm = (j & 3) + 4 * (j >> 3), lm = i + 32 * ((j >> 2) & 1);
            gy[rr] -= (red[0][1][rm][lm] + red[1][1][rm][lm]) + (red[2][1][rm][lm] + red[3][1][rm][lm]);
        }
        const int ti = ILV ? 2 * i + I : 32 * I + i, tj = ILV ? 2 * j + J : 32 * J + j;
        const bool up = ILV && I != J && ti < tj;  // G[ti][tj] = conj(G[tj][ti])
        gi[rr] = up ? tj : ti;
        gj[rr] = up ? ti : tj;
        // 0 - y, not -y: at the purely real bins y is +0 and must stay +0 -- the sign of a zero imaginary
        // part selects the branch of the square root of a negative real element (+i for the lower
        // element, its conjugate for the mirror, as the reference's sqrt of its Hermitian matrix gives)
        gy[rr] = up ? 0.f - gy[rr] : gy[rr];
    }
    if (p.part_in || p.part_out) {  // (uniform) frame chunks: carry the raw sums between the calls
        const int64_t pb = (int64_t)(b - p.b0) * p.n_ch * p.n_ch;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const bool live = gi[rr] < C && gj[rr] < C && gi[rr] >= gj[rr];
            const int64_t at = pb + (int64_t)gi[rr] * p.n_ch + gj[rr];
            if (p.part_in && live) {
                const float2 q = p.part_in[at];
                gx[rr] += q.x;
                gy[rr] += q.y;
            }
            if (p.part_out && live) p.part_out[at] = make_float2(gx[rr], gy[rr]);
        }
        if (p.part_out) return;
    }
    const double e = p.fin.halve_edges ? ((b == 0 || b == p.fin.nb - 1) ? 0.5 * p.fin.factor : p.fin.factor) : 1.0;
    double vx[4], vy[4];
    bool keep_sign[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const bool diag = gi[rr] == gj[rr];
        // one frame (_csm_fft) at a purely real bin (DC / Nyquist): numpy's `csm[[0, -1]] /= 2.0`
        // (complex / real) turns every -0 imaginary part of a negative real element into +0, so BOTH
        // mirror elements take the +i branch of the square root (the matrix is not Hermitian there)
        keep_sign[rr] = F == 1 && gy[rr] == 0.f;
        const double y0 = (diag || keep_sign[rr]) ? 0.0 : (double)gy[rr];
        vx[rr] = (double)gx[rr] * p.fin.inv * e;  // (x inv) e as in finish_cplx; e = 1 is exact
        vy[rr] = y0 * p.fin.inv * e;
    }
    if (p.fin.amp_sqrt) {  // principal square root (csqrt_principal), all four at once
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const double x = vx[rr], y = vy[rr];
            const double r = sqrt(x * x + y * y);
            const double t = sqrt(0.5 * (r + fabs(x)));
            const double q = fabs(y) / (2.0 * t);
            const bool pos = x >= 0.0, zero = r == 0.0;
            vx[rr] = zero ? 0.0 : (pos ? t : q);
            vy[rr] = zero ? y : copysign(pos ? q : t, y);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        if (gi[rr] < C && gj[rr] < C && gi[rr] >= gj[rr]) {
            const float fx = (float)vx[rr], fy = (float)vy[rr];
            out[(int64_t)gi[rr] * ldo + gj[rr]] = make_float2(fx, fy);
            // sqrt(x + 0i) = conj(sqrt(x - 0i)) everywhere but on the negative real axis, where
            // keep_sign leaves the +i branch for the mirror as well
            if (gi[rr] != gj[rr]) out[(int64_t)gj[rr] * ldo + gi[rr]] = make_float2(fx, keep_sign[rr] ? fy : -fy);
        }
    }
}

template <bool DIAG_M = false>
__device__ __forceinline__ void csm_tile_epilogue(CsmRed& red, const f32x16& re, const f32x16& im,
                                                  int I, int J, int b, const CsmArgs& p) {
    csm_tile_put(red, re, im);
    __syncthreads();
    csm_tile_reduce_store<false>(red, I, J, DIAG_M, b, p);
}

// the three tile pairs of a one-workgroup-per-bin kernel (C <= 64), the matrix staged in G[64 * 65]
// and written as whole rows
constexpr int CSM64_G = 64 * 65;
template <bool ILV>
__device__ __forceinline__ void csm_epilogue64(CsmRed& red, float2* G, const f32x16& re00,
                                               const f32x16& im00, const f32x16& re10, const f32x16& im10,
                                               const f32x16& re11, const f32x16& im11, bool diag_m, int b,
                                               const CsmArgs& p, int c0 = 0, int c_group = -1) {
    // c0, c_group: a group of channels [c0, c0 + c_group) of a wider matrix (k_csm_group_b3)
    const int C = c_group < 0 ? p.n_ch : c_group, Ct = p.n_ch;
    // (a rolled loop over the tiles makes hipcc select the accumulators through scratch memory)
    csm_tile_put(red, re00, im00);
    __syncthreads();
    csm_tile_reduce_store<ILV, true>(red, 0, 0, diag_m, b, p, G, C);
    if (ILV || C > 32) {
        __syncthreads();  // red is reused
        csm_tile_put(red, re10, im10);
        __syncthreads();
        csm_tile_reduce_store<ILV, true>(red, 1, 0, false, b, p, G, C);
        __syncthreads();
        csm_tile_put(red, re11, im11);
        __syncthreads();
        csm_tile_reduce_store<ILV, true>(red, 1, 1, diag_m, b, p, G, C);
    }
    __syncthreads();
    if (p.part_out) return;  // a frame chunk that is not the last: its sums went to part_out, nothing is finished yet
    float2* out = p.csm + (int64_t)(b - p.b0) * Ct * Ct + (int64_t)c0 * Ct + c0;
    const int col = threadIdx.x & 63;
    if (col < C)
        for (int row = threadIdx.x >> 6; row < C; row += 4) out[row * Ct + col] = G[row * (C + 1) + col];
}

// generic: grid = (bins, tile pairs I >= J of 32 x 32 channels)
__global__ __launch_bounds__(256) void k_csm_gemm(CsmArgs p) {
    __shared__ CsmRed red;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int b = blockIdx.x + p.b0;
    int I = 0;
    int tp = blockIdx.y;
    while ((I + 1) * (I + 2) / 2 <= tp) ++I;
    tp -= I * (I + 1) / 2;
    const int J = tp;
    const int C = p.n_ch, F = p.n_frames;
    const int ci = 32 * I + (l & 31), cj = 32 * J + (l & 31);
    const float2* Xb = p.X + (int64_t)b * F * C;
    f32x16 re = {0}, im = {0};
    // k-steps (2 frames each) w, w+4, w+8, ... ; U of them are loaded before their 4*U MFMAs
    // are issued, so the global-load latency hides behind the matrix pipe
    constexpr int U = 4;
    const int fo = l >> 5;
    for (int s0 = w; 2 * s0 < F; s0 += 4 * U) {
        float2 a[U], bb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = 2 * (s0 + 4 * u) + fo;
            a[u] = make_float2(0.f, 0.f);
            bb[u] = make_float2(0.f, 0.f);
            if (f < F) {
                if (ci < C) a[u] = Xb[(int64_t)f * C + ci];
                if (cj < C) bb[u] = Xb[(int64_t)f * C + cj];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            re = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, bb[u].x, re, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, bb[u].y, re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, bb[u].x, im, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x2f32(-a[u].x, bb[u].y, im, 0, 0, 0);
        }
    }
    csm_tile_epilogue(red, re, im, I, J, b, p);
}

// Up to 64 channels: ONE workgroup per frequency bin computes all three 32 x 32 tile pairs from
// the same two operand loads per k-step (12 MFMAs per KiB of operands instead of 4: the generic
// kernel is bound by the operand traffic from L2, not by the matrix pipe).  grid = bins - 1
// workgroups, two per CU on the 64-mic shape: workgroup j < bins - 2 takes bin j + 1; the last
// one takes the two purely real bins (DC and Nyquist), which need one MFMA per tile pair and
// k-step instead of four (their imaginary parts are exactly zero and the imaginary accumulator
// of the full form stays +0), so every workgroup has about the same amount of work.
template <bool REAL_BIN>
__device__ __forceinline__ void csm64_bin(int b, const CsmArgs& p, f32x16& re00, f32x16& im00, f32x16& re10,
                                          f32x16& im10, f32x16& re11, f32x16& im11) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int C = p.n_ch, F = p.n_frames;
    const int c0 = l & 31, c1 = 32 + (l & 31);
    const float2* Xb = p.X + (int64_t)b * F * C;
    constexpr int U = 4;  // 40 MFMAs per batch; with 8 the kernel needs 376 registers (one workgroup per CU,
                          // its three fp64 epilogues exposed); 4 fits 256 -> two per CU, epilogues overlap
    const int fo = l >> 5;
    // operands of U k-steps (2 frames each; this wave takes k-steps w, w+4, ...) are fetched one
    // whole iteration (12 U MFMAs) ahead of their use.  No branch per load (hipcc would drain
    // vmcnt at each): channels beyond C and frames beyond F are read from a clamped address and
    // zeroed by a select.
    const int c0c = min(c0, C - 1), c1c = min(c1, C - 1);
    const float m0 = c0 < C ? 1.f : 0.f, m1 = c1 < C ? 1.f : 0.f;
    // raw loads only (clamped addresses): the masks are applied when the batch is consumed, one
    // iteration later, so nothing here waits for the data
    auto fetch = [&](int s0, float2 (&r0)[U], float2 (&r1)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float2* row = Xb + (int64_t)min(2 * (s0 + 4 * u) + fo, F - 1) * C;
            r0[u] = row[c0c];
            r1[u] = row[c1c];
        }
    };
    auto consume = [&](int s0, const float2 (&r0)[U], const float2 (&r1)[U]) {
        float2 x0[U], x1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float v = (2 * (s0 + 4 * u) + fo) < F ? 1.f : 0.f;
            x0[u] = make_float2(r0[u].x * (m0 * v), r0[u].y * (m0 * v));
            x1[u] = make_float2(r1[u].x * (m1 * v), r1[u].y * (m1 * v));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // tile (I, J): A operand = rows of tile I, B operand = columns of tile J
            re00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[u].x, x0[u].x, re00, 0, 0, 0);
            re10 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[u].x, x0[u].x, re10, 0, 0, 0);
            re11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[u].x, x1[u].x, re11, 0, 0, 0);
            if (!REAL_BIN) {
                re00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[u].y, x0[u].y, re00, 0, 0, 0);
                im00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[u].y, x0[u].x, im00, 0, 0, 0);  // M only (DIAG_M)
                re10 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[u].y, x0[u].y, re10, 0, 0, 0);
                im10 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[u].y, x0[u].x, im10, 0, 0, 0);
                im10 = __builtin_amdgcn_mfma_f32_32x32x2f32(-x1[u].x, x0[u].y, im10, 0, 0, 0);
                re11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[u].y, x1[u].y, re11, 0, 0, 0);
                im11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[u].y, x1[u].x, im11, 0, 0, 0);  // M only (DIAG_M)
            }
        }
    };
    // two register sets used alternately (no copies of values that are still in flight)
    float2 a0[U], a1[U], b0[U], b1[U];
    fetch(w, a0, a1);
    for (int s0 = w; 2 * s0 < F; s0 += 8 * U) {
        fetch(s0 + 4 * U, b0, b1);
        __builtin_amdgcn_sched_barrier(0);  // the next batch is in flight during these MFMAs
        consume(s0, a0, a1);
        if (2 * (s0 + 4 * U) >= F) break;
        fetch(s0 + 8 * U, a0, a1);
        __builtin_amdgcn_sched_barrier(0);
        consume(s0 + 4 * U, b0, b1);
    }
}

__global__ __launch_bounds__(256, 2) void k_csm_gemm64(CsmArgs p) {
    __shared__ CsmRed red;
    __shared__ float2 G[CSM64_G];
    const int nb = p.fin.nb;
    if ((int)blockIdx.x < nb - 2) {
        const int b = (int)blockIdx.x + 1;
        f32x16 re00 = {0}, im00 = {0}, re10 = {0}, im10 = {0}, re11 = {0}, im11 = {0};
        csm64_bin<false>(b, p, re00, im00, re10, im10, re11, im11);
        csm_epilogue64<false>(red, G, re00, im00, re10, im10, re11, im11, true, b, p);
        return;
    }
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
        const int b = e ? nb - 1 : 0;
        f32x16 re00 = {0}, im00 = {0}, re10 = {0}, im10 = {0}, re11 = {0}, im11 = {0};
        csm64_bin<true>(b, p, re00, im00, re10, im10, re11, im11);
        if (e) __syncthreads();  // red and G are reused
        csm_epilogue64<false>(red, G, re00, im00, re10, im10, re11, im11, false, b, p);
    }
}

// ---------------------------------------------------------------- delay-and-sum beamformer map
// (reference: BeamformerDASFrequency.get_beamformer_map, beamforming/beamforming.py:853-858)
// map[g][f] = Re( h_f[:, g]^H  CSM_f  h_f[:, g] ) for every grid point g and frequency bin f.
// grid = (ceil(G/128), F), 4 waves, one 32-point grid tile per wave.  Per tile of 32 channels
// the product M = CSM_f[rows, :] h_f[:, tile] runs on the fp32 MFMA pipe (32x32x2, four real
// products per complex one); the epilogue contracts M with conj(h) over the channels in fp64.
struct DasArgs {
    const float2* csm;  // [F][C][C]
    const float2* h;    // [F][C][G]
    int n_bins, n_ch, n_grid;
    float* map;  // [G][F]
};

__global__ __launch_bounds__(256) void k_das_map(DasArgs p) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int f = blockIdx.y;
    const int C = p.n_ch, G = p.n_grid;
    const int g = (blockIdx.x * 4 + w) * 32 + (l & 31);
    const int kh = l >> 5;
    const float2* Cf = p.csm + (int64_t)f * C * C;
    const float2* Hf = p.h + (int64_t)f * C * G;
    const bool gv = g < G;
    double part = 0.0;
    const int gc = min(g, G - 1);
    const float mg = gv ? 1.f : 0.f;
    constexpr int U = 4;
    for (int i0 = 0; i0 < C; i0 += 32) {
        f32x16 mr = {0}, mi = {0};
        const int ci = i0 + (l & 31);
        const int cic = min(ci, C - 1);
        const float mc = ci < C ? 1.f : 0.f;
        // branch-free loads from clamped addresses, masked when consumed; two register sets used
        // alternately so the next batch is in flight during the current batch's MFMAs
        auto fetch = [&](int s0, float2 (&a)[U], float2 (&b)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = min(2 * (s0 + u) + kh, C - 1);
                a[u] = Cf[(int64_t)cic * C + k];
                b[u] = Hf[(int64_t)k * G + gc];
            }
        };
        auto consume = [&](int s0, const float2 (&ra)[U], const float2 (&rb)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float v = (2 * (s0 + u) + kh) < C ? 1.f : 0.f;
                const float2 a = make_float2(ra[u].x * (mc * v), ra[u].y * (mc * v));
                const float2 b = make_float2(rb[u].x * (mg * v), rb[u].y * (mg * v));
                mr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, mr, 0, 0, 0);
                mr = __builtin_amdgcn_mfma_f32_32x32x2f32(-a.y, b.y, mr, 0, 0, 0);
                mi = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.y, mi, 0, 0, 0);
                mi = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, mi, 0, 0, 0);
            }
        };
        float2 a0[U], b0[U], a1[U], b1[U];
        fetch(0, a0, b0);
        for (int s0 = 0; 2 * s0 < C; s0 += 2 * U) {
            fetch(s0 + U, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            consume(s0, a0, b0);
            if (2 * (s0 + U) >= C) break;
            fetch(s0 + 2 * U, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            consume(s0 + U, a1, b1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = i0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float2 hv = Hf[(int64_t)min(c, C - 1) * G + gc];
            const double t = (double)hv.x * (double)mr[r] + (double)hv.y * (double)mi[r];
            part += (gv && c < C) ? t : 0.0;
        }
    }
    part += __shfl_xor(part, 32);
    if (kh == 0 && gv) p.map[(int64_t)g * p.n_bins + f] = (float)part;
}

}  // namespace dsk
