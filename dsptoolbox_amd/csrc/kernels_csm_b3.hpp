// Cross-spectral matrix of up to 64 channels on the bf16 matrix pipe with fp32-exact operands:
// every fp32 value is split into three bf16 pieces (a = a_h + a_m + a_l exactly: 24 = 8 + 8 + 8
// significand bits) and a product a b is the six piece products of weight >= 2^-16,
//   a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_m b_m + a_l b_h),
// each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; what is dropped (a_m b_l,
// a_l b_m, a_l b_l) is below 2^-23 |a b|, the rounding of the fp32 product itself.  The bf16
// pipe runs 16 x the fp32 one (MI355X: 2.5 PFLOP/s against 157 TFLOP/s), so six bf16
// instructions per 16 frames replace eight fp32 ones at 2.7 x the rate: the Gram product stops
// being matrix-pipe-bound (0.114 ms with v_mfma_f32_32x32x2_f32, kernels_finish.hpp) and runs
// at the rate its operand stream arrives from HBM.  (reference: _csm_welch,
// standard/_spectral_methods.py:285-371 -- the per-bin  X^H X  over the frames.)  gfx950.
//
//   X[b][f][c] (the STFT layout), C <= 64.  One workgroup per bin (the two purely real edge bins
//   share the last one), four waves split the k-steps of 16 frames.  Tile T (T = 0, 1) holds
//   the channels 2 r + T, so one 16-byte load per lane and frame -- (re, im) of channels
//   2 r and 2 r + 1 -- feeds row r of both tiles; lanes 0..31 take frames 0..7 of the k-step,
//   lanes 32..63 frames 8..15 (operand map of the instruction: row = lane & 31, k = 8 (lane >> 5) + j).
//   Loads are range-checked buffer loads: frames past F and channels past C read as zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels_finish.hpp"

namespace csmb3 {

using dsk::CsmArgs;
using dsk::f32x16;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

struct Pieces {  // eight consecutive frames of one quantity: three bf16 pieces each
    u32x4 h, m, l;
};

// frames (2 jj, 2 jj + 1) -> dword jj of the three fragments
__device__ __forceinline__ void split_pair(float a, float b, int jj, Pieces& o) {
    const uint32_t h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    const uint32_t m = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    o.h[jj] = h;
    o.m[jj] = m;
    o.l[jj] = cvt_pk_bf16(sa, sb);  // exact: at most 8 significant bits are left
}

__device__ __forceinline__ f32x16 mma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
}

// acc += A B^T over the 16 frames of the k-step, smallest terms first
__device__ __forceinline__ void prod(f32x16& acc, const Pieces& A, const Pieces& B) {
    acc = mma(A.m, B.m, acc);
    acc = mma(A.h, B.l, acc);
    acc = mma(A.l, B.h, acc);
    acc = mma(A.h, B.m, acc);
    acc = mma(A.m, B.h, acc);
    acc = mma(A.h, B.h, acc);
}

__device__ __forceinline__ Pieces negated(const Pieces& a) {
    Pieces o;
    o.h = a.h ^ 0x80008000u;
    o.m = a.m ^ 0x80008000u;
    o.l = a.l ^ 0x80008000u;
    return o;
}

// ---- the k-loop of a complex bin, fed by LDS-DMA (buffer_load ... lds): the loads write a per-wave
// ring of two 8 KB buffers in LDS instead of 32 VGPRs, so TWO k-steps per wave (32 MB over the
// chip) are in flight instead of one.  A workgroup's duration is its 16 dependent load round
// trips; with one k-step in flight per wave (register loads, which is all the 256-register budget
// allows beside 96 accumulators and 60 operand registers) the same loop took 80 us instead of 70.
// The ring shares its LDS with the epilogue's buffers.
typedef int v4i __attribute__((ext_vector_type(4)));

// one wave-instruction: 64 lanes x 16 bytes from rsrc + voff (range-checked: zeros past the end)
// to LDS byte address lds .. lds + 1023, lane-linear.  M0 carries the LDS address and belongs to
// the compiler: saved and restored in the same statement.
__device__ __forceinline__ void dma16(const v4i& rsrc, uint32_t voff, uint32_t lds) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds)
        : "memory");
}

// c0, cg: the channels [c0, c0 + cg) of the rows (cg < 0: all of them)
__device__ __forceinline__ void bin_dma(char* smem, int b, const CsmArgs& p, f32x16& re00, f32x16& im00,
                                        f32x16& re10, f32x16& im10, f32x16& re11, f32x16& im11, int c0 = 0,
                                        int cg = -1) {
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int C = p.n_ch, F = p.n_frames, Cg = cg < 0 ? C : cg;
    const uint32_t row_bytes = (uint32_t)C * 8u, bin_bytes = (uint32_t)F * row_bytes;
    const uint64_t base = (uint64_t)(p.X + (int64_t)b * F * C);
    const v4i rs = {__builtin_amdgcn_readfirstlane((int)(uint32_t)base),
                    __builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) & 0xffff, (int)bin_bytes, 0x00020000};
    const int r = l & 31, h = l >> 5;
    const uint32_t lane_off = 2 * r < Cg ? (uint32_t)(8 * h) * row_bytes + 8u * (uint32_t)(c0 + 2 * r) : bin_bytes;
    const int nks = (F + 15) >> 4;
    // this wave's ring: buffers k = 0, 1 of 8 rows x 1 KB
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)w * 16384u;
    const float4* rd = reinterpret_cast<const float4*>(smem + w * 16384) + l;
    auto issue = [&](int s, int k) {
        const uint32_t off = s < nks ? lane_off + (uint32_t)(16 * s) * row_bytes : bin_bytes;
#pragma unroll
        for (int j = 0; j < 8; ++j) dma16(rs, off + j * row_bytes, ring + (uint32_t)k * 8192u + (uint32_t)j * 1024u);
    };
    issue(w, 0);
    issue(w + 4, 1);
    int k = 0;
    for (int s = w; s < nks; s += 4, k ^= 1) {
        // the eight oldest loads (buffer k) have landed; the other buffer's eight may still be in flight
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        Pieces R0, R1, I0, I1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float4 qa = rd[k * 512 + (2 * jj) * 64], qb = rd[k * 512 + (2 * jj + 1) * 64];
            split_pair(qa.x, qb.x, jj, R0);
            split_pair(qa.z, qb.z, jj, R1);
            split_pair(qa.y, qb.y, jj, I0);
            split_pair(qa.w, qb.w, jj, I1);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue(s + 8, k);  // buffer k has been read: refill it two k-steps ahead
        __builtin_amdgcn_sched_barrier(0);
        prod(re00, R0, R0);
        prod(re10, R1, R0);
        prod(re11, R1, R1);
        prod(re00, I0, I0);
        prod(im00, I0, R0);  // M only (diag_m)
        prod(re10, I1, I0);
        prod(im10, I1, R0);
        prod(re11, I1, I1);
        prod(im11, I1, R1);  // M only (diag_m)
        const Pieces N1 = negated(R1);  // last use of R1: negated in place
        prod(im10, N1, I0);
    }
    // the refills past the last k-step write zeros: all of them must have landed before the
    // epilogue reuses the ring
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The two purely real bins (DC and Nyquist: imaginary parts exactly zero, three products of
// 6 instructions per k-step) in ONE pass over the frames, both operand streams in flight together:
// a workgroup's duration is set by its 16 dependent load round trips, not by its arithmetic, so
// two bins one after the other would make this workgroup the straggler of the launch (twice the
// round trips of every other one).
__device__ __forceinline__ void bins_real2(const CsmArgs& p, f32x16& a00, f32x16& a10, f32x16& a11, f32x16& b00,
                                           f32x16& b10, f32x16& b11) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int C = p.n_ch, F = p.n_frames, nb = p.fin.nb;
    const uint32_t row_bytes = (uint32_t)C * 8u, bin_bytes = (uint32_t)F * row_bytes;
    const __amdgpu_buffer_rsrc_t rsa =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.X), 0, (int)bin_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float2*>(p.X + (int64_t)(nb - 1) * F * C), 0, (int)bin_bytes, 0x00020000);
    const int r = l & 31, h = l >> 5;
    const uint32_t lane_off = 2 * r < C ? (uint32_t)(8 * h) * row_bytes + 16u * r : bin_bytes;
    const int nks = (F + 15) >> 4;
    auto step_off = [&](int s) { return s < nks ? lane_off + (uint32_t)(16 * s) * row_bytes : bin_bytes; };
    auto fetch2 = [&](const __amdgpu_buffer_rsrc_t& rs, uint32_t off, int jj, float4 (&q)[8]) {
        q[2 * jj] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (2 * jj) * row_bytes, 0, 0));
        q[2 * jj + 1] =
            __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (2 * jj + 1) * row_bytes, 0, 0));
    };
    float4 qa[8], qb[8];
    {
        const uint32_t off = step_off(w);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            fetch2(rsa, off, jj, qa);
            fetch2(rsb, off, jj, qb);
        }
    }
    for (int s = w; s < nks; s += 4) {
        Pieces R0a, R1a, R0b, R1b;
        const uint32_t off_next = step_off(s + 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            split_pair(qa[2 * jj].x, qa[2 * jj + 1].x, jj, R0a);
            split_pair(qa[2 * jj].z, qa[2 * jj + 1].z, jj, R1a);
            split_pair(qb[2 * jj].x, qb[2 * jj + 1].x, jj, R0b);
            split_pair(qb[2 * jj].z, qb[2 * jj + 1].z, jj, R1b);
            __builtin_amdgcn_sched_barrier(0);
            fetch2(rsa, off_next, jj, qa);
            fetch2(rsb, off_next, jj, qb);
            __builtin_amdgcn_sched_barrier(0);
        }
        prod(a00, R0a, R0a);
        prod(b00, R0b, R0b);
        prod(a10, R1a, R0a);
        prod(b10, R1b, R0b);
        prod(a11, R1a, R1a);
        prod(b11, R1b, R1b);
    }
}

// grid = bins - 1: workgroup j < bins - 2 takes bin j + 1, the last one both real bins
__global__ __launch_bounds__(256, 2) void k_csm_gemm64_b3(CsmArgs p) {
    // epilogue buffers (partial tiles, staged matrix); the DMA ring of the k-loop lies over them
    __shared__ __attribute__((aligned(16))) char smem[sizeof(dsk::CsmRed) + sizeof(float2) * dsk::CSM64_G];
    static_assert(sizeof(smem) >= 4 * 16384, "ring of four waves");
    dsk::CsmRed& red = *reinterpret_cast<dsk::CsmRed*>(smem);
    float2* G = reinterpret_cast<float2*>(smem + sizeof(dsk::CsmRed));
    const int nb = p.fin.nb;
    if ((int)blockIdx.x >= nb - 2) {
        f32x16 a00 = {0}, a10 = {0}, a11 = {0}, b00 = {0}, b10 = {0}, b11 = {0};
        const f32x16 zero = {0};
        bins_real2(p, a00, a10, a11, b00, b10, b11);
        dsk::csm_epilogue64<true>(red, G, a00, zero, a10, zero, a11, zero, false, 0, p);
        __syncthreads();  // red and G are reused
        dsk::csm_epilogue64<true>(red, G, b00, zero, b10, zero, b11, zero, false, nb - 1, p);
        return;
    }
    const int b = (int)blockIdx.x + 1;
    f32x16 re00 = {0}, im00 = {0}, re10 = {0}, im10 = {0}, re11 = {0}, im11 = {0};
    bin_dma(smem, b, p, re00, im00, re10, im10, re11, im11);
    __syncthreads();  // every wave's ring is idle
    dsk::csm_epilogue64<true>(red, G, re00, im00, re10, im10, re11, im11, true, b, p);
}

// a range of bins (the multi-GPU split by frequency): workgroup j takes bin b0 + j; the purely real
// edge bins go through the complex path (their imaginary pieces are zeros)
__global__ __launch_bounds__(256, 2) void k_csm_gemm64_b3_range(CsmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[sizeof(dsk::CsmRed) + sizeof(float2) * dsk::CSM64_G];
    dsk::CsmRed& red = *reinterpret_cast<dsk::CsmRed*>(smem);
    float2* G = reinterpret_cast<float2*>(smem + sizeof(dsk::CsmRed));
    const int b = p.b0 + (int)blockIdx.x;
    f32x16 re00 = {0}, im00 = {0}, re10 = {0}, im10 = {0}, re11 = {0}, im11 = {0};
    bin_dma(smem, b, p, re00, im00, re10, im10, re11, im11);
    __syncthreads();  // every wave's ring is idle
    dsk::csm_epilogue64<true>(red, G, re00, im00, re10, im10, re11, im11, true, b, p);
}

// ---- more than 64 channels: groups of 64 -----------------------------------------------------
// The matrix is cut into 64 x 64 blocks.  A diagonal block is the Gram matrix of its group: the
// kernel above with a channel offset (grid = (bins, groups); every bin through the complex path).
__global__ __launch_bounds__(256, 2) void k_csm_group_b3(CsmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[sizeof(dsk::CsmRed) + sizeof(float2) * dsk::CSM64_G];
    dsk::CsmRed& red = *reinterpret_cast<dsk::CsmRed*>(smem);
    float2* G = reinterpret_cast<float2*>(smem + sizeof(dsk::CsmRed));
    const int b = p.b0 + (int)blockIdx.x, c0 = 64 * (int)blockIdx.y, cg = min(64, p.n_ch - c0);
    f32x16 re00 = {0}, im00 = {0}, re10 = {0}, im10 = {0}, re11 = {0}, im11 = {0};
    bin_dma(smem, b, p, re00, im00, re10, im10, re11, im11, c0, cg);
    __syncthreads();  // every wave's ring is idle
    dsk::csm_epilogue64<true>(red, G, re00, im00, re10, im10, re11, im11, true, b, p, c0, cg);
}

// An off-diagonal block (group A below group B) has no symmetry: all 64 x 64 elements
//   G[a][b] = sum_f X_a conj(X_b):  Re = Ra Rb^T + Ia Ib^T,  Im = Ia Rb^T - Ra Ib^T.
// One workgroup takes the 32 rows of parity T of group A (channels a0 + 2 r + T, 8-byte loads) against
// both column tiles of group B (16-byte loads): 8 products of 6 instructions per k-step, 64
// accumulator registers (the whole block in one workgroup would need 128).  Register loads, one
// k-step ahead.  The finished elements go straight to global memory: G[a][b] and its mirror
// conj at [b][a].  grid = (16 ceil(bins / 8), pairs A > B).
__global__ __launch_bounds__(256, 2) void k_csm_offdiag_b3(CsmArgs p) {
    __shared__ dsk::CsmRed red;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int C = p.n_ch, F = p.n_frames;
    // workgroups x and x + 8 land on the same XCD one after the other: they take the two row parities of
    // one bin, so the second finds group B's rows in that XCD's L2 (the kernel is bound by its reads)
    const int T = ((int)blockIdx.x >> 3) & 1, bl = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7);
    if (bl >= p.n_groups_bins) return;
    const int b = p.b0 + bl;
    int ga = 1, pair = blockIdx.y;  // pair index -> (ga, gb), ga > gb
    while (pair >= ga) {
        pair -= ga;
        ++ga;
    }
    const int gb = pair, a0 = 64 * ga, b0c = 64 * gb, ca = min(64, C - a0);
    const uint32_t row_bytes = (uint32_t)C * 8u, bin_bytes = (uint32_t)F * row_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float2*>(p.X + (int64_t)b * F * C), 0, (int)bin_bytes, 0x00020000);
    const int r = l & 31, h = l >> 5;
    const uint32_t offa = 2 * r + T < ca ? (uint32_t)(8 * h) * row_bytes + 8u * (uint32_t)(a0 + 2 * r + T) : bin_bytes;
    const uint32_t offb = (uint32_t)(8 * h) * row_bytes + 8u * (uint32_t)(b0c + 2 * r);  // group B is a full one
    const int nks = (F + 15) >> 4;
    f32x16 re0 = {0}, im0 = {0}, re1 = {0}, im1 = {0};
    float2 qa[8];
    float4 qb[8];
    auto fetch = [&](int s) {
        const uint32_t oa = s < nks ? offa + (uint32_t)(16 * s) * row_bytes : bin_bytes;
        const uint32_t ob = s < nks ? offb + (uint32_t)(16 * s) * row_bytes : bin_bytes;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qa[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, oa + j * row_bytes, 0, 0));
            qb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ob + j * row_bytes, 0, 0));
        }
    };
    fetch(w);
    for (int s = w; s < nks; s += 4) {
        Pieces Ra, Ia, R0, I0, R1, I1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            split_pair(qa[2 * jj].x, qa[2 * jj + 1].x, jj, Ra);
            split_pair(qa[2 * jj].y, qa[2 * jj + 1].y, jj, Ia);
            split_pair(qb[2 * jj].x, qb[2 * jj + 1].x, jj, R0);
            split_pair(qb[2 * jj].y, qb[2 * jj + 1].y, jj, I0);
            split_pair(qb[2 * jj].z, qb[2 * jj + 1].z, jj, R1);
            split_pair(qb[2 * jj].w, qb[2 * jj + 1].w, jj, I1);
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(s + 4);
        __builtin_amdgcn_sched_barrier(0);
        prod(re0, Ra, R0);
        prod(re1, Ra, R1);
        prod(re0, Ia, I0);
        prod(re1, Ia, I1);
        prod(im0, Ia, R0);
        prod(im1, Ia, R1);
        const Pieces Na = negated(Ra);  // last use of Ra: negated in place
        prod(im0, Na, I0);
        prod(im1, Na, I1);
    }
    // ---- epilogue: combine the four waves' partial tiles, finish, store the element and its mirror
    const double e = p.fin.halve_edges ? ((b == 0 || b == p.fin.nb - 1) ? 0.5 * p.fin.factor : p.fin.factor) : 1.0;
    float2* out = p.csm + (int64_t)(b - p.b0) * C * C;
#pragma unroll  // (rolled, hipcc selects the accumulators through scratch memory)
    for (int U = 0; U < 2; ++U) {
        if (U) __syncthreads();  // red is reused
        dsk::csm_tile_put(red, U ? re1 : re0, U ? im1 : im0);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int rg = w * 4 + rr;
            const float gx = (red[0][0][rg][l] + red[1][0][rg][l]) + (red[2][0][rg][l] + red[3][0][rg][l]);
            const float gy = (red[0][1][rg][l] + red[1][1][rg][l]) + (red[2][1][rg][l] + red[3][1][rg][l]);
            const int i = (rg & 3) + 8 * (rg >> 2) + 4 * (l >> 5), j = l & 31;
            const int gi = a0 + 2 * i + T, gj = b0c + 2 * j + U;
            // (the dispatch requires >= 8 frames: the one-frame sign rule of csm_tile_reduce_store never applies)
            double vx = (double)gx * p.fin.inv * e, vy = (double)gy * p.fin.inv * e;
            if (p.fin.amp_sqrt) {
                const double x = vx, y = vy;
                const double rad = sqrt(x * x + y * y);
                const double t = sqrt(0.5 * (rad + fabs(x)));
                const double q = fabs(y) / (2.0 * t);
                const bool pos = x >= 0.0, zero = rad == 0.0;
                vx = zero ? 0.0 : (pos ? t : q);
                vy = zero ? y : copysign(pos ? q : t, y);
            }
            if (2 * i + T < ca) {
                const float fx = (float)vx, fy = (float)vy;
                out[(int64_t)gi * C + gj] = make_float2(fx, fy);
                out[(int64_t)gj * C + gi] = make_float2(fx, -fy);
            }
        }
    }
}

// X of one bin must stay below 2^31 bytes for the 32-bit buffer offsets
__host__ inline bool fits_groups(int n_ch, int n_frames) {  // more than 64 channels: groups of 64
    return n_ch > 64 && n_ch <= 1024 && (int64_t)n_frames * n_ch * 8 < (int64_t)1 << 31;
}
__host__ inline bool fits(int n_ch, int n_frames) {
    // odd counts too: the last lane pair's second channel is then the first value of the next row (or
    // zero past the end of the bin) and lands in tile rows / columns the epilogue does not store; the
    // 16-byte loads are 8-byte aligned in that case, which buffer loads allow
    return n_ch >= 2 && n_ch <= 64 && (int64_t)n_frames * n_ch * 8 < (int64_t)1 << 31;
}

}  // namespace csmb3
