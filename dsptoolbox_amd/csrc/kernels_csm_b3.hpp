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
//   X[b][f][c] (the STFT layout), C even.  One workgroup per bin (the two purely real edge bins
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

__device__ __forceinline__ void bin(int b, const CsmArgs& p, f32x16& re00, f32x16& im00, f32x16& re10, f32x16& im10,
                                    f32x16& re11, f32x16& im11) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int C = p.n_ch, F = p.n_frames;
    const uint32_t row_bytes = (uint32_t)C * 8u, bin_bytes = (uint32_t)F * row_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float2*>(p.X + (int64_t)b * F * C), 0, (int)bin_bytes, 0x00020000);
    const int r = l & 31, h = l >> 5;
    // channels 2 r, 2 r + 1 of frame 16 s + 8 h + j; a lane without channels reads past the end
    const uint32_t lane_off = 2 * r < C ? (uint32_t)(8 * h) * row_bytes + 16u * r : bin_bytes;
    const int nks = (F + 15) >> 4;
    // past the last k-step every lane reads past the end (zeros, no memory traffic)
    auto step_off = [&](int s) { return s < nks ? lane_off + (uint32_t)(16 * s) * row_bytes : bin_bytes; };
    auto fetch2 = [&](uint32_t off, int jj, float4 (&q)[8]) {
        q[2 * jj] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (2 * jj) * row_bytes, 0, 0));
        q[2 * jj + 1] =
            __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (2 * jj + 1) * row_bytes, 0, 0));
    };
    // one k-step: split the loaded values into bf16 pieces -- every pair of frames that has been
    // split makes room for the same pair of the next k-step, whose loads go out at once -- then the
    // 60 matrix instructions of this one, under which those loads land
    float4 q[8];
    {
        const uint32_t off = step_off(w);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) fetch2(off, jj, q);
    }
    for (int s = w; s < nks; s += 4) {
        Pieces R0, R1, I0, I1;
        const uint32_t off_next = step_off(s + 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            split_pair(q[2 * jj].x, q[2 * jj + 1].x, jj, R0);
            split_pair(q[2 * jj].z, q[2 * jj + 1].z, jj, R1);
            split_pair(q[2 * jj].y, q[2 * jj + 1].y, jj, I0);
            split_pair(q[2 * jj].w, q[2 * jj + 1].w, jj, I1);
            __builtin_amdgcn_sched_barrier(0);
            fetch2(off_next, jj, q);
            __builtin_amdgcn_sched_barrier(0);
        }
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
    __shared__ dsk::CsmRed red;
    __shared__ float2 G[dsk::CSM64_G];
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
    bin(b, p, re00, im00, re10, im10, re11, im11);
    dsk::csm_epilogue64<true>(red, G, re00, im00, re10, im10, re11, im11, true, b, p);
}

// X of one bin must stay below 2^31 bytes for the 32-bit buffer offsets
__host__ inline bool fits(int n_ch, int n_frames) {
    return n_ch >= 2 && n_ch <= 64 && (n_ch & 1) == 0 && (int64_t)n_frames * n_ch * 8 < (int64_t)1 << 31;
}

}  // namespace csmb3
