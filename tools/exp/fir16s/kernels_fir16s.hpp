// FIR filter bank with 16384-point blocks on INDEPENDENT 256-thread workgroups (round 4).  gfx950.
// (Filter.filter_signal / FilterBank.filter_signal: _lfilter_fir, classes/filter_helpers.py:454-503,
// y = oaconvolve(x, b)[:N]; _filterbank_on_signal :385-451.)
//
// Why.  kernels_fir4k.hpp (hop 2048 on 4096-point transforms) pays one inverse transform per 2048 output
// samples of a channel pair; kernels_fir16k.hpp (16384-point blocks) pays four per 16384 - (T - 1) -- 12288 for
// 4097 taps, one per 3072 -- but runs them as four lock-step teams of ONE 1024-thread workgroup per CU at 128
// registers (94 transforms per microsecond against 174 for independent 256-thread workgroups).  Here the
// 16384-point block is kept and the lock-step is dropped:
//
//   16384 = 4 x 4096.  Sub-spectrum q holds the bins 4 k' + q:
//     forward, once per block and channel pair (k_fwd, decimation in frequency):
//       b_q[n'] = ( sum_j z[n' + 4096 j] W4^(jq) ) W16384^(n' q) ,   X_q[k'] = FFT4096(b_q)[k']
//     stored to memory in the transform's own register layout (16-byte loads), 128 KB per block and pair;
//     per filter (k_fir), ONE workgroup runs the four sub-problems one after the other:
//       g_q = IFFT4096( X_q H_q ) ,   y[n' + 4096 j] = sum_q W4^(-jq) W16384^(-n' q) g_q[n']
//     g_0 .. g_2 wait (the output twiddle already applied) until g_3 is done -- g_1 and g_2 in registers, g_0 in a
//     thread-private 32 KB strip of LDS (three parked arrays beside the transform's working set, its table values
//     and the sixteen 16-byte loads in flight spilled 61 registers) --: all four end in the same thread, so the
//     radix-4 recombination needs no exchange and no barrier.
//
//   The block's spectrum cannot stay in registers across the filters (4 x 32 registers); it is re-read per filter
//   from L2 / the Infinity Cache, like the tap spectra (both 16-byte loads in register layout, requested one
//   sub-problem ahead through the call-outs of the running transform).  Workgroups are ordered filter-slice-major:
//   everybody resident works on the same few filters, whose tap spectra (128 KB each) stay in the L2s.
//
//   Every output sample is produced and stored once (overlap-save); two channels ride one complex transform;
//   signal edges are the buffer range check (loads return 0 outside the signal, stores outside are dropped).
#pragma once
#include "../../../dsptoolbox_amd/csrc/kernels_fir4k.hpp"

namespace fir16s {

namespace w4 = welch4096;
using fir16k::cmulc;
using fir16k::r4i;
using w4::cmul;
constexpr int NBIG = 16384, M = 4096, NT = 256;
constexpr int LDS_BYTES = fir4k::LDS_BYTES;            // exchange image + W256 table + its padded copy
constexpr int LDS_BYTES_FIR = LDS_BYTES + M * 8;       // k_fir: + the parking place of g_0 (32 KB): 70.6 KB, two per CU
constexpr int SUB_BYTES = M * 8;                       // one sub-spectrum in register layout: 32 KB
constexpr int UNIT_BYTES = 4 * SUB_BYTES;              // a block's (or a filter's) four sub-spectra: 128 KB

// ---- forward: a block of 16384 samples of a channel pair (or of ONE real sequence: tap spectra) -----------------
struct FwdArgs {
    const float* x;       // planar rows
    int64_t n_samples, ldx;
    int n_rows;           // channels (pairs mode) or filters (single mode)
    int single;           // 1: one real row per unit (taps), imaginary part 0
    int n_blocks;         // blocks per unit
    int hop, lead;        // block b starts at sample b * hop - lead
    float scale;          // applied to the spectrum (1 / 16384 for tap spectra)
    const float2* twt;    // welch4096::host_tables()
    const float2* twn;    // fir16k::host_tables(): [4][256] W16384^(t q), then [4][16] W64^(n1 q)
    float4* xs;           // [unit][block][4][8][256]
};

// grid = n_units * n_blocks
__global__ __launch_bounds__(NT, 2) void k_fwd(FwdArgs p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    const int tid = threadIdx.x;
    const int u = (int)blockIdx.x / p.n_blocks, b = (int)blockIdx.x - u * p.n_blocks;
    const int ra_row = p.single ? u : 2 * u, rb_row = ra_row + 1;
    const bool vb = !p.single && rb_row < p.n_rows;
    const uint32_t sig_bytes = (uint32_t)(p.n_samples * 4);
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (int64_t)ra_row * p.ldx), 0, (int)sig_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (int64_t)(vb ? rb_row : ra_row) * p.ldx), 0, vb ? (int)sig_bytes : 0, 0x00020000);
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    // samples in front of the signal (the first block of a unit) are zeros.  They are selected in software: a
    // wrapped negative offset whose immediate part carries it back over zero inside a wave was seen to return
    // samples from in front of the row (block lengths that are not multiples of 8 samples), so no load here
    // is ever issued with a negative total offset; behind the last sample the range check returns 0 as everywhere.
    const int first = b * p.hop - p.lead + tid;  // index of this thread's first sample
    float2 z[4][16];
    if (first - tid >= 0) {  // (workgroup-uniform)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1)
                z[j][n1] = make_float2(w4::ld_sample(ra, 4 * first + 4 * (256 * n1 + M * j)), w4::ld_sample(rb, 4 * first + 4 * (256 * n1 + M * j)));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int idx = first + 256 * n1 + M * j;
                int off = 4 * max(idx, 0);
                asm volatile("" : "+v"(off));  // the whole offset in the register: nothing of it in the immediate field
                const float a = w4::ld_sample(ra, off), bb = w4::ld_sample(rb, off);
                z[j][n1] = idx >= 0 ? make_float2(a, bb) : make_float2(0.f, 0.f);
            }
    }
    float4* out = p.xs + ((int64_t)blockIdx.x * 4) * (8 * 256) + tid;
    for (int q = 0; q < 4; ++q) {  // (a run-time loop: one copy of the transform in the code)
        const float2 wt = p.twn[q * 256 + tid];          // W16384^(tid q)
        const float2* c64 = p.twn + 4 * 256 + q * 16;    // W64^(n1 q), wave-uniform
        float2 v[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const float2 z0 = z[0][n1], z1 = z[1][n1], z2 = z[2][n1], z3 = z[3][n1];
            float2 s;
            if (q == 0)
                s = make_float2(z0.x + z1.x + z2.x + z3.x, z0.y + z1.y + z2.y + z3.y);
            else if (q == 1)  // z0 - i z1 - z2 + i z3
                s = make_float2(z0.x + z1.y - z2.x - z3.y, z0.y - z1.x - z2.y + z3.x);
            else if (q == 2)
                s = make_float2(z0.x - z1.x + z2.x - z3.x, z0.y - z1.y + z2.y - z3.y);
            else  // z0 + i z1 - z2 - i z3
                s = make_float2(z0.x - z1.y - z2.x + z3.y, z0.y + z1.x - z2.y - z3.x);
            v[n1] = cmul(s, cmul(wt, c64[n1]));
        }
        w4::fft4096_w(v, tw, buf, tw2, tid);
#pragma unroll
        for (int g = 0; g < 8; ++g)
            out[(q * 8 + g) * 256] = make_float4(v[2 * g].x * p.scale, v[2 * g].y * p.scale, v[2 * g + 1].x * p.scale, v[2 * g + 1].y * p.scale);
    }
}

// ---- per filter: product, four inverse sub-transforms, recombination, store ---------------------------------------
struct Args {
    const float4* xs;   // [pair][block][4][8][256]: k_fwd of the signal
    const float4* hp;   // [filter][4][8][256]: k_fwd of the taps, 1 / 16384 folded in
    int64_t n_samples, ld_y;
    int n_ch, n_filt, n_taps;
    int n_blocks;       // per channel pair
    int n_units;        // pairs * n_blocks
    int fslice;         // filters per workgroup
    const float2* twt;  // welch4096::host_tables()
    const float2* twn;  // fir16k::host_tables()
    float* y;           // [(f n_ch + c) ld_y + n]
};

// grid = n_units * ceil(n_filt / fslice), slice-major: workgroup w -> slice w / n_units, unit w % n_units
__global__ __launch_bounds__(NT, 2) void k_fir(Args p) {
    extern __shared__ __align__(16) float2 lds[];
    float2* buf = lds;
    float2* tw2 = lds + 16 * w4::L1S;
    float2* tw2p = lds + 16 * w4::L1S + 256;
    float2* park = lds + LDS_BYTES / 8 + threadIdx.x;  // g_0[n1] at park[256 n1]: private to the thread
    const int tid = threadIdx.x;
    const int slice = (int)blockIdx.x / p.n_units, unit = (int)blockIdx.x - slice * p.n_units;
    const int pair = unit / p.n_blocks, blk = unit - pair * p.n_blocks;
    const int f0 = slice * p.fslice, f1 = min(f0 + p.fslice, p.n_filt);
    const int ca = 2 * pair, cb = ca + 1;
    const bool vb = cb < p.n_ch;
    const int T1 = p.n_taps - 1, L = NBIG - T1;
    w4::Tw6 tw;
    w4::load_tw6(tw, p.twt, tid);
    tw2[tid] = p.twt[15 * 256 + tid];
    fir4k::fill_tw2p(tw2p, p.twt, tid);
    const float2 wt1 = p.twn[256 + tid], wt2 = p.twn[512 + tid], wt3 = p.twn[768 + tid];  // W16384^(tid q)
    // the block's spectrum and the tap spectra through raw-buffer descriptors with 32-bit byte offsets
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4*>(p.xs + (int64_t)unit * (UNIT_BYTES / 16)), 0, UNIT_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(p.hp), 0, (int)((uint32_t)p.n_filt * (uint32_t)UNIT_BYTES), 0x00020000);
    auto ld16 = [](__amdgpu_buffer_rsrc_t r, int byte_off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
    };
    const int lane_off = 16 * tid;
    float4 xq[8], hq[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        xq[g] = ld16(xrs, lane_off + 4096 * g);
        hq[g] = ld16(hrs, f0 * UNIT_BYTES + lane_off + 4096 * g);
    }
    const uint32_t sig_bytes = (uint32_t)(p.n_samples * 4);
    // output sample nn of the block (nn = tid + 256 m + 4096 j, kept from T1 on) lands at out0 + nn - T1
    const int64_t out_first = (int64_t)blk * L - T1;  // (may be negative: those samples are discarded anyway)
    float2 g1[16], g2[16];
    __syncthreads();  // the tables
    const int n_it = 4 * (f1 - f0);
    for (int it = 0; it < n_it; ++it) {
        const int q = it & 3, f = f0 + (it >> 2);
        float2 v[16];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            v[2 * g] = cmul(make_float2(xq[g].x, xq[g].y), make_float2(hq[g].x, hq[g].y));
            v[2 * g + 1] = cmul(make_float2(xq[g].z, xq[g].w), make_float2(hq[g].z, hq[g].w));
        }
        // the next sub-problem's operands ride in the transform's call-outs (behind the last one the same are fetched
        // again: a branch around loads in the middle of the transform costs more than sixteen cache hits)
        const int itn = min(it + 1, n_it - 1);
        const int xn = (itn & 3) * SUB_BYTES + lane_off;
        const int hn = (f0 + (itn >> 2)) * UNIT_BYTES + (itn & 3) * SUB_BYTES + lane_off;
        fir4k::ifft4096_wi(
            v, tw, buf, tw2p, tid,
            [&](int g) {
                xq[2 * g] = ld16(xrs, xn + 4096 * (2 * g));
                xq[2 * g + 1] = ld16(xrs, xn + 4096 * (2 * g + 1));
            },
            [&](int g) {
                hq[2 * g] = ld16(hrs, hn + 4096 * (2 * g));
                hq[2 * g + 1] = ld16(hrs, hn + 4096 * (2 * g + 1));
            });
        // v[n1] = g_q[tid + 256 n1].  q < 3: park it with its output twiddle conj(W16384^((tid + 256 n1) q)) applied
        if (q == 0) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) park[256 * n1] = v[n1];
        } else if (q == 1) {
            const float2* c64 = p.twn + 4 * 256 + 16;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) g1[n1] = cmulc(v[n1], cmul(wt1, c64[n1]));
        } else if (q == 2) {
            const float2* c64 = p.twn + 4 * 256 + 32;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) g2[n1] = cmulc(v[n1], cmul(wt2, c64[n1]));
        } else {
            const float2* c64 = p.twn + 4 * 256 + 48;
            float* __restrict__ ya = p.y + ((int64_t)f * p.n_ch + ca) * p.ld_y;
            const __amdgpu_buffer_rsrc_t oa = __builtin_amdgcn_make_buffer_rsrc(ya, 0, (int)sig_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t ob = __builtin_amdgcn_make_buffer_rsrc(ya + p.ld_y, 0, vb ? (int)sig_bytes : 0, 0x00020000);
            // byte offset of output sample nn = tid + 256 n1 + 4096 j: 4 (blk L - T1 + nn), formed modulo 2^32 -- a
            // position in front of the signal (first block) wraps past the end of the buffer and the store is
            // dropped, like every store behind the last sample (n_samples < 2^30 - 8192: fits())
            uint32_t base = (uint32_t)(out_first * 4) + 4u * (uint32_t)tid;
            asm volatile("" : "+v"(base));  // not loop invariant for the compiler: no 64 store offsets hoisted above the filter loop
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                float2 u0 = park[256 * n1], u1 = g1[n1], u2 = g2[n1], u3 = cmulc(v[n1], cmul(wt3, c64[n1]));
                r4i(u0, u1, u2, u3);  // u_j = y[tid + 256 n1 + 4096 j]
                const float2 yj[4] = {u0, u1, u2, u3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // T1 <= 8192 (the host's precondition): quarters 2 and 3 are kept whole; 0 and 1 sample by sample
                    if (j < 2 && T1 >= M * (j + 1)) continue;  // (wave-uniform) the whole quarter is discarded
                    uint32_t off = base + 4u * (uint32_t)(256 * n1 + M * j);
                    if (j < 2) off = (tid + 256 * n1 + M * j >= T1) ? off : 0xFFFFFFFCu;
                    asm volatile("" : "+v"(off));  // the whole offset in the register (the first block's base is negative)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yj[j].x), oa, (int)off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yj[j].y), ob, (int)off, 0, 0);
                }
            }
        }
    }
}

// 32-bit byte offsets into the signal rows, the block spectra of one launch and the tap spectra
inline bool fits(int64_t n_samples, int n_taps, int n_filt) {
    return n_samples > 0 && n_samples < ((int64_t)1 << 30) - 8192 && n_taps >= 2 && n_taps - 1 <= 8192 && n_filt >= 1 &&
           n_filt < 16384;
}

}  // namespace fir16s
