// Does the register alignment of the A / B operands of v_mfma_f32_32x32x16_bf16 matter?  (gfx950)
// A VGPR tuple starts at a register number n; n mod 4 is its bank offset.  hipcc allocates operand
// fragments 4-aligned, so A and B of every instruction start in the same bank.
//   mode 0: A = v[8:11],  B = v[12:15]   (offsets 0 / 0)
//   mode 1: A = v[8:11],  B = v[14:17]   (offsets 0 / 2)
//   mode 2: A = B = v[8:11]
//   mode 3: A = v[8:11],  B = v[13:16]   (offsets 0 / 1, if the assembler takes an odd tuple)
// With PARTNER the odd waves of a workgroup run an independent VALU loop instead (the conversion
// work of the real kernel): does the operand fetch of the matrix instructions take issue cycles
// from them?
// build: hipcc -O3 --offload-arch=gfx950 -o tools/exp/mfma_bank tools/exp/mfma_bank.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define MFMA4(A, B)                                                   \
    "v_mfma_f32_32x32x16_bf16 a[0:15], " A ", " B ", a[0:15]\n\t"     \
    "v_mfma_f32_32x32x16_bf16 a[16:31], " A ", " B ", a[16:31]\n\t"   \
    "v_mfma_f32_32x32x16_bf16 a[32:47], " A ", " B ", a[32:47]\n\t"   \
    "v_mfma_f32_32x32x16_bf16 a[48:63], " A ", " B ", a[48:63]\n\t"

#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "a0", "a1", "a2", "a3", "a4", \
    "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21",       \
    "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38",  \
    "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55",  \
    "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"

template <int MODE>
__device__ __forceinline__ void mfma_block() {
    if (MODE == 0) asm volatile(MFMA4("v[8:11]", "v[12:15]") MFMA4("v[8:11]", "v[12:15]")::: CLOB);
    if (MODE == 1) asm volatile(MFMA4("v[8:11]", "v[14:17]") MFMA4("v[8:11]", "v[14:17]")::: CLOB);
    if (MODE == 2) asm volatile(MFMA4("v[8:11]", "v[8:11]") MFMA4("v[8:11]", "v[8:11]")::: CLOB);
#ifdef ODD_TUPLE
    if (MODE == 3) asm volatile(MFMA4("v[8:11]", "v[13:16]") MFMA4("v[8:11]", "v[13:16]")::: CLOB);
#endif
}

template <int MODE, bool PARTNER>
__global__ __launch_bounds__(512) void k(long long* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    long long t0, t1;
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = seed + i;
    if (PARTNER && wave >= 4) {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], 1.0001f, seed);  // 64 independent-ish VALU
        }
        t1 = __builtin_readcyclecounter();
    } else {
        asm volatile(
            "v_mov_b32 v8, 0x3f803f80\n\tv_mov_b32 v9, 0x3f803f80\n\tv_mov_b32 v10, 0x3f803f80\n\tv_mov_b32 v11, 0x3f803f80\n\t"
            "v_mov_b32 v12, 0x3f803f80\n\tv_mov_b32 v13, 0x3f803f80\n\tv_mov_b32 v14, 0x3f803f80\n\tv_mov_b32 v15, 0x3f803f80\n\t"
            "v_mov_b32 v16, 0x3f803f80\n\tv_mov_b32 v17, 0x3f803f80\n\t" ::: CLOB);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) mfma_block<MODE>();
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        t1 = __builtin_readcyclecounter();
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i];
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2] = t1 - t0;
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = (long long)s;
    }
}

template <int MODE, bool PARTNER>
static void run(const char* name, int threads, int iters) {
    const int blocks = 256, waves = blocks * threads / 64;
    long long* d;
    hipMalloc(&d, sizeof(long long) * 2 * waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, PARTNER><<<blocks, threads>>>(d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, PARTNER><<<blocks, threads>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * waves);
    hipMemcpy(h.data(), d, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    int nm = 0, nv = 0;
    for (int w = 0; w < waves; ++w) {
        const bool partner = PARTNER && (w % (threads / 64)) >= 4;
        (partner ? v : m) += (double)h[2 * w];
        (partner ? nv : nm) += 1;
    }
    printf("%-28s threads %4d  %.3f ms  cycles/MFMA %.1f", name, threads, ms, m / nm / (8.0 * iters));
    if (nv) printf("   partner cycles/VALU %.2f", v / nv / (64.0 * iters));
    printf("\n");
    hipFree(d);
}

int main() {
    const int iters = 2000;
    for (int threads : {256, 512}) {
        run<0, false>("A v[8:11]  B v[12:15]", threads, iters);
        run<1, false>("A v[8:11]  B v[14:17]", threads, iters);
        run<2, false>("A = B = v[8:11]", threads, iters);
#ifdef ODD_TUPLE
        run<3, false>("A v[8:11]  B v[13:16]", threads, iters);
#endif
    }
    for (int threads : {512}) {
        run<0, true>("partner, B v[12:15]", threads, iters);
        run<1, true>("partner, B v[14:17]", threads, iters);
        run<2, true>("partner, A = B", threads, iters);
    }
    return 0;
}
