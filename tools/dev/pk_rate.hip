// Dev microbenchmark: issue cost of v_pk_fma_f32 / v_pk_add_f32 against v_fma_f32 / v_add_f32 at 1, 2, 3 waves
// per SIMD (the register-FFT kernels run 3).  Independent accumulators, s_memtime around an unrolled stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/pk_rate tools/dev/pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f2 a[8];
    float s[16];
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, 1.f};
    for (int i = 0; i < 16; ++i) s[i] = (float)threadIdx.x + i;
    f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    float ms = 1.0001f, cs = 0.5f;
    asm volatile("" : "+v"(m), "+v"(c), "+v"(ms), "+v"(cs));
    __syncthreads();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {  // 16 scalar fma = 8 complex
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(ms), "v"(cs));
            } else if (MODE == 1) {  // 8 packed fma
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            } else if (MODE == 2) {  // 16 scalar add
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(cs));
            } else if (MODE == 3) {  // 8 packed add
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            } else if (MODE == 4) {  // 8 packed add with swizzle + negate (multiply by -i folded in)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            } else if (MODE >= 6 && MODE <= 9) {  // packed add, dependency distance 1, 2, 4 (8 = the independent case); 9: fma distance 1
                constexpr int D = MODE == 6 ? 1 : MODE == 7 ? 2 : MODE == 8 ? 4 : 1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == 9)
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
                    else
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i % D]) : "v"(c));
                }
            } else if (MODE >= 10 && MODE <= 12) {  // scalar add, dependency distance 1, 2, 4 (two per complex op)
                constexpr int D = MODE == 10 ? 1 : MODE == 11 ? 2 : 4;
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i % D]) : "v"(cs));
            } else if (MODE == 14) {  // the complex multiplication as used: mul + fma with half selects, 4 independent values
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f2 t;
                    asm volatile("v_pk_mul_f32 %1, %0, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
                                 "v_pk_fma_f32 %0, %0, %2, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
                                 : "+v"(a[i]), "=&v"(t) : "v"(m));
                }
            } else if (MODE == 15) {  // the same complex multiplication in scalar code: 4 values x 4 instructions
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t0, t1;
                    asm volatile("v_mul_f32 %2, %1, %5\n\tv_mul_f32 %3, %0, %5\n\t"
                                 "v_fma_f32 %0, %0, %4, -%2\n\tv_fma_f32 %1, %1, %4, %3"
                                 : "+v"(s[2 * i]), "+v"(s[2 * i + 1]), "=&v"(t0), "=&v"(t1) : "v"(ms), "v"(cs));
                }
            } else if (MODE == 16) {  // packed fma with half selects only
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(a[i]) : "v"(m), "v"(c));
            } else if (MODE == 17) {  // packed mul with a broadcast low half
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(m));
            } else if (MODE == 18) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(ms));
            } else if (MODE == 19) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(ms), "v"(cs));
            } else if (MODE == 20) {  // the main loop's mix: 7 add/sub, 3 mul, 6 fma of 16
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i % 16 < 7)
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(cs));
                    else if (i % 16 < 10)
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(ms));
                    else
                        asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(ms), "v"(cs));
                }
            } else if (MODE == 13) {  // packed add, independent, s_nop 0 between
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(a[i]) : "v"(c));
            } else {  // 8 packed mul
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        cyc[1024 + blockIdx.x] = r1 - r0;
    }
}
template <int MODE>
void run(const char* name, int wg_per_cu, float* out, unsigned long long* cyc, int per_complex = 1) {
    const int iters = 200, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2048);
    hipMemcpy(h.data(), cyc, 2048 * 8, hipMemcpyDeviceToHost);
    double avg = 0, rt = 0;
    for (int i = 0; i < grid; ++i) avg += h[i], rt += h[1024 + i];
    avg /= grid;
    rt /= grid;
    const double complex_ops = (double)iters * REP;  // complex (2-float) operations per lane
    // s_memtime counts at 100 MHz on gfx950
    // s_memtime counts shader cycles, s_memrealtime 100 MHz
    const double n_inst = complex_ops * per_complex * wg_per_cu;  // wave instructions per SIMD
    printf("%-40s %d waves/SIMD: %7.1f us  in-kernel %6.1f us at %.2f GHz: %5.2f cycles = %5.2f ns per wave instruction per SIMD\n",
           name, wg_per_cu, ms * 1e3, rt / 100.0, avg / rt / 10.0, avg / n_inst, rt * 10.0 / n_inst);
}
int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 256 * 4);
    hipMalloc(&cyc, 2048 * 8);
    for (int w = 1; w <= 3; ++w) {
        run<0>("v_fma_f32 x2 (scalar)", w, out, cyc, 2);
        run<1>("v_pk_fma_f32", w, out, cyc);
        run<2>("v_add_f32 x2 (scalar)", w, out, cyc, 2);
        run<3>("v_pk_add_f32", w, out, cyc);
        run<4>("v_pk_add_f32 op_sel+neg", w, out, cyc);
        run<5>("v_pk_mul_f32", w, out, cyc);
        run<6>("v_pk_add_f32 dep distance 1", w, out, cyc);
        run<7>("v_pk_add_f32 dep distance 2", w, out, cyc);
        run<8>("v_pk_add_f32 dep distance 4", w, out, cyc);
        run<9>("v_pk_fma_f32 dep distance 1", w, out, cyc);
        run<10>("v_add_f32 x2 dep distance 1", w, out, cyc, 2);
        run<11>("v_add_f32 x2 dep distance 2", w, out, cyc, 2);
        run<12>("v_add_f32 x2 dep distance 4", w, out, cyc, 2);
        run<13>("v_pk_add_f32 + s_nop 0", w, out, cyc);
        run<18>("v_mul_f32 x2", w, out, cyc, 2);
        run<19>("v_fmac_f32_e32 x2", w, out, cyc, 2);
        run<20>("mix 7 add : 3 mul : 6 fmac", w, out, cyc, 2);
        run<14>("cmul packed x4 [half the complex ops]", w, out, cyc);
        run<15>("cmul scalar x4 [half the complex ops]", w, out, cyc);
        run<16>("v_pk_fma_f32 half selects", w, out, cyc);
        run<17>("v_pk_mul_f32 broadcast", w, out, cyc);
    }
    return 0;
}
