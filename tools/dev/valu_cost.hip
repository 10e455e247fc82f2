// Dev microbenchmark (round 5): what a vector instruction costs by the number of VGPR operands it reads,
// in cycles (s_memtime) AND in steady-state wall time (the chip lowers its clock under dense vector load).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/valu_cost tools/dev/valu_cost.hip
// Every mode: 16 independent accumulators, 64 instructions per loop trip, 3 workgroups of 256 per CU by default.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float ka, float kb) {
    float s[16], m[4], c[4];
    for (int i = 0; i < 16; ++i) s[i] = (float)threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; ++i) m[i] = 1.0001f + 1e-6f * (threadIdx.x + i), c[i] = 0.5f + 1e-5f * (threadIdx.x + i);
    asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = (i + r) & 3;
                if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c[j]));
                if (MODE == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(m[j]));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m[j]), "v"(c[(j + 1) & 3]));
                if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "s"(ka), "v"(c[j]));      // SGPR multiplier
                if (MODE == 4) asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(s[i]) : "v"(c[j]));     // literal multiplier 1.0001
                if (MODE == 5) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(s[i]) : "v"(m[j]));     // literal addend
                if (MODE == 6) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(m[j]), "v"(c[j]));   // reads dst too
                if (MODE == 7) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "s"(kb), "v"(c[j]));     // SGPR x VGPR + dst
                if (MODE == 8) asm volatile("v_mul_f32 %0, 0x3f800347, %0" : "+v"(s[i]));                       // literal x VGPR: one VGPR read
                if (MODE == 9) {  // the loop mix of welch4096::k_y3 as built in round 4: 7 add, 3 mul, 6 fmac
                    if (i < 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c[j]));
                    else if (i < 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(m[j]));
                    else asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(m[j]), "v"(c[j]));
                }
                if (MODE == 10) {  // a folded-constant mix: 10 add, 4 fmamk (literal), 2 fma with three VGPRs
                    if (i < 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c[j]));
                    else if (i < 14) asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(s[i]) : "v"(c[j]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m[j]), "v"(c[(j + 1) & 3]));
                }
                if (MODE == 11) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(s[i]) : "v"(c[j]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        cyc[2048 + blockIdx.x] = r1 - r0;
    }
}
template <int MODE>
void run(const char* name, int wg_per_cu, float* out, unsigned long long* cyc, int iters, int reps) {
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.9999f);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(4096);
    hipMemcpy(h.data(), cyc, 4096 * 8, hipMemcpyDeviceToHost);
    double avg = 0, rt = 0;
    for (int i = 0; i < grid; ++i) avg += h[i], rt += h[2048 + i];
    avg /= grid;
    rt /= grid;
    const double n_inst = (double)iters * 64 * wg_per_cu;  // wave instructions per SIMD per launch
    printf("%-44s %d/SIMD: wall %8.1f us/launch  clock %.2f GHz  %5.2f cycles  %6.3f ns(wall) per wave instruction per SIMD\n", name,
           wg_per_cu, ms * 1e3 / reps, avg / rt / 10.0, avg / n_inst, ms * 1e6 / reps / n_inst);
}
int main(int argc, char** argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 3;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;   // 128 k instructions per wave: ~0.3 ms per launch
    const int reps = argc > 3 ? atoi(argv[3]) : 600;     // ~0.2 s per mode: past the clock ramp
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&cyc, 4096 * 8);
    for (int pass = 0; pass < 2; ++pass) {
        printf("pass %d\n", pass);
        run<0>("v_add_f32 v,v", w, out, cyc, iters, reps);
        run<11>("v_sub_f32 v,v", w, out, cyc, iters, reps);
        run<1>("v_mul_f32 v,v", w, out, cyc, iters, reps);
        run<8>("v_mul_f32 literal,v", w, out, cyc, iters, reps);
        run<2>("v_fma_f32 v,v,v", w, out, cyc, iters, reps);
        run<3>("v_fma_f32 v,s,v", w, out, cyc, iters, reps);
        run<4>("v_fmamk_f32 v,literal,v", w, out, cyc, iters, reps);
        run<5>("v_fmaak_f32 v,v,literal", w, out, cyc, iters, reps);
        run<6>("v_fmac_f32 v,v (+dst)", w, out, cyc, iters, reps);
        run<7>("v_fmac_f32 s,v (+dst)", w, out, cyc, iters, reps);
        run<9>("mix 7 add : 3 mul : 6 fmac", w, out, cyc, iters, reps);
        run<10>("mix 10 add : 4 fmamk : 2 fma", w, out, cyc, iters, reps);
    }
    return 0;
}
