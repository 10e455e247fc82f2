// Pure-VALU rate of the radix-16 butterfly code of the Welch kernel (no LDS, no HBM in the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../dsptoolbox_amd/csrc/kernels_welch4096.hpp"
using namespace welch4096;

template <int ITERS>
__global__ __launch_bounds__(256) void k_valu(float2* out, const float2* in, const float2* twt) {
    extern __shared__ float2 dummy[];
    float2 v[16];
    Tw tw;
    for (int k = 0; k < 15; ++k) tw.w[k] = twt[k * 256 + threadIdx.x];
    for (int j = 0; j < 16; ++j) v[j] = in[threadIdx.x + 256 * j];
    for (int i = 0; i < ITERS; ++i) {
        dft16(v);
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) v[pos16(k1)] = cmul(v[pos16(k1)], tw.w[k1 - 1]);
    }
    for (int j = 0; j < 16; ++j) out[(blockIdx.x * 256 + threadIdx.x) + j] = v[j];
    if (threadIdx.x == 1000) dummy[0] = v[0];
}

int main() {
    float2 *out, *in, *twt;
    hipMalloc(&out, 2048 * 256 * 17 * 8);
    hipMalloc(&in, 4096 * 8);
    hipMalloc(&twt, 4096 * 8);
    hipMemset(in, 0, 4096 * 8);
    hipMemset(twt, 0, 4096 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    constexpr int IT = 400;
    auto k = k_valu<IT>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    struct Cfg { int blocks; int lds; const char* name; };
    Cfg cfgs[] = {{256, 150 * 1024, "1 WG/CU (1 wave/SIMD)"}, {512, 76 * 1024, "2 WG/CU"}, {768, 50 * 1024, "3 WG/CU"},
                  {1024, 38 * 1024, "4 WG/CU"}, {2048, 16 * 1024, "8 WG/CU"}};
    for (auto& c : cfgs) {
        hipLaunchKernelGGL(k, dim3(c.blocks), dim3(256), c.lds, 0, out, in, twt);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(c.blocks), dim3(256), c.lds, 0, out, in, twt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double inst = 218.0;  // ~158 (dft16) + 60 (15 cmul) VALU instructions per iteration
        double per_simd = (double)c.blocks * 4 / 1024 * IT * inst;
        printf("%-24s %.3f ms  -> %.2f ns per VALU instr per SIMD\n", c.name, ms, ms * 1e6 / per_simd);
    }
    return 0;
}
