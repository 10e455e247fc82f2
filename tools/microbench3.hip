// VMEM issue-cost micro-benchmark for gfx950: cycles a wave spends ISSUING 24 strided dword loads
// (no wait for the data) with (a) global_load + 64-bit VGPR address, (b) global_load saddr + 32-bit
// VGPR offset, (c) buffer_load offen; then the same for 8 dwordx4 loads.  512 workgroups of 256.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* base, int64_t stride_blk, int iters, float* out,
                                            unsigned long long* cyc) {
    const int tid = threadIdx.x;
    const float* p = base + (int64_t)blockIdx.x * stride_blk;
    float acc = 0.f;
    unsigned long long issue = 0, total = 0;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 1 << 30, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const float* q = p + (int64_t)it * 2048;
        float v[24];
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {  // 64-bit per-lane address
            const float* ql = q + tid;
            asm volatile("" : "+v"(ql));
#pragma unroll
            for (int m = 0; m < 24; ++m) v[m] = ql[256 * m];
        } else if (MODE == 1) {  // uniform base + 32-bit lane offset
#pragma unroll
            for (int m = 0; m < 24; ++m) v[m] = q[tid + 256 * m];
        } else if (MODE == 2) {  // buffer load, offen
#pragma unroll
            for (int m = 0; m < 24; ++m)
                v[m] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (tid + 256 * (m & 3)) * 4,
                                                                           (it * 2048 + 1024 * (m >> 2)) * 4, 0));
        } else {  // 6 x dwordx4, perfectly coalesced
            const float4* q4 = reinterpret_cast<const float4*>(q) + tid;
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                float4 t = q4[256 * m];
                v[4 * m] = t.x; v[4 * m + 1] = t.y; v[4 * m + 2] = t.z; v[4 * m + 3] = t.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 24; ++m) acc += v[m];
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        issue += t1 - t0;
        total += t2 - t0;
    }
    out[blockIdx.x * 256 + tid] = acc;
    if (tid == 0 && blockIdx.x == 0) { cyc[0] = issue; cyc[1] = total; }
}

int main() {
    const int blocks = 512, iters = 64;
    const int64_t stride = 2048 * (iters + 4);
    float *buf, *out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&buf, sizeof(float) * stride * blocks + (1 << 20)));
    CHECK(hipMemset(buf, 0, sizeof(float) * stride * blocks + (1 << 20)));
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    CHECK(hipMalloc(&cyc, 16));
    const char* names[4] = {"global 64-bit vaddr x24", "global saddr+voff  x24", "buffer offen        x24", "global dwordx4      x6"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, stride, iters, out, cyc);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, stride, iters, out, cyc);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, stride, iters, out, cyc);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, buf, stride, iters, out, cyc);
            CHECK(hipDeviceSynchronize());
            unsigned long long h[2];
            CHECK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
            if (rep) printf("%s: issue %.0f cyc/iter (%.1f per instr), until data %.0f cyc\n", names[mode],
                            (double)h[0] / iters, (double)h[0] / iters / (mode == 3 ? 6 : 24), (double)h[1] / iters);
        }
    return 0;
}
