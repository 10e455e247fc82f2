// Dev tool: what does a pure store stream reach on this GPU?  (The FIR bank writes 32 x its input.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/dev/fill_bw tools/dev/fill_bw.hip && tools/dev/fill_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)
template <int W>  // W floats per lane and store
__global__ void k_fill(float* p, size_t n, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x * W;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W; i + W <= n; i += stride) {
        if (W == 4) *reinterpret_cast<float4*>(p + i) = make_float4(v, v, v, v);
        else p[i] = v;
    }
}
// the FIR kernel's store shape: 16 dword stores per thread, 1 KB apart, 256 B per wave instruction
__global__ void k_fill_fir(float* p, size_t n, float v) {
    const size_t blocks = n / 4096;
    for (size_t b = blockIdx.x; b < blocks; b += gridDim.x)
#pragma unroll
        for (int m = 0; m < 16; ++m) p[b * 4096 + threadIdx.x + 256 * m] = v;
}
__global__ void k_copy(const float4* a, float4* b, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
__global__ void k_read(const float4* a, float* out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { float4 q = a[i]; s += q.x + q.y + q.z + q.w; }
    if (s == 1.2345f) out[0] = s;
}
int main() {
    const size_t n = (size_t)1 << 30;  // 4 GiB of floats
    float *a, *b;
    CK(hipMalloc((void**)&a, n * 4));
    CK(hipMalloc((void**)&b, n * 4));
    CK(hipMemset(a, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    };
    for (int grid : {2048, 8192}) {
        printf("grid %d x 256\n", grid);
        time("fill, 16 B per lane", [&] { hipLaunchKernelGGL(k_fill<4>, dim3(grid), dim3(256), 0, 0, b, n, 1.f); }, n * 4.0);
        time("fill, 4 B per lane", [&] { hipLaunchKernelGGL(k_fill<1>, dim3(grid), dim3(256), 0, 0, b, n, 1.f); }, n * 4.0);
        time("fill, FIR store shape (16 x 4 B, 1 KB apart)", [&] { hipLaunchKernelGGL(k_fill_fir, dim3(grid), dim3(256), 0, 0, b, n, 1.f); }, n * 4.0);
        time("read, 16 B per lane", [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, (const float4*)a, b, n / 4); }, n * 4.0);
        time("copy, 16 B per lane (read + written bytes)", [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); }, n * 8.0);
    }
    return 0;
}
