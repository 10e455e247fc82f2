// VALU / LDS micro-benchmarks for gfx950: v_fma_f32 vs v_pk_fma_f32 issue rate,
// ds_write_b64 / ds_read_b64 / ds_read_b128 rates.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ITERS>
__global__ void k_fma(float* out, float a, float b) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

typedef float f2 __attribute__((ext_vector_type(2)));
template <int ITERS>
__global__ void k_pkfma(float* out, float a, float b) {
    f2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
    f2 aa = {a, a}, bb = {b, b};
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(aa), "v"(bb));
        }
    }
    f2 s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// LDS: each thread writes/reads 16 x 8 B per iteration, conflict-free (consecutive lanes)
template <int MODE, int ITERS>
__global__ void k_lds(float* out) {
    __shared__ __align__(16) float2 buf[4608];
    float2 v[16];
    for (int j = 0; j < 16; ++j) v[j] = make_float2(threadIdx.x + j, j);
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // ds_write_b64
#pragma unroll
            for (int j = 0; j < 16; ++j) buf[j * 272 + threadIdx.x] = v[j];
        } else if (MODE == 1) {  // ds_read_b64
#pragma unroll
            for (int j = 0; j < 16; ++j) { float2 t = buf[j * 272 + threadIdx.x]; acc += t.x + t.y; }
        } else {  // ds_read_b128
            const float4* row = reinterpret_cast<const float4*>(buf + threadIdx.x * 18);
#pragma unroll
            for (int j = 0; j < 8; ++j) { float4 t = row[j]; acc += t.x + t.y + t.z + t.w; }
        }
        __syncthreads();
        v[i & 15].x += acc;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + v[3].x + buf[threadIdx.x].x;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 4096 * 1024 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    constexpr int IT = 2000;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads, 256 CUs
        int blocks = 256 * wps;
        float ms = time([&] { hipLaunchKernelGGL(k_fma<IT>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
        double fl = (double)blocks * 256 * IT * 64 * 2;
        printf("v_fma_f32     %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", wps, ms, fl / ms / 1e9);
        ms = time([&] { hipLaunchKernelGGL(k_pkfma<IT>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
        printf("v_pk_fma_f32  %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", wps, ms, 2 * fl / ms / 1e9);
    }
    constexpr int LI = 2000;
    for (int wps : {1, 2, 4}) {
        int blocks = 256 * wps;
        float ms = time([&] { hipLaunchKernelGGL((k_lds<0, LI>), dim3(blocks), dim3(256), 0, 0, out); });
        double by = (double)blocks * 256 * LI * 16 * 8;
        printf("ds_write_b64  %d WG/CU: %.3f ms  %.1f TB/s  (%.1f B/clk/CU @2.4GHz)\n", wps, ms, by / ms / 1e9, by / ms / 1e-3 / 256 / 2.4e9);
        ms = time([&] { hipLaunchKernelGGL((k_lds<1, LI>), dim3(blocks), dim3(256), 0, 0, out); });
        printf("ds_read_b64   %d WG/CU: %.3f ms  %.1f TB/s  (%.1f B/clk/CU)\n", wps, ms, by / ms / 1e9, by / ms / 1e-3 / 256 / 2.4e9);
        ms = time([&] { hipLaunchKernelGGL((k_lds<2, LI>), dim3(blocks), dim3(256), 0, 0, out); });
        printf("ds_read_b128  %d WG/CU: %.3f ms  %.1f TB/s  (%.1f B/clk/CU)\n", wps, ms, by / ms / 1e9, by / ms / 1e-3 / 256 / 2.4e9);
    }
    return 0;
}
