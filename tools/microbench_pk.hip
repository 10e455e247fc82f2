// Dev microbenchmark: radix-16 butterfly + 15 twiddle multiplies per iteration, scalar fp32 VALU vs
// packed v_pk_{add,mul,fma}_f32 (ext_vector_type(2), op_sel / neg modifiers, no glue moves).
// MI355X: 214 scalar instructions at 2.2 cycles each vs 127 packed at 4.2 cycles: packed fp32 has
// no throughput advantage on gfx950.  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk tools/microbench_pk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

// ---------- scalar reference (as in kernels_welch4096.hpp)
__device__ __forceinline__ void r4s(float2& a, float2& b, float2& c, float2& d) {
    float2 s0 = make_float2(a.x + c.x, a.y + c.y), d0 = make_float2(a.x - c.x, a.y - c.y);
    float2 s1 = make_float2(b.x + d.x, b.y + d.y), d1 = make_float2(b.x - d.x, b.y - d.y);
    a = make_float2(s0.x + s1.x, s0.y + s1.y);
    c = make_float2(s0.x - s1.x, s0.y - s1.y);
    b = make_float2(d0.x + d1.y, d0.y - d1.x);
    d = make_float2(d0.x - d1.y, d0.y + d1.x);
}
__device__ __forceinline__ void dft16s(float2 (&v)[16]) {
    constexpr float C8 = 0.92387953251128673848f, S8 = 0.38268343236508978178f;
    constexpr float R2 = 0.70710678118654752440f;
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) r4s(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    auto mulw = [](float2 z, float c, float s) { return make_float2(fmaf(z.x, c, z.y * s), fmaf(z.y, c, -z.x * s)); };
    v[5] = mulw(v[5], C8, S8);
    v[9] = make_float2((v[9].x + v[9].y) * R2, (v[9].y - v[9].x) * R2);
    v[13] = mulw(v[13], S8, C8);
    v[6] = make_float2((v[6].x + v[6].y) * R2, (v[6].y - v[6].x) * R2);
    v[10] = make_float2(v[10].y, -v[10].x);
    v[14] = make_float2((v[14].y - v[14].x) * R2, -(v[14].x + v[14].y) * R2);
    v[7] = mulw(v[7], S8, C8);
    v[11] = make_float2((v[11].y - v[11].x) * R2, -(v[11].x + v[11].y) * R2);
    v[15] = mulw(v[15], -C8, -S8);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) r4s(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
// ---------- packed
__device__ __forceinline__ v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
// a - i b = (a.x + b.y, a.y - b.x) ; a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f sub_i(v2f a, v2f b) { const v2f m = {1.f, -1.f}; return swp(b) * m + a; }
__device__ __forceinline__ v2f add_i(v2f a, v2f b) { const v2f m = {-1.f, 1.f}; return swp(b) * m + a; }
__device__ __forceinline__ void r4p(v2f& a, v2f& b, v2f& c, v2f& d) {
    v2f s0 = a + c, d0 = a - c, s1 = b + d, d1 = b - d;
    a = s0 + s1;
    c = s0 - s1;
    b = sub_i(d0, d1);
    d = add_i(d0, d1);
}
// z * (c - i s) = (x c + y s, y c - x s)
__device__ __forceinline__ v2f mulw_p(v2f z, float c, float s) {
    v2f cc = {c, c}, ss = {s, -s};
    return z * cc + swp(z) * ss;
}
__device__ __forceinline__ void dft16p(v2f (&v)[16]) {
    constexpr float C8 = 0.92387953251128673848f, S8 = 0.38268343236508978178f;
    constexpr float R2 = 0.70710678118654752440f;
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) r4p(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    v[5] = mulw_p(v[5], C8, S8);
    v[9] = mulw_p(v[9], R2, R2);
    v[13] = mulw_p(v[13], S8, C8);
    v[6] = mulw_p(v[6], R2, R2);
    { const v2f m = {1.f, -1.f}; v[10] = swp(v[10]) * m; }
    v[14] = mulw_p(v[14], -R2, R2);
    v[7] = mulw_p(v[7], S8, C8);
    v[11] = mulw_p(v[11], -R2, R2);
    v[15] = mulw_p(v[15], -C8, -S8);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) r4p(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}
// complex multiply z * w
__device__ __forceinline__ v2f cmulp(v2f z, v2f w) {
    v2f wr = {w.x, w.x}, wi = {-w.y, w.y};
    return z * wr + swp(z) * wi;
}

__global__ void k_scalar(float2* io, const float2* tw, int iters) {
    float2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = io[threadIdx.x + 256 * i + blockIdx.x * 4096];
    float2 w[15];
    for (int i = 0; i < 15; ++i) w[i] = tw[threadIdx.x + 256 * i];
    for (int it = 0; it < iters; ++it) {
        dft16s(v);
#pragma unroll
        for (int i = 1; i < 16; ++i) {
            float2 a = v[i], b = w[i - 1];
            v[i] = make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
        }
    }
    for (int i = 0; i < 16; ++i) io[threadIdx.x + 256 * i + blockIdx.x * 4096] = v[i];
}
__global__ void k_packed(v2f* io, const v2f* tw, int iters) {
    v2f v[16];
    for (int i = 0; i < 16; ++i) v[i] = io[threadIdx.x + 256 * i + blockIdx.x * 4096];
    v2f w[15];
    for (int i = 0; i < 15; ++i) w[i] = tw[threadIdx.x + 256 * i];
    for (int it = 0; it < iters; ++it) {
        dft16p(v);
#pragma unroll
        for (int i = 1; i < 16; ++i) v[i] = cmulp(v[i], w[i - 1]);
    }
    for (int i = 0; i < 16; ++i) io[threadIdx.x + 256 * i + blockIdx.x * 4096] = v[i];
}

int main() {
    const int nb = 2048, n = nb * 4096;
    float2 *d, *tw;
    hipMalloc(&d, n * sizeof(float2));
    hipMalloc(&tw, 4096 * sizeof(float2));
    float2* h = (float2*)malloc(n * sizeof(float2));
    float2* ht = (float2*)malloc(4096 * sizeof(float2));
    for (int i = 0; i < 4096; ++i) { float a = 0.001f * i; ht[i] = make_float2(cosf(a), sinf(a)); }
    hipMemcpy(tw, ht, 4096 * sizeof(float2), hipMemcpyHostToDevice);
    float2* out[2];
    for (int var = 0; var < 2; ++var) {
        for (int i = 0; i < n; ++i) h[i] = make_float2(0.001f * (i % 977), -0.002f * (i % 311));
        hipMemcpy(d, h, n * sizeof(float2), hipMemcpyHostToDevice);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        const int iters = 200;
        // warm
        if (var == 0) k_scalar<<<nb, 256>>>(d, tw, 1); else k_packed<<<nb, 256>>>((v2f*)d, (const v2f*)tw, 1);
        hipMemcpy(d, h, n * sizeof(float2), hipMemcpyHostToDevice);
        hipEventRecord(a);
        if (var == 0) k_scalar<<<nb, 256>>>(d, tw, iters); else k_packed<<<nb, 256>>>((v2f*)d, (const v2f*)tw, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        out[var] = (float2*)malloc(n * sizeof(float2));
        hipMemcpy(out[var], d, n * sizeof(float2), hipMemcpyDeviceToHost);
        printf("%s: %.3f ms for %d iters (%.1f ns per dft16+tw per wave-slot)\n", var ? "packed" : "scalar", ms, iters, ms * 1e6 / iters);
    }
    double md = 0, mx = 0;
    for (int i = 0; i < n; ++i) { md = fmax(md, fabs(out[0][i].x - out[1][i].x)); mx = fmax(mx, fabs(out[0][i].x)); }
    printf("max diff %g of %g\n", md, mx);
    return 0;
}
