// Dev harness: the one-launch Welch kernel (welch4096::k_h1f) against the three-launch path
// (k_x3 + k_y3 + k_welch_finish) on the headline shape: outputs, repeated launches (counter
// reset), launches beside a competing kernel (uneven load), timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/exp/exp_fused tools/exp/exp_fused.hip
//   tools/exp/exp_fused [n_samples] [n_ch] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kernels_welch4096f.hpp"

namespace w4 = welch4096;
#define CK(e)                                                                      \
    do {                                                                           \
        hipError_t e_ = (e);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <typename T>
static T* dalloc(size_t n) {
    T* p;
    CK(hipMalloc((void**)&p, n * sizeof(T)));
    CK(hipMemset(p, 0, n * sizeof(T)));
    return p;
}
__global__ void k_noise(float* buf, size_t n, int iters) {  // competing load: streams memory on some CUs
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it)
        for (size_t j = i; j < n; j += (size_t)gridDim.x * blockDim.x) acc += buf[j];
    if (acc == 123.456f) buf[0] = acc;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int n_ch = argc > 2 ? atoi(argv[2]) : 64;
    const int rounds = argc > 3 ? atoi(argv[3]) : 6;
    const int hop = 2048;
    const int n_frames = (int)((n + hop - 1) / hop);
    w4::Plan pl = w4::plan3(n_frames, n_ch);
    printf("n %lld ch %d frames %d pairs %d chunks %d grid %d\n", (long long)n, n_ch, n_frames, pl.n_pairs, pl.n_chunks, pl.n_chunks * n_ch);
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, w4::k_h1f, w4::NT, w4::LDS3_BYTES));
    printf("occupancy query: %d workgroups of k_h1f per CU\n", per_cu);
    if (pl.n_chunks * n_ch > std::min(per_cu, 3) * 256) {
        printf("grid not resident at once: refusing\n");
        return 1;
    }
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 0.3f);
    std::vector<float> hx(n), hy((size_t)n_ch * n), hw(4096);
    for (auto& v : hx) v = nd(rng);
    for (size_t i = 0; i < hy.size(); ++i) hy[i] = (0.2f + 0.01f * (i / n)) * hx[i % n] + nd(rng);
    for (int i = 0; i < 4096; ++i) hw[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / 4096.0));
    std::vector<float2> ht;
    w4::host_tables(ht);
    float *x = dalloc<float>(n), *y = dalloc<float>((size_t)n_ch * n), *win = dalloc<float>(4096);
    float2* twt = dalloc<float2>(ht.size());
    CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(win, hw.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(twt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
    const size_t nxy = (size_t)pl.n_chunks * n_ch * w4::NB, nout = (size_t)w4::NB * n_ch;
    float2* xs = dalloc<float2>((size_t)pl.n_pairs * w4::N);
    float* px = dalloc<float>((size_t)pl.n_pairs * w4::NB);
    float* psx = dalloc<float>((size_t)pl.n_chunks * w4::NB);
    float2* pxy = dalloc<float2>(nxy);
    float* pyy = dalloc<float>(nxy);
    float2 *tf0 = dalloc<float2>(nout), *tf1 = dalloc<float2>(nout);
    float *coh0 = dalloc<float>(nout), *coh1 = dalloc<float>(nout);
    unsigned* sync = dalloc<unsigned>(w4::F_SYNC_WORDS);
    w4::Args ax{x, n, n, 1, hop, n_frames, pl.n_pairs, 1, pl.n_chunks, pl.ppc, win, twt, xs, px, pxy, pyy, psx};
    ax.n_cx = 1;
    w4::Args ay = ax;
    ay.sig = y;
    ay.n_ch = n_ch;
    w4::place_remainder(ay, n_ch);
    dsk::FinishPar fin{1.0 / n_frames, 1.0, 0, 1, w4::NB};
    dsk::WelchFinArgs f0{psx, pxy, pyy, pl.n_chunks, pl.n_chunks, 1, n_ch, 0, 1, fin, tf0, coh0};
    w4::FusedArgs fa;
    fa.a = ay;
    fa.a.xsig = x;
    fa.sync = sync;
    fa.mode = 1;
    fa.fin = fin;
    fa.tf = tf1;
    fa.coh = coh1;
#if W4F_STAMPS
    fa.stamps = dalloc<unsigned long long>((size_t)pl.n_chunks * n_ch * 16);
#endif
    hipStream_t st, st2;
    CK(hipStreamCreate(&st));
    CK(hipStreamCreate(&st2));
    const int64_t total = (int64_t)w4::NB * n_ch;
    auto run3 = [&]() {
        hipLaunchKernelGGL(w4::k_x3, dim3(pl.n_pairs), dim3(256), w4::LDS3_BYTES, st, ax);
        hipLaunchKernelGGL((w4::k_y3<false>), dim3(pl.n_chunks * n_ch), dim3(256), w4::LDS3_BYTES, st, ay);
        hipLaunchKernelGGL(dsk::k_welch_finish, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, f0);
    };
    auto runf = [&]() { hipLaunchKernelGGL(w4::k_h1f, dim3(pl.n_chunks * n_ch), dim3(256), w4::LDS3_BYTES, st, fa); };
    std::vector<float2> a(nout), b(nout);
    std::vector<float> ca(nout), cb(nout);
    std::vector<unsigned> hs(w4::F_SYNC_WORDS);
    auto compare = [&](const char* what) {
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        CK(hipMemcpy(a.data(), tf0, nout * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), tf1, nout * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ca.data(), coh0, nout * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(cb.data(), coh1, nout * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), sync, hs.size() * 4, hipMemcpyDeviceToHost));
        double m = 0, d = 0, dc = 0;
        size_t bad = 0;
        for (size_t i = n_ch; i < nout; ++i) {  // (bin 0 is 0/0 with detrend)
            m = std::max(m, (double)hypotf(a[i].x, a[i].y));
            double e = hypotf(a[i].x - b[i].x, a[i].y - b[i].y);
            d = std::max(d, e);
            dc = std::max(dc, (double)fabsf(ca[i] - cb[i]));
            if (!(e <= 1e-5 * m + 1e-30) || !(fabsf(ca[i] - cb[i]) <= 1e-5f)) ++bad;
        }
        unsigned left = 0;
        for (auto v : hs) left |= v;
        printf("%-34s tf rel-max %.3e  coh abs-max %.3e  mismatching %zu  counters left %u (timeout code %u)\n", what, d / m, dc, bad, left, hs[0]);
        CK(hipMemsetAsync(tf1, 0xff, nout * 8, st));
        CK(hipMemsetAsync(coh1, 0xff, nout * 4, st));
    };
    run3();
    runf();
    compare("first launch");
    for (int i = 0; i < 5; ++i) runf();
    compare("after 5 more launches");
    // uneven load: a streaming kernel on a second stream occupies part of the chip while the fused one runs
    float* junk = dalloc<float>((size_t)64 << 20);
    hipEvent_t n0, n1, f0e, f1e;
    CK(hipEventCreate(&n0));
    CK(hipEventCreate(&n1));
    CK(hipEventCreate(&f0e));
    CK(hipEventCreate(&f1e));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(n0, st2));
        hipLaunchKernelGGL(k_noise, dim3(64 * (rep + 1)), dim3(256), 0, st2, junk, (size_t)64 << 20, 3);
        CK(hipEventRecord(n1, st2));
        CK(hipEventRecord(f0e, st));
        runf();
        CK(hipEventRecord(f1e, st));
        CK(hipStreamSynchronize(st2));
        CK(hipStreamSynchronize(st));
        float tn, tf_ms;
        CK(hipEventElapsedTime(&tn, n0, n1));
        CK(hipEventElapsedTime(&tf_ms, f0e, f1e));
        printf("  competing kernel %.2f ms, one-launch kernel beside it %.2f ms\n", tn, tf_ms);
        compare("beside a competing kernel");
    }
    // other data in between (the spectra buffers are rewritten every launch: stale copies would show)
    for (int rep = 0; rep < 2; ++rep) {
        for (auto& v : hx) v = nd(rng);
        CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
        run3();
        runf();
        compare("new input data");
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20;
    for (int r = 0; r < rounds; ++r) {
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) {
                if (v == 0)
                    run3();
                else
                    runf();
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[v], e0, e1));
        }
        printf("round %d: three launches %.1f us | one launch %.1f us\n", r, 1e3 * ms[0] / iters, 1e3 * ms[1] / iters);
    }
    compare("after timing");
#if W4F_STAMPS
    {
        const int nb = pl.n_chunks * n_ch;
        CK(hipMemset(fa.stamps, 0, (size_t)nb * 128));
        for (int i = 0; i < 3; ++i) runf();
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)nb * 16);
        CK(hipMemcpy(h.data(), fa.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb; ++b) t0 = std::min(t0, h[16 * b]);
        const char* names[16] = {"start", "produce done", "first xs poll", "poll satisfied", "loop done", "partials published", "grid wait over", "finish done",
                                 "P: samples windowed", "P: transform done", "P: stores issued", "P: drained", "", "", "", ""};
        for (int i = 0; i < 12; ++i) {
            std::vector<double> v;
            for (int b = 0; b < nb; ++b)
                if (h[16 * b + i]) v.push_back((double)(h[16 * b + i] - t0) / 100.0);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  %-20s n %4zu  min %7.1f  p10 %7.1f  p50 %7.1f  p90 %7.1f  max %7.1f us\n", names[i], v.size(), v[0], v[v.size() / 10], v[v.size() / 2], v[9 * v.size() / 10], v.back());
        }
    }
#endif
    return 0;
}
