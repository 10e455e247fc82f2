// Dev harness (round 4): the Welch H1 step as three launches (k_x3 -> k_y3 -> k_welch_finish, what ships)
// against two launches with last-arriver finishes (kernels_welch4096la.hpp), same box, alternating rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/exp/exp_la tools/exp/exp_la.hip
//   (-DW4LA_STAMPS=1: per-workgroup s_memrealtime stamps of the two-launch form)
//   tools/exp/exp_la [n_samples] [n_ch] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kernels_welch4096la.hpp"

namespace w4 = welch4096;
#define CK(e)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (e);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

template <typename T>
static T* dalloc(size_t n) {
    T* p;
    CK(hipMalloc((void**)&p, n * sizeof(T)));
    CK(hipMemset(p, 0, n * sizeof(T)));
    return p;
}
static double relmax(const std::vector<float>& a, const std::vector<float>& b) {
    double m = 0, d = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        m = std::max(m, (double)fabsf(b[i]));
        d = std::max(d, (double)fabsf(a[i] - b[i]));
    }
    return d / (m > 0 ? m : 1);
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int n_ch = argc > 2 ? atoi(argv[2]) : 64;
    const int rounds = argc > 3 ? atoi(argv[3]) : 8;
    const int hop = 2048;
    const int n_frames = (int)((n + hop - 1) / hop);
    w4::Plan pl = w4::plan3(n_frames, n_ch);
    printf("n %lld ch %d frames %d pairs %d chunks %d (grid %d)\n", (long long)n, n_ch, n_frames, pl.n_pairs, pl.n_chunks,
           pl.n_chunks * n_ch);
    if (pl.n_chunks > w4::LA_MAX || n_ch > w4::LA_MAX) return 1;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 0.3f);
    std::vector<float> hx(n), hy((size_t)n_ch * n), hw(4096);
    for (auto& v : hx) v = nd(rng);
    for (size_t i = 0; i < hy.size(); ++i) hy[i] = 0.5f * hx[i % n] + nd(rng);
    for (int i = 0; i < 4096; ++i) hw[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / 4096.0));
    std::vector<float2> ht;
    w4::host_tables(ht);
    float *x = dalloc<float>(n), *y = dalloc<float>((size_t)n_ch * n), *win = dalloc<float>(4096);
    float2* twt = dalloc<float2>(ht.size());
    CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(win, hw.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(twt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
    const size_t units = (size_t)pl.n_chunks * n_ch;
    float2* xs = dalloc<float2>((size_t)pl.n_pairs * w4::N);
    // three launches: rows of NB
    float* px = dalloc<float>((size_t)pl.n_pairs * w4::NB);
    float* psx = dalloc<float>((size_t)pl.n_chunks * w4::NB);
    float2* pxy = dalloc<float2>(units * w4::NB);
    float* pyy = dalloc<float>(units * w4::NB);
    // two launches: rows of NBP
    float* lpx = dalloc<float>((size_t)pl.n_pairs * w4::NBP);
    float* lpsx = dalloc<float>((size_t)pl.n_chunks * w4::NBP);
    float2* lpxy = dalloc<float2>(units * w4::NBP);
    float* lpyy = dalloc<float>(units * w4::NBP);
    unsigned* cnt = dalloc<unsigned>(2 * w4::LA_MAX);
    const size_t nout = (size_t)w4::NB * n_ch;
    float2 *tfA = dalloc<float2>(nout), *tfB = dalloc<float2>(nout);
    float *cohA = dalloc<float>(nout), *cohB = dalloc<float>(nout);

    w4::Args ax{x, n, n, 1, hop, n_frames, pl.n_pairs, 1, pl.n_chunks, pl.ppc, win, twt, xs, px, pxy, pyy, psx};
    ax.n_cx = 1;
    w4::Args ay = ax;
    ay.sig = y;
    ay.n_ch = n_ch;
    w4::place_remainder(ay, n_ch);
    w4::Args axc = ax;  // k_x3la needs the chunk split too
    axc.n_chunks = ay.n_chunks;
    axc.use_plus = ay.use_plus;
    for (int i = 0; i < 24; ++i) axc.plus[i] = ay.plus[i];
    axc.n_ch = n_ch;
    const dsk::FinishPar fin{1.0 / (double)n_frames, 1.0, 0, 1, w4::NB};
    dsk::WelchFinArgs f{psx, pxy, pyy, pl.n_chunks, pl.n_chunks, 1, n_ch, 0, 1, fin, tfA, cohA};
    w4::LaArgs lx{};
    lx.a = axc;
    lx.a.px = lpx;
    lx.a.psx = lpsx;
    lx.a.pxy = lpxy;
    lx.a.pyy = lpyy;
    lx.cnt = cnt;
    lx.mode = 1;
    lx.fin = fin;
    lx.tf = tfB;
    lx.coh = cohB;
    w4::LaArgs ly = lx;
    ly.a.sig = y;
    ly.a.n_ch = n_ch;
#if W4LA_STAMPS
    unsigned long long* stamps = dalloc<unsigned long long>(units * 8);
    lx.stamps = stamps;
    ly.stamps = stamps;
#endif
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const unsigned fin_grid = (unsigned)((nout + 63) / 64);
    auto run_a = [&]() {
        hipLaunchKernelGGL(w4::k_x3, dim3(pl.n_pairs), dim3(256), w4::LDS3_BYTES, st, ax);
        hipLaunchKernelGGL((w4::k_y3<false>), dim3(units), dim3(256), w4::LDS3_BYTES, st, ay);
        hipLaunchKernelGGL(dsk::k_welch_finish, dim3(fin_grid), dim3(256), 0, st, f);
    };
    auto run_b = [&]() {
        hipLaunchKernelGGL(w4::k_x3la, dim3(pl.n_pairs), dim3(256), w4::LDS3_BYTES, st, lx);
        hipLaunchKernelGGL(w4::k_y3la, dim3(units), dim3(256), w4::LDS3_BYTES, st, ly);
    };
    std::vector<float> a_tf(2 * nout), b_tf(2 * nout), a_coh(nout), b_coh(nout);
    run_a();
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    for (int rep = 0; rep < 3; ++rep) {  // repeated launches: the counters must come back to zero
        CK(hipMemsetAsync(tfB, 0xff, nout * 8, st));
        CK(hipMemsetAsync(cohB, 0xff, nout * 4, st));
        run_b();
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        CK(hipMemcpy(a_tf.data(), tfA, nout * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(a_coh.data(), cohA, nout * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b_tf.data(), tfB, nout * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b_coh.data(), cohB, nout * 4, hipMemcpyDeviceToHost));
        size_t differ = 0;
        for (size_t i = 0; i < a_tf.size(); ++i) differ += a_tf[i] != b_tf[i];
        for (size_t i = 0; i < a_coh.size(); ++i) differ += a_coh[i] != b_coh[i];
        std::vector<unsigned> hc(2 * w4::LA_MAX);
        CK(hipMemcpy(hc.data(), cnt, hc.size() * 4, hipMemcpyDeviceToHost));
        unsigned left = 0;
        for (unsigned v : hc) left += v;
        printf("two launches vs three, run %d: tf %.3e  coh %.3e  (values that differ: %zu of %zu; counters left: %u)\n", rep,
               relmax(b_tf, a_tf), relmax(b_coh, a_coh), differ, a_tf.size() + a_coh.size(), left);
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
                    run_a();
                else
                    run_b();
            }
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[v], e0, e1));
        }
        printf("round %d: three launches %.1f us per step | two launches, last-arriver finishes %.1f us per step\n", r,
               1e3 * ms[0] / iters, 1e3 * ms[1] / iters);
    }
    // and the parts of each form
    {
        float t[5];
        auto timeit = [&](auto&& fn) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) fn();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            return 1e3f * ms / iters;
        };
        t[0] = timeit([&]() { hipLaunchKernelGGL(w4::k_x3, dim3(pl.n_pairs), dim3(256), w4::LDS3_BYTES, st, ax); });
        t[1] = timeit([&]() { hipLaunchKernelGGL((w4::k_y3<false>), dim3(units), dim3(256), w4::LDS3_BYTES, st, ay); });
        t[2] = timeit([&]() { hipLaunchKernelGGL(dsk::k_welch_finish, dim3(fin_grid), dim3(256), 0, st, f); });
        t[3] = timeit([&]() { hipLaunchKernelGGL(w4::k_x3la, dim3(pl.n_pairs), dim3(256), w4::LDS3_BYTES, st, lx); });
        t[4] = timeit([&]() { hipLaunchKernelGGL(w4::k_y3la, dim3(units), dim3(256), w4::LDS3_BYTES, st, ly); });
        printf("alone, back to back: k_x3 %.1f  k_y3 %.1f  k_welch_finish %.1f | k_x3la %.1f  k_y3la %.1f us\n", t[0], t[1], t[2],
               t[3], t[4]);
    }
#if W4LA_STAMPS
    {
        CK(hipMemset(stamps, 0, units * 8 * 8));
        run_b();
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> s(units * 8);
        CK(hipMemcpy(s.data(), stamps, s.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (size_t b = 0; b < units; ++b) t0 = std::min(t0, s[8 * b]);
        const char* names[] = {"start", "pair loop done", "partial stores issued", "stores drained (every wave, barrier)",
                               "ticket returned", "finish done (last arrivers only)"};
        for (int i = 0; i < 6; ++i) {
            std::vector<double> v;
            for (size_t b = 0; b < units; ++b)
                if (s[8 * b + i]) v.push_back((double)(s[8 * b + i] - t0) / 100.0);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            const size_t m = v.size();
            printf("  %-40s n %4zu  min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us\n", names[i], m, v[0], v[m / 10], v[m / 2],
                   v[9 * m / 10], v[m - 1]);
        }
        std::vector<double> fin_len;
        for (size_t b = 0; b < units; ++b)
            if (s[8 * b + 5]) fin_len.push_back((double)(s[8 * b + 5] - s[8 * b + 4]) / 100.0);
        std::sort(fin_len.begin(), fin_len.end());
        if (!fin_len.empty())
            printf("  a last arriver's finish takes min %.1f  p50 %.1f  max %.1f us (n %zu)\n", fin_len[0], fin_len[fin_len.size() / 2],
                   fin_len.back(), fin_len.size());
    }
#endif
    return 0;
}
