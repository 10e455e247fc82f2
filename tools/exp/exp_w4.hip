// Dev harness: welch4096::k_y (2 workgroups / CU) vs k_y3 (3 / CU) on the headline shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/exp/exp_w4 tools/exp/exp_w4.hip
//   tools/exp/exp_w4 [n_samples] [n_ch] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <map>
#include <vector>

#include "../../dsptoolbox_amd/csrc/kernels_welch4096w.hpp"

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
    w4::Plan pl = w4::plan(n_frames, n_ch);
    const int new_chunks = getenv("NEWCHUNKS") ? atoi(getenv("NEWCHUNKS")) : std::max(1, std::min(pl.n_pairs, (W4_OCC * 256) / n_ch));
    const int max_chunks = std::max(pl.n_chunks, new_chunks);
    printf("n %lld ch %d frames %d pairs %d chunks %d new_chunks %d\n", (long long)n, n_ch, n_frames, pl.n_pairs, pl.n_chunks, new_chunks);
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
    const size_t nxy = (size_t)max_chunks * n_ch * w4::NB;
    float2* xs = dalloc<float2>((size_t)pl.n_pairs * w4::N);
    float* px = dalloc<float>((size_t)pl.n_pairs * w4::NB);
    float* psx = dalloc<float>((size_t)max_chunks * w4::NB);
    float2* pxy = dalloc<float2>(nxy);
    float* pyy = dalloc<float>(nxy);
    w4::Args ax{x, n, n, 1, hop, n_frames, pl.n_pairs, 1, pl.n_chunks, pl.ppc, win, twt, xs, px, pxy, pyy, psx};
    w4::Args ay = ax;
    ay.sig = y;
    ay.n_ch = n_ch;
    w4::Args ay3 = ay;
    ay3.n_chunks = new_chunks;
    if (!getenv("EVEN_SPLIT")) w4::place_remainder(ay3, n_ch);
    printf("use_plus %d mask %08x\n", ay3.use_plus, ay3.plus[0]);
    CK(hipFuncSetAttribute((const void*)w4::k_y<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES_2));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto run_old = [&]() {
        hipLaunchKernelGGL((w4::k_x<true>), dim3(pl.n_pairs), dim3(256), w4::LDS_BYTES, st, ax);
        hipLaunchKernelGGL((w4::k_y<true, false>), dim3(pl.n_chunks * n_ch), dim3(256), w4::LDS_BYTES_2, st, ay);
    };
    auto run_new = [&]() {
        hipLaunchKernelGGL(w4::k_x3, dim3(pl.n_pairs), dim3(256), w4::LDS3_BYTES, st, ax);
        hipLaunchKernelGGL((w4::k_y3<false>), dim3(new_chunks * n_ch), dim3(256), w4::LDS3_BYTES, st, ay3);
    };
    std::vector<float> o_xy(2 * nxy), o_yy(nxy), o_sx((size_t)max_chunks * w4::NB), n_xy(2 * nxy), n_yy(nxy), n_sx(o_sx.size());
    run_old();
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    CK(hipMemcpy(o_xy.data(), pxy, nxy * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o_yy.data(), pyy, nxy * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o_sx.data(), psx, o_sx.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(pxy, 0, nxy * 8));
    CK(hipMemset(pyy, 0, nxy * 4));
    CK(hipMemset(psx, 0, o_sx.size() * 4));
    run_new();
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    CK(hipMemcpy(n_xy.data(), pxy, nxy * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(n_yy.data(), pyy, nxy * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(n_sx.data(), psx, n_sx.size() * 4, hipMemcpyDeviceToHost));
    // sum the chunk partials (the chunk counts differ) and compare per channel
    auto reduce = [&](const std::vector<float>& v, int chunks, int width, int nc) {
        std::vector<float> r((size_t)nc * w4::NB * width, 0.f);
        std::vector<double> acc(r.size(), 0.0);
        for (int q = 0; q < chunks; ++q)
            for (size_t i = 0; i < acc.size(); ++i) acc[i] += v[(size_t)q * acc.size() + i];
        for (size_t i = 0; i < acc.size(); ++i) r[i] = (float)acc[i];
        return r;
    };
    auto a_xy = reduce(n_xy, new_chunks, 2, n_ch), b_xy = reduce(o_xy, pl.n_chunks, 2, n_ch);
    auto a_yy = reduce(n_yy, new_chunks, 1, n_ch), b_yy = reduce(o_yy, pl.n_chunks, 1, n_ch);
    auto a_sx = reduce(n_sx, new_chunks, 1, 1), b_sx = reduce(o_sx, pl.n_chunks, 1, 1);
    printf("new vs old: pxy %.3e  pyy %.3e  psx %.3e\n", relmax(a_xy, b_xy), relmax(a_yy, b_yy), relmax(a_sx, b_sx));
    double worst = 0;
    for (int c = 0; c < n_ch; ++c) {
        std::vector<float> a(a_yy.begin() + (size_t)c * w4::NB, a_yy.begin() + (size_t)(c + 1) * w4::NB),
            b(b_yy.begin() + (size_t)c * w4::NB, b_yy.begin() + (size_t)(c + 1) * w4::NB);
        worst = std::max(worst, relmax(a, b));
    }
    printf("worst per-channel pyy %.3e\n", worst);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20;
    for (int r = 0; r < rounds; ++r) {
        float ms[2][2];
        for (int v = 0; v < 2; ++v) {
            for (int which = 0; which < 2; ++which) {  // 0: main kernel only, 1: x + main
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) {
                    if (v == 0) {
                        if (which) hipLaunchKernelGGL((w4::k_x<true>), dim3(pl.n_pairs), dim3(256), w4::LDS_BYTES, st, ax);
                        hipLaunchKernelGGL((w4::k_y<true, false>), dim3(pl.n_chunks * n_ch), dim3(256), w4::LDS_BYTES_2, st, ay);
                    } else {
                        if (which) hipLaunchKernelGGL(w4::k_x3, dim3(pl.n_pairs), dim3(256), w4::LDS3_BYTES, st, ax);
                        hipLaunchKernelGGL((w4::k_y3<false>), dim3(new_chunks * n_ch), dim3(256), w4::LDS3_BYTES, st, ay3);
                    }
                }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[v][which], e0, e1));
            }
        }
        printf("round %d: old main %.1f us (x+main %.1f) | new main %.1f us (x+main %.1f)\n", r, 1e3 * ms[0][0] / iters,
               1e3 * ms[0][1] / iters, 1e3 * ms[1][0] / iters, 1e3 * ms[1][1] / iters);
    }
#if W4_TIMING
    unsigned long long tm[16], z[16] = {};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(w4::w3_timing), z, sizeof(z)));
    for (int i = 0; i < 5; ++i) run_new();
    CK(hipStreamSynchronize(st));
    CK(hipMemcpyFromSymbol(tm, HIP_SYMBOL(w4::w3_timing), sizeof(tm)));
    const char* names[] = {"window (LDS reads)", "dft16#1 + pass-1 stores", "barrier 2 (+store drain)", "pass-2 read issue",
                           "dft16#2 stage A (+read wait, loads)", "dft16#2 stage B + transpose stores + xs loads", "wave sync + pass-3 read issue",
                           "dft16#3 stage A (+read wait)", "barrier 1", "dft16#3 rest", "accumulate (+xs wait)"};
    double tot = 0;
    for (int i = 0; i < 11; ++i) tot += (double)tm[i];
    printf("stamps: %llu iterations, %.0f cycles / iteration\n", tm[15], tot / tm[15]);
    for (int i = 0; i < 11; ++i) printf("  %-52s %8.0f %5.1f %%\n", names[i], (double)tm[i] / tm[15], 100.0 * tm[i] / tot);
    {
        const int nb = new_chunks * n_ch;
        std::vector<unsigned long long> life((size_t)4096 * 8);
        CK(hipMemcpyFromSymbol(life.data(), HIP_SYMBOL(w4::w3_life), life.size() * 8));
        unsigned long long r_min = ~0ull, r_max = 0;
        for (int b = 0; b < nb; ++b) {
            r_min = std::min(r_min, life[8 * b + 2]);
            r_max = std::max(r_max, life[8 * b + 3]);
        }
        double cyc = 0, rt = 0;
        int late = 0;
        std::vector<double> starts;
        for (int b = 0; b < nb; ++b) {
            cyc += (double)(life[8 * b + 1] - life[8 * b]);
            rt += (double)(life[8 * b + 3] - life[8 * b + 2]);
            const double st_us = (double)(life[8 * b + 2] - r_min) / 100.0;
            if (st_us > 5.0) ++late;
            starts.push_back(st_us);
        }
        std::sort(starts.begin(), starts.end());
        std::vector<double> lt, pro, loop, epi;
        double xcd_lt[8] = {};
        for (int b = 0; b < nb; ++b) {
            lt.push_back((double)(life[8 * b + 3] - life[8 * b + 2]) / 100.0);
            pro.push_back((double)(life[8 * b + 4] - life[8 * b + 2]) / 100.0);
            loop.push_back((double)(life[8 * b + 5] - life[8 * b + 4]) / 100.0);
            epi.push_back((double)(life[8 * b + 3] - life[8 * b + 5]) / 100.0);
            xcd_lt[b & 7] += lt.back() / (nb / 8);
        }
        auto pct = [](std::vector<double> v, const char* nm) {
            std::sort(v.begin(), v.end());
            size_t n = v.size();
            printf("  %-10s min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us\n", nm, v[0], v[n / 10], v[n / 2], v[9 * n / 10], v[n - 1]);
        };
        pct(lt, "lifetime");
        pct(pro, "prologue");
        pct(loop, "loop");
        pct(epi, "epilogue");
        {   // per CU: do the three workgroups of a CU end together, and do CUs differ?
            std::map<unsigned, std::vector<double>> by_cu;
            for (int b = 0; b < nb; ++b) {
                const unsigned hw = (unsigned)life[8 * b + 6], xcc = (unsigned)life[8 * b + 7] & 0xf;
                const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
                by_cu[(xcc << 8) | (se << 5) | (sh << 4) | cu].push_back((double)(life[8 * b + 3] - r_min) / 100.0);
            }
            std::vector<double> cu_end, cu_spread;
            size_t n3 = 0;
            for (auto& kv : by_cu) {
                auto& v = kv.second;
                std::sort(v.begin(), v.end());
                cu_end.push_back(v.back());
                cu_spread.push_back(v.back() - v.front());
                n3 += v.size() == 3;
            }
            printf("  CUs seen %zu (with exactly three workgroups: %zu)\n", by_cu.size(), n3);
            pct(cu_end, "CU end");
            pct(cu_spread, "in-CU gap");
        }
        printf("  mean lifetime by blockIdx & 7:");
        for (int x = 0; x < 8; ++x) printf(" %.1f", xcd_lt[x]);
        printf("\n");
        printf("workgroups %d: kernel span %.1f us; mean lifetime %.1f us = %.0f cycles -> clock %.2f GHz; started later than 5 us: %d (median start %.2f us, max %.2f us)\n",
               nb, (double)(r_max - r_min) / 100.0, rt / nb / 100.0, cyc / nb, cyc / rt / 10.0, late, starts[nb / 2], starts.back());
    }
#endif
    return 0;
}
