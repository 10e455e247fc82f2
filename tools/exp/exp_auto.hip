// Dev harness (round 4): the headline transform loop at THREE against FOUR workgroups per CU.
// The auto-spectrum variant k_y3<true> (no input spectra, no cross sums) fits 128 registers; with the window
// taken out of LDS (-DW4_WIN_GLOBAL=1: buffer loads, L1-resident) a workgroup needs 36 KB, so four fit a CU.
//   three per CU (what ships):  hipcc ... -o tools/exp/exp_auto3 tools/exp/exp_auto.hip
//   three per CU, window by loads: ... -DW4_WIN_GLOBAL=1 -o tools/exp/exp_auto3g
//   four per CU:                ... -DW4_OCC=4 -DW4_TW6=1 -DW4_WIN_GLOBAL=1 -o tools/exp/exp_auto4
//   tools/exp/exp_autoN [n_samples] [n_ch] [rounds] [CROSS]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../dsptoolbox_amd/csrc/kernels_welch4096w.hpp"

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

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int n_ch = argc > 2 ? atoi(argv[2]) : 64;
    const int rounds = argc > 3 ? atoi(argv[3]) : 6;
    const bool cross = argc > 4 && !strcmp(argv[4], "CROSS");
    const int hop = 2048, n_frames = (int)((n + hop - 1) / hop), n_pairs = (n_frames + 1) / 2;
    const int n_chunks = getenv("NEWCHUNKS") ? atoi(getenv("NEWCHUNKS")) : std::max(1, std::min(n_pairs, (W4_OCC * 256) / n_ch));
    const int lds = W4_WIN_GLOBAL ? w4::LDS3G_BYTES : w4::LDS3_BYTES;
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
    const size_t nxy = (size_t)n_chunks * n_ch * w4::NB;
    float2* xs = dalloc<float2>((size_t)n_pairs * w4::N);
    float* px = dalloc<float>((size_t)n_pairs * w4::NB);
    float* psx = dalloc<float>((size_t)n_chunks * w4::NB);
    float2* pxy = dalloc<float2>(nxy);
    float* pyy = dalloc<float>(nxy);
    w4::Args ax{x, n, n, 1, hop, n_frames, n_pairs, 1, n_chunks, (n_pairs + n_chunks - 1) / n_chunks, win, twt, xs, px, pxy, pyy, psx};
    ax.n_cx = 1;
    w4::Args ay = ax;
    ay.sig = y;
    ay.n_ch = n_ch;
    w4::place_remainder(ay, n_ch);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int per_cu = 0;
    if (cross)
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, w4::k_y3<false>, 256, lds));
    else
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, w4::k_y3<true>, 256, lds));
    printf("build: W4_OCC %d  window %s  lds %d B  %s | chunks %d grid %d | workgroups per CU by the occupancy query: %d\n", W4_OCC,
           W4_WIN_GLOBAL ? "buffer loads" : "LDS", lds, cross ? "cross (k_y3<false>)" : "auto (k_y3<true>)", n_chunks, n_chunks * n_ch,
           per_cu);
    hipLaunchKernelGGL(w4::k_x3, dim3(n_pairs), dim3(256), w4::LDS3_BYTES, st, ax);
    auto run = [&]() {
        if (cross)
            hipLaunchKernelGGL((w4::k_y3<false>), dim3(n_chunks * n_ch), dim3(256), lds, st, ay);
        else
            hipLaunchKernelGGL((w4::k_y3<true>), dim3(n_chunks * n_ch), dim3(256), lds, st, ay);
    };
    run();
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    {   // chunk-summed auto spectra: a few values and a checksum, to compare builds by eye
        std::vector<float> h(nxy);
        CK(hipMemcpy(h.data(), pyy, nxy * 4, hipMemcpyDeviceToHost));
        std::vector<double> s((size_t)n_ch * w4::NB, 0.0);
        for (int q = 0; q < n_chunks; ++q)
            for (size_t i = 0; i < s.size(); ++i) s[i] += h[(size_t)q * s.size() + i];
        double tot = 0;
        for (double v : s) tot += v;
        printf("pyy: sum %.9e  [ch0 bin1] %.7e [ch0 bin2048] %.7e [ch63 bin1000] %.7e\n", tot, s[1], s[2048],
               s[(size_t)(n_ch - 1) * w4::NB + 1000]);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20;
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("round %d: %.1f us per launch (%d transforms: %.2f ns each)\n", r, 1e3 * ms / iters, n_pairs * n_ch,
               1e6 * ms / iters / ((double)n_pairs * n_ch));
    }
    return 0;
}
