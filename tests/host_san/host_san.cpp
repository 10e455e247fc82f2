// Host-only sanitizer run of the boundary's marshalling code (dsptoolbox_amd/csrc/host_marshal.hpp):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -pthread host_san.cpp
// The pipelines run with a memcpy transport standing in for the asynchronous device copies: the
// "device" is a host array, every transfer completes at once, `wait` checks the chunk bookkeeping.
// Exercised: ragged chunk tails, one to many channels, row pitches wider than the data, thread counts
// 1 ... 7 (more threads than 256-sample tiles included), zero-length inputs, the too-many-channels refusal.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../dsptoolbox_amd/csrc/host_marshal.hpp"
#include "../../dsptoolbox_amd/csrc/size_guards.hpp"

static int failures = 0;
#define EXPECT(cond)                                                    \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::fprintf(stderr, "FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                 \
        }                                                               \
    } while (0)

struct MemTransport {
    bool busy[2] = {false, false};
    int transfers = 0, waits_on_idle = 0;
    bool wait(int b) {
        if (!busy[b]) ++waits_on_idle;
        busy[b] = false;
        return true;
    }
    bool copy2d(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t rows, int b) {
        if (busy[b]) return false;  // a chunk must not be reused before its transfer was waited for
        for (size_t r = 0; r < rows; ++r) std::memcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
        busy[b] = true;
        ++transfers;
        return true;
    }
    bool h2d_2d(float* d, size_t dp, const float* s, size_t sp, size_t w, size_t r, int b) { return copy2d(d, dp, s, sp, w, r, b); }
    bool d2h_2d(float* d, size_t dp, const float* s, size_t sp, size_t w, size_t r, int b) { return copy2d(d, dp, s, sp, w, r, b); }
    bool d2h(float* d, const float* s, size_t bytes, int b) { return copy2d(d, bytes, s, bytes, bytes, 1, b); }
    bool h2d(float* d, const float* s, size_t bytes, int b) { return copy2d(d, bytes, s, bytes, bytes, 1, b); }
};

int main() {
    // ---- k_welch_finish: which partial slabs leave the 32-bit raw-buffer descriptors (W = 2^24 with 64
    // output channels is sy * 8 = 4 294 967 808 bytes and used to wrap to 512)
    {
        const int64_t nb24 = ((int64_t)1 << 23) + 1, nb23 = ((int64_t)1 << 22) + 1, nb4096 = 2049;
        EXPECT(!welch_finish_wide_slab(1 * nb4096, 64 * nb4096));    // the headline shape
        EXPECT(!welch_finish_wide_slab(1 * nb24, 63 * nb24));        // 4 227 858 936 bytes: below 2^32 - 16
        EXPECT(welch_finish_wide_slab(1 * nb24, 64 * nb24));         // 4 294 967 808: wraps
        EXPECT(!welch_finish_wide_slab(1 * nb23, 127 * nb23));
        EXPECT(welch_finish_wide_slab(1 * nb23, 128 * nb23));
        EXPECT(welch_finish_wide_slab(128 * nb24, 0));               // auto spectra: sx * 4
        EXPECT(!welch_finish_wide_slab(127 * nb24, 0));
        EXPECT(welch_finish_wide_slab(0, ((int64_t)0xfffffff0 + 7) / 8));
        EXPECT(!welch_finish_wide_slab(0, (int64_t)0xfffffff0 / 8 - 1));
    }
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> ud(-1.0, 1.0);
    // ---- the plain helpers, every thread count
    for (int n_ch : {1, 2, 3, 7, 64}) {
        for (int64_t n : {(int64_t)0, (int64_t)1, (int64_t)255, (int64_t)256, (int64_t)257, (int64_t)5000}) {
            const int64_t ld = n + 5;
            std::vector<double> src((size_t)n * n_ch);
            for (auto& v : src) v = ud(rng);
            for (int threads = 1; threads <= 7; threads += 2) {
                std::vector<float> planar((size_t)n_ch * ld, -7.f);
                dshost::planar_f32(src.data(), n, n_ch, planar.data(), ld, threads);
                bool ok = true;
                for (int c = 0; c < n_ch && ok; ++c) {
                    for (int64_t i = 0; i < n; ++i) ok = ok && planar[(size_t)c * ld + i] == (float)src[(size_t)i * n_ch + c];
                    for (int64_t i = n; i < ld; ++i) ok = ok && planar[(size_t)c * ld + i] == -7.f;  // the pitch gap is untouched
                }
                EXPECT(ok);
                std::vector<double> back((size_t)n * n_ch, 9.0);
                dshost::interleave_f64(planar.data(), n, n_ch, ld, back.data(), threads);
                ok = true;
                for (size_t i = 0; i < back.size(); ++i) ok = ok && back[i] == (double)(float)src[i];
                EXPECT(ok);
                std::vector<double> wide((size_t)n_ch * ld);
                dshost::widen_f64(planar.data(), (int64_t)planar.size(), wide.data(), threads);
                ok = true;
                for (size_t i = 0; i < wide.size(); ++i) ok = ok && wide[i] == (double)planar[i];
                EXPECT(ok);
            }
        }
    }
    EXPECT(dshost::host_threads(8, 100) == 1 && dshost::host_threads(8, (int64_t)1 << 24) == 8);
    // ---- the pipelines: a staging chunk of 8 KB holds 2048 samples of one channel, 256 of seven
    const size_t pin_bytes = 8192;
    std::vector<float> p0(pin_bytes / 4), p1(pin_bytes / 4);
    float* pin[2] = {p0.data(), p1.data()};
    for (int n_ch : {1, 2, 7}) {
        for (int64_t n : {(int64_t)0, (int64_t)100, (int64_t)256, (int64_t)1000, (int64_t)2048, (int64_t)5000, (int64_t)12345}) {
            const int64_t ld = n + 3;
            std::vector<double> src((size_t)n * n_ch);
            for (auto& v : src) v = ud(rng);
            std::vector<float> dev((size_t)n_ch * ld + 1, -3.f);
            MemTransport up;
            EXPECT(dshost::upload_planar(up, pin, pin_bytes, src.data(), n, n_ch, dev.data(), ld));
            bool ok = true;
            for (int c = 0; c < n_ch; ++c)
                for (int64_t i = 0; i < n; ++i) ok = ok && dev[(size_t)c * ld + i] == (float)src[(size_t)i * n_ch + c];
            EXPECT(ok && dev.back() == -3.f);
            const int64_t cs = dshost::chunk_samples(pin_bytes, n_ch);
            EXPECT(up.transfers == (n + cs - 1) / cs);
            std::vector<double> back((size_t)n * n_ch + 1, 5.0);
            MemTransport down;
            EXPECT(dshost::download_interleave(down, pin, pin_bytes, dev.data(), n, n_ch, ld, back.data()));
            ok = back.back() == 5.0;
            for (size_t i = 0; i + 1 < back.size(); ++i) ok = ok && back[i] == (double)(float)src[i];
            EXPECT(ok && down.waits_on_idle == 0);
            std::vector<double> flat(dev.size());
            MemTransport dw;
            EXPECT(dshost::download_widen(dw, pin, pin_bytes, dev.data(), (int64_t)dev.size(), flat.data()));
            ok = true;
            for (size_t i = 0; i < flat.size(); ++i) ok = ok && flat[i] == (double)dev[i];
            EXPECT(ok && dw.waits_on_idle == 0);
            // contiguous float64 -> float32 (complex128 -> complex64) through the same chunks
            std::vector<float> narrow(src.size() + 1, -2.f);
            MemTransport un;
            EXPECT(dshost::upload_narrow(un, pin, pin_bytes, src.data(), (int64_t)src.size(), narrow.data()));
            ok = narrow.back() == -2.f;
            for (size_t i = 0; i < src.size(); ++i) ok = ok && narrow[i] == (float)src[i];
            EXPECT(ok && un.transfers == (int)((src.size() * 4 + pin_bytes - 1) / pin_bytes));
        }
    }
    {   // more channels than a chunk holds 256 samples of: refused, nothing written
        MemTransport t;
        std::vector<double> src(9 * 300, 1.0);
        std::vector<float> dev(9 * 300, 0.f);
        EXPECT(!dshost::upload_planar(t, pin, pin_bytes, src.data(), 300, 9, dev.data(), 300) && t.transfers == 0);
    }
    if (failures) {
        std::fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    std::puts("host_san: ok");
    return 0;
}
