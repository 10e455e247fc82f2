// Host side of the boundary, free of any device API: thread-parallel float64 <-> float32 marshalling
// between the reference's (samples, channels) float64 arrays and the library's planar float32 layout,
// and the double-buffered pipelines that stream it through two pinned staging chunks.  The pipelines
// are templates over a small transport policy (the asynchronous copies and their completion events):
// api.hip instantiates them with HIP calls, tests/host_san/ with plain memcpy -- so that this code runs
// under -fsanitize=address,undefined on a CPU box.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

namespace dshost {

inline int host_threads(int threads, int64_t work_items) {
    if (threads <= 0) {
        unsigned hc = std::thread::hardware_concurrency();
        threads = (int)std::min<unsigned>(16u, hc ? hc : 1u);
        // (host-side casts have no context: read once per process)
        static const int forced = [] {
            const char* e = getenv("DSPTOOLBOX_AMD_HOST_THREADS");
            return e ? std::max(1, atoi(e)) : 0;
        }();
        if (forced) threads = forced;
    }
    // below ~1 M elements a thread start costs more than it saves
    const int64_t by_work = std::max<int64_t>(1, work_items / (1 << 20));
    return (int)std::min<int64_t>(threads, by_work);
}
template <typename F>
inline void host_parallel(int threads, int64_t n, F body) {  // body(begin, end) over [0, n)
    if (threads <= 1) {
        body((int64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    const int64_t per = ((n + threads - 1) / threads + 255) & ~(int64_t)255;
    for (int t = 0; t < threads; ++t) {
        const int64_t b = (int64_t)t * per, e = std::min(n, b + per);
        if (b >= e) break;
        pool.emplace_back([=]() { body(b, e); });
    }
    for (auto& th : pool) th.join();
}

// src (n_samples, n_ch) float64, C order  ->  dst[c * ld + n] float32
inline void planar_f32(const double* src, int64_t n_samples, int n_ch, float* dst, int64_t ld, int threads) {
    host_parallel(threads, n_samples, [=](int64_t b, int64_t e) {
        constexpr int64_t TILE = 256;  // samples per tile: TILE x n_ch doubles stay in the cache
        for (int64_t s0 = b; s0 < e; s0 += TILE) {
            const int64_t s1 = std::min(e, s0 + TILE);
            for (int c = 0; c < n_ch; ++c) {
                float* __restrict__ d = dst + (int64_t)c * ld;
                const double* __restrict__ s = src + c;
                for (int64_t n = s0; n < s1; ++n) d[n] = (float)s[n * n_ch];
            }
        }
    });
}
inline void widen_f64(const float* src, int64_t n, double* dst, int threads) {
    host_parallel(threads, n, [=](int64_t b, int64_t e) {
        const float* __restrict__ s = src;
        double* __restrict__ d = dst;
        for (int64_t i = b; i < e; ++i) d[i] = (double)s[i];
    });
}
inline void narrow_f32(const double* src, int64_t n, float* dst, int threads) {
    host_parallel(threads, n, [=](int64_t b, int64_t e) {
        const double* __restrict__ s = src;
        float* __restrict__ d = dst;
        for (int64_t i = b; i < e; ++i) d[i] = (float)s[i];
    });
}
// src[c * ld + n] float32  ->  dst (n_samples, n_ch) float64, C order
inline void interleave_f64(const float* src, int64_t n_samples, int n_ch, int64_t ld, double* dst, int threads) {
    host_parallel(threads, n_samples, [=](int64_t b, int64_t e) {
        constexpr int64_t TILE = 256;
        for (int64_t s0 = b; s0 < e; s0 += TILE) {
            const int64_t s1 = std::min(e, s0 + TILE);
            for (int c = 0; c < n_ch; ++c) {
                const float* __restrict__ s = src + (int64_t)c * ld;
                double* __restrict__ d = dst + c;
                for (int64_t n = s0; n < s1; ++n) d[n * n_ch] = (double)s[n];
            }
        }
    });
}

// samples of every channel that fit one staging chunk (a multiple of 256; 0: too many channels)
inline int64_t chunk_samples(size_t pin_bytes, int n_ch) {
    return (int64_t)(pin_bytes / ((size_t)n_ch * sizeof(float))) & ~(int64_t)255;
}

// Transport policy T:
//   bool wait(int b)                                   the last transfer that used staging chunk b is done
//   bool h2d_2d(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t rows, int b)
//   bool d2h_2d(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t rows, int b)
//   bool d2h(float* dst, const float* src, size_t bytes, int b)
// (pitches / width in bytes; each transfer is asynchronous and tagged with the chunk it uses).

// float64 (n_samples, n_ch) on the host -> planar float32 rows of pitch ld on the device: chunk k is cast
// by the host threads into staging chunk k & 1 while the copy of chunk k - 1 is in flight.
template <typename T>
inline bool upload_planar(T& tr, float* const pin[2], size_t pin_bytes, const double* src, int64_t n_samples,
                          int n_ch, float* dst_dev, int64_t ld) {
    const int64_t cs = chunk_samples(pin_bytes, n_ch);
    if (cs < 256) return false;
    int k = 0;
    for (int64_t s0 = 0; s0 < n_samples; s0 += cs, ++k) {
        const int b = k & 1;
        const int64_t cn = std::min(cs, n_samples - s0);
        if (!tr.wait(b)) return false;
        planar_f32(src + s0 * n_ch, cn, n_ch, pin[b], cn, host_threads(0, cn * n_ch));
        if (!tr.h2d_2d(dst_dev + s0, (size_t)ld * sizeof(float), pin[b], (size_t)cn * sizeof(float),
                       (size_t)cn * sizeof(float), (size_t)n_ch, b))
            return false;
    }
    return true;
}
// planar float32 rows on the device -> float64 (n_samples, n_ch) on the host; the copy of chunk k + 1 is in
// flight while the host threads widen and interleave chunk k.
template <typename T>
inline bool download_interleave(T& tr, float* const pin[2], size_t pin_bytes, const float* src_dev, int64_t n_samples,
                                int n_ch, int64_t ld, double* dst) {
    const int64_t cs = chunk_samples(pin_bytes, n_ch);
    if (cs < 256) return false;
    const int64_t n_chunks = (n_samples + cs - 1) / cs;
    auto issue = [&](int64_t k) {
        const int64_t s0 = k * cs, cn = std::min(cs, n_samples - s0);
        return tr.d2h_2d(pin[k & 1], (size_t)cn * sizeof(float), src_dev + s0, (size_t)ld * sizeof(float),
                         (size_t)cn * sizeof(float), (size_t)n_ch, (int)(k & 1));
    };
    if (n_chunks > 0 && !issue(0)) return false;
    for (int64_t k = 0; k < n_chunks; ++k) {
        if (k + 1 < n_chunks && !issue(k + 1)) return false;
        if (!tr.wait((int)(k & 1))) return false;
        const int64_t s0 = k * cs, cn = std::min(cs, n_samples - s0);
        interleave_f64(pin[k & 1], cn, n_ch, cn, dst + s0 * n_ch, host_threads(0, cn * n_ch));
    }
    return true;
}
// contiguous float64 on the host -> float32 on the device (complex128 -> complex64 element by element): chunk k is
// cast into staging chunk k & 1 while the copy of chunk k - 1 is in flight.
//   T also needs:  bool h2d(float* dst, const float* src, size_t bytes, int b)
template <typename T>
inline bool upload_narrow(T& tr, float* const pin[2], size_t pin_bytes, const double* src, int64_t n, float* dst_dev) {
    const int64_t cs = (int64_t)(pin_bytes / sizeof(float));
    int k = 0;
    for (int64_t s0 = 0; s0 < n; s0 += cs, ++k) {
        const int b = k & 1;
        const int64_t cn = std::min(cs, n - s0);
        if (!tr.wait(b)) return false;
        narrow_f32(src + s0, cn, pin[b], host_threads(0, cn));
        if (!tr.h2d(dst_dev + s0, pin[b], (size_t)cn * sizeof(float), b)) return false;
    }
    return true;
}
// contiguous float32 on the device -> float64 on the host
template <typename T>
inline bool download_widen(T& tr, float* const pin[2], size_t pin_bytes, const float* src_dev, int64_t n, double* dst) {
    const int64_t cs = (int64_t)(pin_bytes / sizeof(float));
    const int64_t n_chunks = (n + cs - 1) / cs;
    auto issue = [&](int64_t k) {
        const int64_t s0 = k * cs, cn = std::min(cs, n - s0);
        return tr.d2h(pin[k & 1], src_dev + s0, (size_t)cn * sizeof(float), (int)(k & 1));
    };
    if (n_chunks > 0 && !issue(0)) return false;
    for (int64_t k = 0; k < n_chunks; ++k) {
        if (k + 1 < n_chunks && !issue(k + 1)) return false;
        if (!tr.wait((int)(k & 1))) return false;
        const int64_t s0 = k * cs, cn = std::min(cs, n - s0);
        widen_f64(pin[k & 1], cn, dst + s0, host_threads(0, cn));
    }
    return true;
}

}  // namespace dshost
