// C-ABI of dsptoolbox_amd (see include/dsptoolbox_amd.h): context, memory,
// plan (twiddle) cache, launch logic.  gfx950 only.
#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dsptoolbox_amd.h"
#include "config.hpp"
#include "host_marshal.hpp"
#include "kernels_bigfft.hpp"
#include "kernels_bluestein.hpp"
#include "kernels_finish.hpp"
#include "kernels_csm_b3.hpp"
#include "kernels_generic.hpp"
#include "kernels_welch4096.hpp"
#include "kernels_welch4096w.hpp"
#include "kernels_fir16k.hpp"
#include "kernels_fir4k.hpp"
#include "kernels_stft4096.hpp"
#include "kernels_deconv8k.hpp"
#include "kernels_stft1024.hpp"
#include "kernels_welch1024.hpp"
#include "kernels_welch8192.hpp"
#include "kernels_welch16384.hpp"
#include "kernels_welch_long.hpp"
#include "kernels_stft_long.hpp"
#include "kernels_welch2048h.hpp"
#include "kernels_istft_long.hpp"
#include "kernels_welch_f64.hpp"
#include "kernels_stft_any.hpp"
#include "kernels_fir_stream.hpp"
#include "kernels_freqz.hpp"

using namespace dsk;

static thread_local std::string g_err;

struct ds_ctx {
    int device = 0;
    ds_config cfg;  // every DSPTOOLBOX_AMD_* switch, read once by ds_init (config.hpp)
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;  // second stream for a kernel that may run beside the main one
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_chunk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // csm_chunked
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    std::map<int, float2*> tw;  // twiddle tables by length
    // Bluestein chirp-filter spectra by (L, M): a least-recently-used cache under a byte cap (a
    // table is M float2, up to 128 MB; recordings of ever-changing lengths must not pin one each)
    struct BlueEntry {
        float2* ptr;
        size_t bytes;
        uint64_t stamp;
    };
    std::map<std::pair<int64_t, int64_t>, BlueEntry> blue;
    size_t blue_bytes = 0;
    uint64_t blue_clock = 0;
    float2* w4_tables = nullptr;  // welch4096::host_tables()
    float2* stft_dif_tw[2] = {nullptr, nullptr};  // stft4k::host_twiddles(8192 / 16384)
    float2* fir16k_tables = nullptr;  // fir16k::host_tables()
    float2* wl_tables[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // welchl::host_tables(R), R = 2, 4, ..., 64
    float2* deconv8k_tables = nullptr;  // deconv8k::host_tables()
    float2* deconv_rperm = nullptr;     // deconv8k::k_rperm's output (64 KB), rewritten by every ds_deconv_dev call that uses it
    int n_cu = 0;                       // compute units of the device (persistent grids)
    float2* stft1k_tables = nullptr;  // stft1k::host_tables<1024>()
    float2* stft_wave_tables[3] = {nullptr, nullptr, nullptr};  // stft1k::host_tables<512>(), <256>(), <2048>()
    void* ws = nullptr;         // kernel workspace (spectra, partials)
    size_t ws_bytes = 0;
    void* io = nullptr;  // staging for the host-pointer entry points
    size_t io_bytes = 0;
    void* aux = nullptr;  // small persistent scratch (combined FIR taps, cascade ping-pong buffer)
    void* frames = nullptr;  // time-domain frames of the inverse STFT with a non-power-of-two length
    size_t frames_bytes = 0;
    void* pin[2] = {nullptr, nullptr};  // pinned host chunks of the fused float64 upload (double buffer)
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};
    size_t aux_bytes = 0;
    // RCCL (dlopen'ed lazily)
    void* rccl = nullptr;
    void* comm = nullptr;
    // per-kernel HIP-event timing (ds_profile_*)
    bool prof = false;
    struct ProfRec {
        const char* name;
        hipEvent_t a, b;
    };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    std::string prof_text;
    std::string prof_only;  // non-empty: only launches of this kernel name are bracketed
    int prof_stride = 1;    // bracket every prof_stride-th matching launch (an event pair costs ~3 us of stream time)
    long prof_seen = 0;
    std::set<const char*> routes;                     // launch names ("group@variant" literals) since the last ds_routes()
    std::map<const char*, std::string> group_names;   // "group@variant" literal -> "group"
    std::string routes_text;
};

static int fail(ds_ctx* c, int code, const std::string& msg) {
    g_err = msg;
    if (c) c->err = msg;
    return code;
}
#define HIPCHK(c, expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(c, DS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define CHK(expr)                 \
    do {                          \
        int r_ = (expr);          \
        if (r_ != DS_OK) return r_; \
    } while (0)

static bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }
static const int kMaxFft = 16384, kMinFft = 8;
static const int64_t kMaxBigFft = (int64_t)1 << 24;  // four-step path (kernels_bigfft.hpp)

extern "C" int ds_version(void) { return 100; }

// ---- host marshalling helpers (no device work): csrc/host_marshal.hpp --------
using dshost::host_parallel;
using dshost::host_threads;
extern "C" int ds_host_planar_f32(const double* src, int64_t n_samples, int n_ch, float* dst, int64_t ld,
                                  int threads) {
    if (!src || !dst || n_samples < 0 || n_ch <= 0 || ld < n_samples)
        return fail(nullptr, DS_ERR_ARG, "ds_host_planar_f32: bad argument");
    dshost::planar_f32(src, n_samples, n_ch, dst, ld, host_threads(threads, n_samples * n_ch));
    return DS_OK;
}
extern "C" int ds_host_widen_f64(const float* src, int64_t n, double* dst, int threads) {
    if (!src || !dst || n < 0) return fail(nullptr, DS_ERR_ARG, "ds_host_widen_f64: bad argument");
    dshost::widen_f64(src, n, dst, host_threads(threads, n));
    return DS_OK;
}
extern "C" int ds_host_interleave_f64(const float* src, int64_t n_samples, int n_ch, int64_t ld, double* dst,
                                      int threads) {
    if (!src || !dst || n_samples < 0 || n_ch <= 0 || ld < n_samples)
        return fail(nullptr, DS_ERR_ARG, "ds_host_interleave_f64: bad argument");
    dshost::interleave_f64(src, n_samples, n_ch, ld, dst, host_threads(threads, n_samples * n_ch));
    return DS_OK;
}
extern "C" int ds_max_fft_len(void) { return kMaxFft; }
extern "C" int ds_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ds_init(int device, ds_ctx** out) {
    if (!out) return fail(nullptr, DS_ERR_ARG, "ds_init: out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, DS_ERR_HIP, "ds_init: no HIP device visible");
    if (device < 0 || device >= n) return fail(nullptr, DS_ERR_ARG, "ds_init: bad device index");
    ds_ctx* c = new ds_ctx();
    c->device = device;
    c->cfg = ds_config::from_env();
    HIPCHK(c, hipSetDevice(device));
    HIPCHK(c, hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device));
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    HIPCHK(c, hipEventCreate(&c->ev0));
    HIPCHK(c, hipEventCreate(&c->ev1));
    *out = c;
    return DS_OK;
}

extern "C" int ds_comm_destroy(ds_ctx* c);

extern "C" void ds_destroy(ds_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    ds_comm_destroy(c);
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->tw) (void)hipFree(kv.second);
    if (c->w4_tables) (void)hipFree(c->w4_tables);
    for (float2* t : c->stft_dif_tw)
        if (t) (void)hipFree(t);
    if (c->fir16k_tables) (void)hipFree(c->fir16k_tables);
    for (auto t : c->wl_tables)
        if (t) (void)hipFree(t);
    if (c->deconv8k_tables) (void)hipFree(c->deconv8k_tables);
    if (c->deconv_rperm) (void)hipFree(c->deconv_rperm);
    if (c->stft1k_tables) (void)hipFree(c->stft1k_tables);
    for (float2* t : c->stft_wave_tables)
        if (t) (void)hipFree(t);
    for (auto& kv : c->blue) (void)hipFree(kv.second.ptr);
    for (auto& r : c->prof_recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto& e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->ws) (void)hipFree(c->ws);
    if (c->io) (void)hipFree(c->io);
    if (c->aux) (void)hipFree(c->aux);
    if (c->frames) (void)hipFree(c->frames);
    for (int i = 0; i < 2; ++i) {
        if (c->pin[i]) (void)hipHostFree(c->pin[i]);
        if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
    }
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (auto e : c->ev_chunk)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    delete c;
}

extern "C" const char* ds_last_error(ds_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

extern "C" int ds_malloc(ds_ctx* c, void** dptr, size_t bytes) {
    if (!c || !dptr) return fail(c, DS_ERR_ARG, "ds_malloc: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(c, DS_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return DS_OK;
}
extern "C" int ds_free(ds_ctx* c, void* dptr) {
    if (!c) return fail(c, DS_ERR_ARG, "ds_free: null ctx");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(dptr));
    return DS_OK;
}
extern "C" int ds_host_alloc(ds_ctx* c, void** hptr, size_t bytes) {
    if (!c || !hptr) return fail(c, DS_ERR_ARG, "ds_host_alloc: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(c, DS_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return DS_OK;
}
extern "C" int ds_host_free(ds_ctx* c, void* hptr) {
    if (!c) return fail(c, DS_ERR_ARG, "ds_host_free: null ctx");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipHostFree(hptr));
    return DS_OK;
}
extern "C" int ds_upload(ds_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return fail(c, DS_ERR_ARG, "ds_upload: null argument");
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DS_OK;
}
extern "C" int ds_download(ds_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (bytes && (!dst || !src))) return fail(c, DS_ERR_ARG, "ds_download: null argument");
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DS_OK;
}
extern "C" int ds_memset(ds_ctx* c, void* dst, int value, size_t bytes) {
    if (!c) return fail(c, DS_ERR_ARG, "ds_memset: null ctx");
    HIPCHK(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return DS_OK;
}
extern "C" int ds_sync(ds_ctx* c) {
    if (!c) return fail(c, DS_ERR_ARG, "ds_sync: null ctx");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DS_OK;
}
extern "C" int ds_timer_start(ds_ctx* c) {
    if (!c) return fail(c, DS_ERR_ARG, "ds_timer_start: null ctx");
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return DS_OK;
}
extern "C" int ds_timer_stop(ds_ctx* c, float* ms) {
    if (!c || !ms) return fail(c, DS_ERR_ARG, "ds_timer_stop: null argument");
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return DS_OK;
}

extern "C" int ds_profile_enable(ds_ctx* c, int on) {
    if (!c) return fail(c, DS_ERR_ARG, "ds_profile_enable: null ctx");
    c->prof = on != 0;
    return DS_OK;
}

extern "C" int ds_profile_only(ds_ctx* c, const char* kernel_name) {
    if (!c) return fail(c, DS_ERR_ARG, "ds_profile_only: null ctx");
    c->prof_only = kernel_name ? kernel_name : "";
    return DS_OK;
}

extern "C" int ds_profile_stride(ds_ctx* c, int every) {
    if (!c || every < 1) return fail(c, DS_ERR_ARG, "ds_profile_stride: bad argument");
    c->prof_stride = every;
    c->prof_seen = 0;
    return DS_OK;
}

// What an event pair adds to the kernels it brackets: `n_kernels` empty kernels bracketed exactly as
// launch() does it, behind another kernel on the same stream (the start event then waits for a
// predecessor, as in a real step).  Average elapsed time over `reps` brackets, in ms.  With
// b1 = one and b2 = two kernels inside, b2 - b1 is what an empty kernel costs in the stream and
// 2 b1 - b2 a lower bound of the bracket's fixed cost.
__global__ void k_profile_nop() {}
static int prof_event(ds_ctx* c, hipEvent_t* ev);
extern "C" int ds_profile_overhead(ds_ctx* c, int reps, int n_kernels, double* ms) {
    if (!c || !ms || reps < 1 || n_kernels < 1) return fail(c, DS_ERR_ARG, "ds_profile_overhead: bad argument");
    hipEvent_t a, b;
    CHK(prof_event(c, &a));
    CHK(prof_event(c, &b));
    double sum = 0.0;
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(k_profile_nop, dim3(1), dim3(64), 0, c->stream);
        HIPCHK(c, hipEventRecord(a, c->stream));
        for (int k = 0; k < n_kernels; ++k) hipLaunchKernelGGL(k_profile_nop, dim3(1), dim3(64), 0, c->stream);
        HIPCHK(c, hipEventRecord(b, c->stream));
        HIPCHK(c, hipEventSynchronize(b));
        float e = 0.f;
        HIPCHK(c, hipEventElapsedTime(&e, a, b));
        sum += e;
    }
    c->prof_pool.push_back(a);
    c->prof_pool.push_back(b);
    *ms = sum / reps;
    return DS_OK;
}

// "name total_ms count\n" per kernel since the last call; synchronises the stream
extern "C" const char* ds_profile_report(ds_ctx* c) {
    if (!c) return "";
    (void)hipStreamSynchronize(c->stream);
    std::map<std::string, std::pair<double, long>> acc;
    for (auto& r : c->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& e = acc[r.name];
            e.first += ms;
            e.second += 1;
        }
        c->prof_pool.push_back(r.a);
        c->prof_pool.push_back(r.b);
    }
    c->prof_recs.clear();
    c->prof_text.clear();
    char line[256];
    for (auto& kv : acc) {
        snprintf(line, sizeof line, "%s %.6f %ld\n", kv.first.c_str(), kv.second.first, kv.second.second);
        c->prof_text += line;
    }
    return c->prof_text.c_str();
}

// launch names since the last call, space separated, each "group" or "group@variant" (which kernel
// family of a group ran: the tests of the kernel-selecting switches read it)
extern "C" const char* ds_routes(ds_ctx* c) {
    if (!c) return "";
    std::set<std::string> names;
    for (const char* n : c->routes) names.insert(n);
    c->routes.clear();
    c->routes_text.clear();
    for (auto& n : names) {
        if (!c->routes_text.empty()) c->routes_text += ' ';
        c->routes_text += n;
    }
    return c->routes_text.c_str();
}

// ---- internal helpers ------------------------------------------------------
static int get_twiddles(ds_ctx* c, int n, const float2** out);

static int reserve(ds_ctx* c, void** buf, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return DS_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (*buf) HIPCHK(c, hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(buf, want);
    if (e != hipSuccess) return fail(c, DS_ERR_NOMEM, "workspace hipMalloc failed");
    *cap = want;
    return DS_OK;
}

struct Carver {  // 256-byte aligned sub-allocations out of one buffer
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base((char*)b) {}
    template <typename T>
    T* take(size_t count) {
        T* p = (T*)(base + off);
        off += (count * sizeof(T) + 255) & ~size_t(255);
        return p;
    }
    static size_t pad(size_t bytes) { return (bytes + 255) & ~size_t(255); }
};

static int prof_event(ds_ctx* c, hipEvent_t* ev) {
    if (!c->prof_pool.empty()) {
        *ev = c->prof_pool.back();
        c->prof_pool.pop_back();
        return DS_OK;
    }
    HIPCHK(c, hipEventCreate(ev));
    return DS_OK;
}

template <typename K, typename A>
static int launch(ds_ctx* c, const char* name, K kernel, dim3 grid, int threads, size_t lds,
                  const A& args) {
    if (lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // "group@variant": the group is the name the profile reports (and ds_profile_only matches), the
    // whole string goes into the set ds_routes() hands out -- which kernel family really ran
    c->routes.insert(name);
    if (const char* at = strchr(name, '@')) {
        auto it = c->group_names.find(name);
        if (it == c->group_names.end()) it = c->group_names.emplace(name, std::string(name, at)).first;
        name = it->second.c_str();
    }
    ds_ctx::ProfRec rec{name, nullptr, nullptr};
    bool prof = c->prof && (c->prof_only.empty() || c->prof_only == name);
    if (prof && c->prof_stride > 1 && (c->prof_seen++ % c->prof_stride) != 0) prof = false;
    if (prof) {
        // The two events ride on the dispatch itself (hipExtLaunchKernel): they carry the kernel's own
        // begin / end timestamps -- what rocprofv3's kernel trace reports -- not the times of two marker
        // packets around it (hipEventRecord brackets held 4-6 us of dispatch and completion latency on top).
        CHK(prof_event(c, &rec.a));
        CHK(prof_event(c, &rec.b));
        hipExtLaunchKernelGGL(kernel, grid, dim3(threads), (std::uint32_t)lds, c->stream, rec.a, rec.b, 0u, args);
        HIPCHK(c, hipGetLastError());
        c->prof_recs.push_back(rec);
        return DS_OK;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, c->stream, args);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}

// every Welch route ends here: chunk partials -> spectra / transfer functions (kernels_finish.hpp).  Partial slabs of
// 4 GiB or more leave the kernel's 32-bit raw-buffer descriptors (size_guards.hpp); DSPTOOLBOX_AMD_FINISH_WIDE=1
// sends every call down that 64-bit-load path (the GPU test of it: such slabs themselves do not fit a test).
static int launch_finish(ds_ctx* c, dim3 grid, WelchFinArgs f) {
    f.force_wide = c->cfg.finish_wide ? 1 : 0;
    const bool wide = f.force_wide || welch_finish_wide_slab((int64_t)f.n_cx * (f.in_nb > 0 ? f.in_nb : f.fin.nb),
                                                             (int64_t)f.n_cy * (f.in_nb > 0 ? f.in_nb : f.fin.nb));
    return launch(c, wide ? "welch_finish@wide" : "welch_finish", k_welch_finish, grid, 256, 0, f);
}

#define DISPATCH_N(n, CALL)                                                        \
    switch (n) {                                                                   \
        case 8: { constexpr int NN = 8; CALL; } break;                             \
        case 16: { constexpr int NN = 16; CALL; } break;                           \
        case 32: { constexpr int NN = 32; CALL; } break;                           \
        case 64: { constexpr int NN = 64; CALL; } break;                           \
        case 128: { constexpr int NN = 128; CALL; } break;                         \
        case 256: { constexpr int NN = 256; CALL; } break;                         \
        case 512: { constexpr int NN = 512; CALL; } break;                         \
        case 1024: { constexpr int NN = 1024; CALL; } break;                       \
        case 2048: { constexpr int NN = 2048; CALL; } break;                       \
        case 4096: { constexpr int NN = 4096; CALL; } break;                       \
        case 8192: { constexpr int NN = 8192; CALL; } break;                       \
        case 16384: { constexpr int NN = 16384; CALL; } break;                     \
        default: return fail(c, DS_ERR_UNSUP, "FFT length must be a power of two in [8, 16384]"); \
    }

// per-length twiddle blob (fft_lds.hpp: one [k][t] table per pass, forward + reversed sequence)
static int get_twiddles(ds_ctx* c, int n, const float2** out) {
    auto it = c->tw.find(n);
    if (it != c->tw.end()) {
        *out = it->second;
        return DS_OK;
    }
    std::vector<float2> h;
    DISPATCH_N(n, {
        h.resize(std::max(1, tw_table_len<NN>()));
        fill_tw_table<NN>(h.data());
    });
    float2* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, sizeof(float2) * h.size()));
    HIPCHK(c, hipMemcpyAsync(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tw[n] = d;
    *out = d;
    return DS_OK;
}

static int check_fft_len(ds_ctx* c, int n, const char* what) {
    if (!is_pow2(n) || n < kMinFft)
        return fail(c, DS_ERR_ARG, std::string(what) + ": length must be a power of two >= 8");
    if (n > kMaxFft)
        return fail(c, DS_ERR_UNSUP, std::string(what) + ": lengths above 16384 are not built yet (LDS-resident FFT)");
    return DS_OK;
}

static int upload_table_fwd(ds_ctx* c, float2** slot, const std::vector<float2>& h);
static size_t stft_big_ws(int n_ch, int n_frames, int64_t nfft);
static int stft_big(ds_ctx* c, Carver& cv, const float* x, int n_ch, int64_t ld, int64_t n_samples,
                    int W, int hop, int64_t nfft, int64_t pad_front, int n_frames, const float* window,
                    int detrend, float scale, float edge_scale, int power, int layout, float2* out);
static int welch_big(ds_ctx* c, int kind, const float* x, int n_cx, int64_t ldx, const float* y,
                     int n_cy, int64_t ldy, int64_t n_samples, int W, int hop, int n_frames,
                     const float* window, int detrend, int average, int mode, int amp_sqrt,
                     double norm_scale, double factor, int halve_edges, float2* out_c, float* out_r);

extern "C" int ds_rfft_dev(ds_ctx* c, const float* x, int n_ch, int64_t ld, int64_t n_samples,
                           int n_fft, float scale, ds_c32* spec);

// any fft length (kernels_stft_any.hpp): rows = windowed frames -> ds_rfft_dev -> scaling pass
static int stft_any(ds_ctx* c, const float* x, int64_t n_samples, int n_ch, int64_t ld, int W, int hop,
                    int nfft, int64_t pad_front, int n_frames, const float* window, int detrend, float scale,
                    float edge_scale, int power, float2* out) {
    if (nfft < 2) return fail(c, DS_ERR_ARG, "ds_stft_r2c: fft length must be >= 2");
    const int keep = std::min(W, nfft), B = nfft / 2 + 1;
    const int64_t total_rows = (int64_t)n_frames * n_ch;
    // rows per group: the transform's scratch grows with the (padded) length; keep a group's
    // rows + spectra near 256 MB (Bluestein lengths count with their convolution length)
    int64_t conv = nfft;
    if (!is_pow2(nfft)) {
        conv = (int64_t)1 << 15;
        while (conv < 2 * (int64_t)nfft - 1) conv <<= 1;
    }
    int64_t group = std::max<int64_t>(2, ((int64_t)256 << 20) / (conv * 16));
    group = std::min<int64_t>(total_rows, group & ~(int64_t)1);
    if (group < 1) group = 1;
    CHK(reserve(c, &c->aux, &c->aux_bytes,
                Carver::pad(sizeof(float) * (size_t)group * keep) + Carver::pad(sizeof(float2) * (size_t)group * B)));
    Carver cv(c->aux);
    float* rows = cv.take<float>((size_t)group * keep);
    float2* tmp = cv.take<float2>((size_t)group * B);
    for (int64_t r0 = 0; r0 < total_rows; r0 += group) {
        const int nr = (int)std::min<int64_t>(group, total_rows - r0);
        stftany::PrepArgs pa{x, n_samples, ld, pad_front, n_ch, W, hop, detrend, window, keep, (int)r0, rows};
        CHK(launch(c, "stft_any_prepare", stftany::k_prepare, dim3(nr), 256, 0, pa));
        CHK(ds_rfft_dev(c, rows, nr, keep, keep, nfft, 1.0f, (ds_c32*)tmp));
        stftany::PostArgs po{tmp, out, B, nr, (int)r0, (int)total_rows, scale, edge_scale, (nfft % 2) == 0, power};
        const int64_t tot = (int64_t)B * nr;
        CHK(launch(c, "stft_any_post", stftany::k_post, dim3((unsigned)((tot + 255) / 256)), 256, 0, po));
    }
    return DS_OK;
}

// ---- STFT ------------------------------------------------------------------
extern "C" int ds_stft_r2c_dev(ds_ctx* c, const float* x, int64_t n_samples, int n_ch, int64_t ld,
                               int W, int hop, int nfft, int64_t pad_front, int n_frames,
                               const float* window, int detrend, float scale, float edge_scale,
                               int power, ds_c32* out) {
    if (!c || !x || !out || !window) return fail(c, DS_ERR_ARG, "ds_stft_r2c: null argument");
    if (n_ch <= 0 || n_samples <= 0 || W <= 0 || hop <= 0 || n_frames <= 0 || ld < n_samples)
        return fail(c, DS_ERR_ARG, "ds_stft_r2c: bad shape");
    if (!is_pow2(nfft) || nfft < kMinFft)  // numpy's rfft(n=...) takes any n: crop or pad
        return stft_any(c, x, n_samples, n_ch, ld, W, hop, nfft, pad_front, n_frames, window, detrend, scale,
                        edge_scale, power, (float2*)out);
    // 2^15 ... 2^18 points: one decimation-in-frequency pass, then the 4096-point register transform per class
    // (kernels_stft_long.hpp)
    // (k_sdif: frames of a group on grid.y, channel pairs on grid.z)
    if (const int R = stftl::classes_of(nfft); R && W <= nfft && (W == nfft || !detrend) && !c->cfg.stft_generic && (n_ch + 1) / 2 <= 65535) {
        int lgR = 0;
        while ((1 << lgR) < R) ++lgR;
        if (!c->w4_tables) {
            std::vector<float2> h;
            welch4096::host_tables(h);
            CHK(upload_table_fwd(c, &c->w4_tables, h));
        }
        float2** slot = &c->wl_tables[lgR - 1];
        if (!*slot) {
            std::vector<float2> h;
            welchl::host_tables(R, h);
            CHK(upload_table_fwd(c, slot, h));
        }
        const int n_pc = (n_ch + 1) / 2, n_groups = (n_ch + 15) / 16;
        const int per = std::min(65535, stftl::frames_per_group(n_ch, nfft, n_frames));
        CHK(reserve(c, &c->ws, &c->ws_bytes, Carver::pad(sizeof(float2) * (size_t)n_pc * per * nfft)));
        Carver cv(c->ws);
        float2* b = cv.take<float2>((size_t)n_pc * per * nfft);
        for (int f0 = 0; f0 < n_frames; f0 += per) {
            const int nf = std::min(per, n_frames - f0);
            // chunks of (frame, kind) units: two rounds of one workgroup (8 channels) per CU, as for 8192 / 16384 points
            const int n_units = nf * (R - 1);
            int n_chunks = std::max(1, std::min(n_units, 256 / std::max(1, std::min(256, n_groups))));
            if (c->cfg.stft4k_chunks > 0) n_chunks = std::min(n_units, c->cfg.stft4k_chunks);
            stftl::Args a{x, n_samples, ld, pad_front, n_ch, W, hop, n_frames, detrend, n_chunks, n_groups, R, lgR, f0, nf,
                          window, c->w4_tables, *slot, scale, edge_scale, b, (float2*)out};
            const dim3 gd(16, (unsigned)nf, (unsigned)n_pc);
            switch (R) {
                case 8: CHK(launch(c, "stft_long_dif", stftl::k_sdif<8>, gd, 256, 0, a)); break;
                case 16: CHK(launch(c, "stft_long_dif", stftl::k_sdif<16>, gd, 256, 0, a)); break;
                case 32: CHK(launch(c, "stft_long_dif", stftl::k_sdif<32>, gd, 256, 0, a)); break;
                default: CHK(launch(c, "stft_long_dif", stftl::k_sdif<64>, gd, 256, 0, a)); break;
            }
            const dim3 grid((unsigned)stft4k::grid_size(n_groups, n_chunks));
            CHK(power ? launch(c, "stft@long", stftl::k_stft_cls<true>, grid, stftl::NT, stftl::LDS_BYTES, a)
                      : launch(c, "stft@long", stftl::k_stft_cls<false>, grid, stftl::NT, stftl::LDS_BYTES, a));
        }
        return DS_OK;
    }
    if (nfft > kMaxFft && is_pow2(nfft)) {  // four-step transform per frame pair
        CHK(reserve(c, &c->ws, &c->ws_bytes, stft_big_ws(n_ch, n_frames, nfft)));
        Carver cv(c->ws);
        return stft_big(c, cv, x, n_ch, ld, n_samples, W, hop, nfft, pad_front, n_frames, window, detrend,
                        scale, edge_scale, power, 1, (float2*)out);
    }
    CHK(check_fft_len(c, nfft, "ds_stft_r2c nfft"));
    // 256-, 512- and 1024-point transforms (1024 = the reference's default frame): wave-level
    // register transforms, one frame pair per team of nfft/16 lanes (kernels_stft1024.hpp)
    const bool stft_generic = c->cfg.stft_generic;
    // frames of 128 / 64 / 32 samples: their transform is every 2nd / 4th / 8th bin of the 256-point transform of the
    // zero-padded frame (the wave kernel with decim; removing the frame mean still only clears bin 0 of the kept bins)
    const int decim = (nfft == 128 || nfft == 64 || nfft == 32) ? 256 / nfft : 1;
    const int nfft_k = decim > 1 ? 256 : nfft;  // the transform that runs
    if ((nfft_k == 2048 || nfft_k == 1024 || nfft_k == 512 || nfft_k == 256) && (W == nfft || (W < nfft && !detrend)) &&
        !stft_generic && stft1k::stft_wave_fits(n_samples, n_ch, ld, pad_front, nfft_k)) {
        const int nfft_api = nfft;
        (void)nfft_api;
        nfft = nfft_k;
        const int slot = nfft == 1024 ? 0 : (nfft == 512 ? 1 : (nfft == 256 ? 2 : 3));
        float2** tab = slot == 0 ? &c->stft1k_tables : &c->stft_wave_tables[slot - 1];
        if (!*tab) {
            std::vector<float2> h;
            if (nfft == 2048) stft1k::host_tables<2048>(h);
            else if (nfft == 1024) stft1k::host_tables<1024>(h);
            else if (nfft == 512) stft1k::host_tables<512>(h);
            else stft1k::host_tables<256>(h);
            CHK(upload_table_fwd(c, tab, h));
        }
        // channels per workgroup: 16 teams = 128-byte runs of the output X[bin][frame][channel] (whole
        // cache lines; at 1024 points that is one 1024-thread workgroup of 140 KB per CU instead of two
        // of 8 channels with 64-byte runs: transform of the 64-microphone shape 105 -> 97 us);
        // 2048 points: 8 x 17 KB images, one 1024-thread workgroup per CU (4 channels = 32-byte runs,
        // two per CU: 0.20 ms against 0.16)
        const int lanes = nfft / 16;
        int ct = std::min(nfft >= 2048 ? 8 : 16, n_ch);
        if (const int v = c->cfg.stft_ct; v >= 1 && v <= 16 && v * lanes <= 1024) ct = std::min(v, n_ch);
        while (ct & (ct - 1)) ct &= ct - 1;
        const size_t lds = nfft == 2048 ? stft1k::lds_bytes<2048>(ct) : nfft == 1024 ? stft1k::lds_bytes<1024>(ct)
                                        : (nfft == 512 ? stft1k::lds_bytes<512>(ct) : stft1k::lds_bytes<256>(ct));
        const int threads = lanes * ct;
        // frame pairs per workgroup: as few as keep the whole grid resident at once, at most 16
        const int per_cu = std::max(1, std::min<int>({(int)((160 * 1024) / lds), 2048 / std::max(64, threads), 8}));
        const int n_fp = (n_frames + 1) / 2, n_ct = (n_ch + ct - 1) / ct;
        const int64_t resident = 256 * (int64_t)per_cu;
        int fpw = std::max(1, std::min(16, (int)(((int64_t)n_fp * n_ct + resident - 1) / resident)));
        if (c->cfg.stft_fpw > 0) fpw = c->cfg.stft_fpw;
        StftArgs a{x, n_samples, ld, pad_front, n_ch, W, hop, n_frames, detrend, power, ct, fpw, window,
                   *tab, scale, edge_scale, (float2*)out, decim};
        dim3 grid((unsigned)((n_fp + fpw - 1) / fpw), (unsigned)n_ct);
        if (nfft == 2048)
            return power ? launch(c, "stft@wave", stft1k::k_stft_wave<2048, true>, grid, threads, lds, a)
                         : launch(c, "stft@wave", stft1k::k_stft_wave<2048, false>, grid, threads, lds, a);
        if (nfft == 1024)
            return power ? launch(c, "stft@wave", stft1k::k_stft_wave<1024, true>, grid, threads, lds, a)
                         : launch(c, "stft@wave", stft1k::k_stft_wave<1024, false>, grid, threads, lds, a);
        if (nfft == 512)
            return power ? launch(c, "stft@wave", stft1k::k_stft_wave<512, true>, grid, threads, lds, a)
                         : launch(c, "stft@wave", stft1k::k_stft_wave<512, false>, grid, threads, lds, a);
        return power ? launch(c, "stft@wave", stft1k::k_stft_wave<256, true>, grid, threads, lds, a)
                     : launch(c, "stft@wave", stft1k::k_stft_wave<256, false>, grid, threads, lds, a);
    }
    // 4096-point transforms: the register-resident transform of the Welch path, four teams of two neighbouring
    // channels per workgroup and frame (kernels_stft4096.hpp)
    if (nfft == 4096 && W <= nfft && (W == nfft || !detrend) && !stft_generic && stft4k::fits(n_samples, pad_front)) {
        if (!c->w4_tables) {
            std::vector<float2> h;
            welch4096::host_tables(h);
            CHK(upload_table_fwd(c, &c->w4_tables, h));
        }
        const int n_groups = (n_ch + 15) / 16;
        // chunks of frames: as many as put one workgroup (8 channels) on each of the 256 CUs
        int n_chunks = std::max(1, std::min(n_frames, 128 / std::max(1, std::min(128, n_groups))));
        if (c->cfg.stft4k_chunks > 0) n_chunks = std::min(n_frames, c->cfg.stft4k_chunks);
        stft4k::Args a{x, n_samples, ld, pad_front, n_ch, W, hop, n_frames, detrend, n_chunks, n_groups, window,
                       c->w4_tables, scale, edge_scale, (float2*)out, nullptr};
        const dim3 grid((unsigned)stft4k::grid_size(n_groups, n_chunks));
        return power ? launch(c, "stft@4k", stft4k::k_stft<true>, grid, stft4k::NT, stft4k::LDS_BYTES, a)
                     : launch(c, "stft@4k", stft4k::k_stft<false>, grid, stft4k::NT, stft4k::LDS_BYTES, a);
    }
    // 8192 / 16384 points: one radix-2 / radix-4 decimation-in-frequency stage on the windowed samples, then the
    // 4096-point kernel's structure per residue (kernels_stft4096.hpp, k_stft_dif)
    if ((nfft == 8192 || nfft == 16384) && W <= nfft && (W == nfft || !detrend) && !stft_generic &&
        stft4k::fits_long(n_samples, pad_front, nfft)) {
        if (!c->w4_tables) {
            std::vector<float2> h;
            welch4096::host_tables(h);
            CHK(upload_table_fwd(c, &c->w4_tables, h));
        }
        float2** twn = &c->stft_dif_tw[nfft == 8192 ? 0 : 1];
        if (!*twn) {
            std::vector<float2> h;
            stft4k::host_twiddles(nfft, h);
            CHK(upload_table_fwd(c, twn, h));
        }
        const int n_groups = (n_ch + 15) / 16;
        // chunks of (frame, phase) units: two rounds of one workgroup (8 channels) per CU (64 x 512 000 samples, 8192
        // points: 0.182 ms against 0.198 with one round)
        int n_chunks = std::max(1, std::min(n_frames, 256 / std::max(1, std::min(256, n_groups))));
        if (c->cfg.stft4k_chunks > 0) n_chunks = std::min(n_frames, c->cfg.stft4k_chunks);
        stft4k::Args a{x, n_samples, ld, pad_front, n_ch, W, hop, n_frames, detrend, n_chunks, n_groups, window,
                       c->w4_tables, scale, edge_scale, (float2*)out, *twn};
        const dim3 grid((unsigned)stft4k::grid_size(n_groups, n_chunks));
        if (nfft == 8192)
            return power ? launch(c, "stft@dif", stft4k::k_stft_dif<2, true>, grid, stft4k::NT, stft4k::Dif<2>::LDS_BYTES, a)
                         : launch(c, "stft@dif", stft4k::k_stft_dif<2, false>, grid, stft4k::NT, stft4k::Dif<2>::LDS_BYTES, a);
        return power ? launch(c, "stft@dif", stft4k::k_stft_dif<4, true>, grid, stft4k::NT, stft4k::Dif<4>::LDS_BYTES, a)
                     : launch(c, "stft@dif", stft4k::k_stft_dif<4, false>, grid, stft4k::NT, stft4k::Dif<4>::LDS_BYTES, a);
    }
    const float2* tw;
    CHK(get_twiddles(c, nfft, &tw));
    // channel tile: ct teams of NT threads (<= 1024 threads, <= 74 KB of LDS so two
    // workgroups share a CU; 8 channels = 64-byte runs of the (bins, frames, channels) output for
    // nfft 1024: 0.16 ms instead of 0.23 ms with 4 on the 64-mic CSM shape)
    int ct = 1;
    size_t lds = 0;
    int threads = 0;
    DISPATCH_N(nfft, {
        const size_t per = (size_t)stft_ch_stride<NN>() * sizeof(float2);
        ct = std::min<int>(stft_max_teams<NN>(), n_ch);
        // (override: fewer teams only, the kernel is compiled for the maximum)
        if (const int v = c->cfg.stft_ct; v >= 1 && v <= stft_max_teams<NN>()) ct = std::min(v, std::max(1, n_ch));
        while (ct & (ct - 1)) ct &= ct - 1;  // power of two (shift-only index math in the kernel)
        lds = per * ct;
        threads = ct * Cfg<NN>::NT;
    });
    // frame pairs per workgroup: as few as keep the whole grid resident at once (two workgroups
    // on each of the 256 CUs: no second, partly filled round), at most 16
    const int n_fp = (n_frames + 1) / 2, n_ct = (n_ch + ct - 1) / ct;
    int fpw = std::max(1, std::min(16, (int)(((int64_t)n_fp * n_ct + 511) / 512)));
    if (c->cfg.stft_fpw > 0) fpw = c->cfg.stft_fpw;
    StftArgs a{x, n_samples, ld, pad_front, n_ch, W, hop, n_frames, detrend, power, ct, fpw, window, tw,
               scale, edge_scale, (float2*)out};
    dim3 grid((unsigned)((n_fp + fpw - 1) / fpw), (unsigned)n_ct);
    DISPATCH_N(nfft, CHK(launch(c, "stft@generic", k_stft<NN>, grid, threads, lds, a)));
    return DS_OK;
}

// ---- inverse STFT ----------------------------------------------------------
// ---- inverse STFT with an FFT length that is not a power of two (transforms/transforms.py:548-577 calls
// np.fft.irfft(stft, n=fft_length_samples) with any n) ------------------------------------------------------
// Every (channel, frame) spectrum is one "channel" of the spectral-division machinery, which already
// inverts any length (Bluestein on the four-step transform): irfft_n(1 * R) with a unit impulse as the
// numerator.  k_istft_spec lays the spectra out per (channel, frame) -- cropped or zero-padded to
// n / 2 + 1 bins as numpy does --, the division writes the frames [c][f][W], k_istft_scale applies the
// synthesis window and the scale; the overlap-add kernel is the same as for powers of two.
extern "C" int ds_deconv_dev(ds_ctx* c, const float* y, int n_items, int n_ch, int64_t ld, int64_t n_samples, int n_fft,
                             const ds_c32* r, int r_per_channel, int64_t n_out, int64_t ld_out, float* ir);
static int64_t blue_len(int64_t L);
static int check_blue_len(ds_ctx* c, int64_t L, const char* what);
__global__ void k_istft_spec(const float2* stft, int n_bins, int n_frames, int n_ch, int f0, int nf, int nb, float2* r,
                             float* ones) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (int64_t)n_ch * nf * nb;
    if (i < (int64_t)n_ch * nf) ones[i] = 1.f;
    if (i >= total) return;
    const int k = (int)(i % nb);
    const int64_t cf = i / nb;
    const int f = (int)(cf % nf), ch = (int)(cf / nf);
    r[i] = k < n_bins ? stft[((int64_t)k * n_frames + f0 + f) * n_ch + ch] : make_float2(0.f, 0.f);
}
__global__ void k_istft_scale(float* frames, int64_t total, int W, const float* window, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) frames[i] *= scale * window[i % W];
}
static int istft_any_frames(ds_ctx* c, const float2* stft, int n_bins, int n_frames, int n_ch, int nfft, int W,
                            const float* window, float scale, float* frames /* [c][f][W] */) {
    CHK(check_blue_len(c, nfft, "ds_istft nfft"));
    const int nb = nfft / 2 + 1;
    // groups of frames: the division's scratch is ~4 x 8 bytes x (frames x channels / 2) x transform length
    const int64_t m_len = blue_len(nfft);
    int group = (int)std::max<int64_t>(1, ((int64_t)64 << 20) / std::max<int64_t>(1, m_len * n_ch));
    group = std::min(group, n_frames);
    CHK(reserve(c, &c->aux, &c->aux_bytes,
                Carver::pad(sizeof(float2) * (size_t)n_ch * group * nb) + Carver::pad(sizeof(float) * (size_t)n_ch * group) +
                    Carver::pad(sizeof(float) * (size_t)n_ch * group * W) + 4096));
    Carver cv(c->aux);
    float2* r = cv.take<float2>((size_t)n_ch * group * nb);
    float* ones = cv.take<float>((size_t)n_ch * group);
    float* part = cv.take<float>((size_t)n_ch * group * W);
    for (int f0 = 0; f0 < n_frames; f0 += group) {
        const int nf = std::min(group, n_frames - f0);
        const int64_t total = (int64_t)n_ch * nf * nb;
        hipLaunchKernelGGL(k_istft_spec, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, stft, n_bins, n_frames,
                           n_ch, f0, nf, nb, r, ones);
        HIPCHK(c, hipGetLastError());
        CHK(ds_deconv_dev(c, ones, 1, n_ch * nf, 1, 1, nfft, (const ds_c32*)r, 1, W, W, part));
        // part: [(ch nf + f) W + m] -> frames [(ch n_frames + f0 + f) W + m], windowed and scaled
        const int64_t tw = (int64_t)n_ch * nf * W;
        // (`scale` is defined against an UNnormalised inverse transform, as k_istft computes it; the division
        // machinery returns numpy's normalised irfft)
        hipLaunchKernelGGL(k_istft_scale, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, c->stream, part, tw, W, window,
                           scale * (float)nfft);
        HIPCHK(c, hipGetLastError());
        for (int ch = 0; ch < n_ch; ++ch)
            HIPCHK(c, hipMemcpyAsync(frames + ((int64_t)ch * n_frames + f0) * W, part + (int64_t)ch * nf * W,
                                     sizeof(float) * (size_t)nf * W, hipMemcpyDeviceToDevice, c->stream));
    }
    return DS_OK;
}

extern "C" int ds_istft_dev(ds_ctx* c, const ds_c32* stft, int n_bins, int n_frames, int n_ch, int nfft,
                            int W, int step, int frame_offset, int n_frames_total, const float* window,
                            float scale, int64_t total_length, float* out, int64_t ld_out) {
    if (!c || !stft || !window || !out) return fail(c, DS_ERR_ARG, "ds_istft: null argument");
    if (n_bins <= 0 || n_frames <= 0 || n_ch <= 0 || W <= 0 || step <= 0 || step > W || frame_offset < 0 ||
        n_frames_total < n_frames + frame_offset || total_length <= 0 || ld_out < total_length)
        return fail(c, DS_ERR_ARG, "ds_istft: bad shape");
    if (W > nfft) return fail(c, DS_ERR_ARG, "ds_istft: window longer than the FFT length");
    // 8192 ... 262144 points: class transforms on the 4096-point register kernel, then the radix-R stage with the
    // overlap-add fused where frames overlap by half (kernels_istft_long.hpp)
    if (const int R = istftl::classes_of(nfft); R && c->cfg.istft_wave && c->cfg.istft_fused && n_frames <= 65535 &&
                                                 (n_ch + 1) / 2 <= 65535 &&
                                                 (int64_t)n_bins * n_frames * n_ch * 8 < ((int64_t)1 << 32) - 16) {
        int lgR = 0;
        while ((1 << lgR) < R) ++lgR;
        if (!c->w4_tables) {
            std::vector<float2> h;
            welch4096::host_tables(h);
            CHK(upload_table_fwd(c, &c->w4_tables, h));
        }
        float2** slot = &c->wl_tables[lgR - 1];
        if (!*slot) {
            std::vector<float2> h;
            welchl::host_tables(R, h);
            CHK(upload_table_fwd(c, slot, h));
        }
        const int n_pc = (n_ch + 1) / 2, n_groups = (n_ch + 15) / 16;
        // (fused: the class sequences of ALL frames at once, addressed through a raw-buffer descriptor: below 4 GB)
        const bool fused = W == nfft && 2 * step == nfft && R <= 16 &&
                           (int64_t)n_pc * n_frames * nfft * 8 < ((int64_t)1 << 32) - 16;
        const int per = fused ? n_frames : istftl::frames_per_group(n_ch, nfft, n_frames);
        const size_t cq_bytes = Carver::pad(sizeof(float2) * (size_t)n_pc * per * nfft);
        CHK(reserve(c, &c->ws, &c->ws_bytes, cq_bytes + (fused ? 0 : Carver::pad(sizeof(float) * (size_t)n_ch * n_frames * W))));
        Carver cv(c->ws);
        float2* cq = cv.take<float2>((size_t)n_pc * per * nfft);
        float* frames = fused ? nullptr : cv.take<float>((size_t)n_ch * n_frames * W);
        istftl::Args a{(const float2*)stft, n_bins, n_frames, n_ch, W, step, 1, n_groups, R, lgR, 0, n_frames, window,
                       c->w4_tables, *slot, scale, cq, frames, frame_offset, n_frames_total, total_length, ld_out, out};
        for (int f0 = 0; f0 < n_frames; f0 += per) {
            a.f0 = f0;
            a.nf = std::min(per, n_frames - f0);
            const int n_units = a.nf * (R - 1);
            a.n_chunks = std::max(1, std::min(n_units, 256 / std::max(1, std::min(256, n_groups))));
            const dim3 gc((unsigned)stft4k::grid_size(n_groups, a.n_chunks));
            CHK((n_ch & 1) ? launch(c, "istft_long_cls", istftl::k_icls<false>, gc, istftl::NT, istftl::LDS_BYTES, a)
                           : launch(c, "istft_long_cls", istftl::k_icls<true>, gc, istftl::NT, istftl::LDS_BYTES, a));
            if (fused) break;
            const dim3 gd(16, (unsigned)a.nf, (unsigned)n_pc);
            switch (R) {
                case 2: CHK(launch(c, "istft@long", istftl::k_isdif_frames<2>, gd, 256, istftl::xch_bytes(2), a)); break;
                case 4: CHK(launch(c, "istft@long", istftl::k_isdif_frames<4>, gd, 256, istftl::xch_bytes(4), a)); break;
                case 8: CHK(launch(c, "istft@long", istftl::k_isdif_frames<8>, gd, 256, istftl::xch_bytes(8), a)); break;
                case 16: CHK(launch(c, "istft@long", istftl::k_isdif_frames<16>, gd, 256, istftl::xch_bytes(16), a)); break;
                case 32: CHK(launch(c, "istft@long", istftl::k_isdif_frames<32>, gd, 256, istftl::xch_bytes(32), a)); break;
                default: CHK(launch(c, "istft@long", istftl::k_isdif_frames<64>, gd, 256, istftl::xch_bytes(64), a)); break;
            }
        }
        if (fused) {
            // chunks of frames (+ 1 frame each for the carry): 8192 workgroups of 256 threads where the frames allow
            // (a thread has only R loads in flight), at least 4 frames per chunk
            const int want = std::max(1, 8192 / std::max(1, 16 * n_pc));
            a.n_chunks = std::max(1, std::min((n_frames + 3) / 4, want));
            if (c->cfg.istft_fpw > 0) a.n_chunks = std::min(n_frames, c->cfg.istft_fpw);
            const dim3 gd(16, (unsigned)a.n_chunks, (unsigned)n_pc);
            switch (R) {
                case 2: return launch(c, "istft@long_ola", istftl::k_isdif_ola<2>, gd, 256, istftl::xch_bytes(2), a);
                case 4: return launch(c, "istft@long_ola", istftl::k_isdif_ola<4>, gd, 256, istftl::xch_bytes(4), a);
                case 8: return launch(c, "istft@long_ola", istftl::k_isdif_ola<8>, gd, 256, istftl::xch_bytes(8), a);
                default: return launch(c, "istft@long_ola", istftl::k_isdif_ola<16>, gd, 256, istftl::xch_bytes(16), a);
            }
        }
        IstftOlaArgs o{frames, n_frames, n_ch, W, step, frame_offset, n_frames_total, window, total_length, ld_out, out};
        if (W % 4 == 0 && step % 4 == 0 && total_length % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
            ((uintptr_t)window & 15) == 0)
            return launch(c, "istft_ola", k_istft_ola4, dim3((unsigned)((total_length / 4 + 255) / 256), n_ch), 256, 0, o);
        return launch(c, "istft_ola", k_istft_ola, dim3((unsigned)((total_length + 255) / 256), n_ch), 256, 0, o);
    }
    if (!is_pow2(nfft) || nfft < kMinFft || nfft > kMaxFft) {
        // (ws, io and aux are all taken -- the division's scratch, the host entry point's staging, the
        // per-group spectra -- so the frames live in a fourth context-owned reserve: grown on demand like
        // the others, no allocation, synchronisation or free per call; this route is correct, not tuned)
        CHK(reserve(c, &c->frames, &c->frames_bytes, sizeof(float) * (size_t)n_ch * n_frames * W));
        float* frames = (float*)c->frames;
        CHK(istft_any_frames(c, (const float2*)stft, n_bins, n_frames, n_ch, nfft, W, window, scale, frames));
        IstftOlaArgs o{frames, n_frames, n_ch, W, step, frame_offset, n_frames_total, window, total_length, ld_out, out};
        return launch(c, "istft_ola", k_istft_ola, dim3((unsigned)((total_length + 255) / 256), n_ch), 256, 0, o);
    }
    CHK(check_fft_len(c, nfft, "ds_istft nfft"));
    const float2* tw;
    CHK(get_twiddles(c, nfft, &tw));
    // 50 % overlap of full-length frames: transform and overlap-add in one kernel, no frames in memory
    const bool no_fuse = !c->cfg.istft_fused;
    // ... on the wave-level transform for 256 ... 2048 points (kernels_stft1024.hpp, k_istft_wave)
    const bool no_wave = !c->cfg.istft_wave;
    if (W == nfft && 2 * step == nfft && n_ch > 1 && !no_fuse && !no_wave &&
        (nfft == 2048 || nfft == 1024 || nfft == 512 || nfft == 256) &&
        (int64_t)n_bins * n_frames * n_ch * 8 < ((int64_t)1 << 32) - 16) {
        // (2048 points: 45 registers over the 128 of a 1024-thread workgroup: four teams = 512 threads there)
        const int slot = nfft == 1024 ? 0 : (nfft == 512 ? 1 : (nfft == 256 ? 2 : 3));
        float2** tab = slot == 0 ? &c->stft1k_tables : &c->stft_wave_tables[slot - 1];
        if (!*tab) {
            std::vector<float2> h;
            if (nfft == 2048) stft1k::host_tables<2048>(h);
            else if (nfft == 1024) stft1k::host_tables<1024>(h);
            else if (nfft == 512) stft1k::host_tables<512>(h);
            else stft1k::host_tables<256>(h);
            CHK(upload_table_fwd(c, tab, h));
        }
        const int lanes = nfft / 16;
        int ct = std::min(nfft == 2048 ? 4 : 16, n_ch);
        while (ct & (ct - 1)) ct &= ct - 1;
        if (ct > 1) {
            const size_t lds = nfft == 2048 ? stft1k::istft_lds_bytes<2048>(ct) : nfft == 1024 ? stft1k::istft_lds_bytes<1024>(ct)
                                            : (nfft == 512 ? stft1k::istft_lds_bytes<512>(ct) : stft1k::istft_lds_bytes<256>(ct));
            const int threads = lanes * ct;
            const int n_fp = (n_frames + 1) / 2, n_ct = (n_ch + ct - 1) / ct;
            const int per_cu = std::max(1, std::min<int>((int)((160 * 1024) / lds), 2048 / std::max(64, threads)));
            // frame pairs per workgroup (+ 1 for the carry): two rounds of resident workgroups, at least 4 (1024 points: 8)
            // -- 64 x 512 000 samples: 0.154 / 0.130 / 0.135 ms at 256 / 512 / 1024 points, 0.18 / 0.13 / 0.145 one step off
            int fpw = std::max(nfft >= 1024 ? 8 : 4, std::min(64, (int)(((int64_t)n_fp * n_ct + 512 * per_cu - 1) / (512 * per_cu))));
            if (c->cfg.istft_fpw > 0) fpw = c->cfg.istft_fpw;
            IstftFusedArgs fa{IstftArgs{(const float2*)stft, n_bins, n_frames, n_ch, W, window, *tab, scale, nullptr, ct, fpw},
                              frame_offset, n_frames_total, total_length, ld_out, out};
            const dim3 grid((unsigned)((n_fp + fpw - 1) / fpw), (unsigned)n_ct);
            if (nfft == 2048) return launch(c, "istft@wave", stft1k::k_istft_wave<2048>, grid, threads, lds, fa);
            if (nfft == 1024) return launch(c, "istft@wave", stft1k::k_istft_wave<1024>, grid, threads, lds, fa);
            if (nfft == 512) return launch(c, "istft@wave", stft1k::k_istft_wave<512>, grid, threads, lds, fa);
            return launch(c, "istft@wave", stft1k::k_istft_wave<256>, grid, threads, lds, fa);
        }
    }
    // ... and on the 4096-point register transform, two neighbouring channels per team (kernels_stft4096.hpp, k_istft)
    if (W == nfft && nfft == 4096 && 2 * step == nfft && n_ch > 1 && !no_fuse && !no_wave &&
        total_length < ((int64_t)1 << 31) && (int64_t)n_bins * n_frames * n_ch * 8 < ((int64_t)1 << 32) - 16) {
        if (!c->w4_tables) {
            std::vector<float2> h;
            welch4096::host_tables(h);
            CHK(upload_table_fwd(c, &c->w4_tables, h));
        }
        const int n_groups = (n_ch + 15) / 16;
        // chunks of frames (+ 1 frame each for the carry): two rounds of one workgroup (8 channels) per CU
        int n_chunks = std::max(1, std::min((n_frames + 3) / 4, 256 / std::max(1, std::min(256, n_groups))));
        if (c->cfg.istft_fpw > 0) n_chunks = std::min(n_frames, c->cfg.istft_fpw);
        IstftFusedArgs fa{IstftArgs{(const float2*)stft, n_bins, n_frames, n_ch, W, window, c->w4_tables, scale, nullptr, 1, n_chunks},
                          frame_offset, n_frames_total, total_length, ld_out, out};
        const dim3 g4((unsigned)stft4k::grid_size(n_groups, n_chunks));
        return (n_ch & 1) ? launch(c, "istft@4k", stft4k::k_istft<false>, g4, stft4k::NT, stft4k::ISTFT_LDS_BYTES, fa)
                          : launch(c, "istft@4k", stft4k::k_istft<true>, g4, stft4k::NT, stft4k::ISTFT_LDS_BYTES, fa);
    }
    if (W == nfft && 2 * step == nfft && n_ch > 1 && !no_fuse) {
        int ct = 1;
        size_t lds = 0;
        int threads = 0, nt = 64;
        DISPATCH_N(nfft, {
            ct = std::min<int>(stft_max_teams<NN>(), n_ch);
            while (ct & (ct - 1)) ct &= ct - 1;
            lds = (size_t)stft_ch_stride<NN>() * sizeof(float2) * ct + sizeof(float) * (size_t)(NN / 2);  // + 1 / envelope
            threads = ct * Cfg<NN>::NT;
            nt = Cfg<NN>::NT;
        });
        if (ct > 1 && nfft % (2 * nt) == 0) {
            const int n_fp = (n_frames + 1) / 2, n_ct = (n_ch + ct - 1) / ct;
            // frame pairs per workgroup: every workgroup transforms one more pair (the carry in front of its range)
            // (64 x 512 000 samples, windows of 256 / 1024 / 4096: ~250 workgroups measured best: 0.21 / 0.21 / 0.29 ms)
            int fpw = std::max(4, std::min(64, (int)(((int64_t)n_fp * n_ct + 255) / 256)));
            if (c->cfg.istft_fpw > 0) fpw = c->cfg.istft_fpw;
            IstftFusedArgs fa{IstftArgs{(const float2*)stft, n_bins, n_frames, n_ch, W, window, tw, scale, nullptr, ct, fpw},
                              frame_offset, n_frames_total, total_length, ld_out, out};
            DISPATCH_N(nfft, {
                if constexpr (stft_max_teams<NN>() > 1 && NN % (2 * Cfg<NN>::NT) == 0)  // (ct > 1 never holds otherwise)
                    CHK(launch(c, "istft@fused", k_istft_fused<NN>, dim3((unsigned)((n_fp + fpw - 1) / fpw), (unsigned)n_ct),
                               threads, lds, fa));
            });
            return DS_OK;
        }
    }
    CHK(reserve(c, &c->ws, &c->ws_bytes, sizeof(float) * (size_t)n_ch * n_frames * W));
    float* frames = (float*)c->ws;
    IstftArgs a{(const float2*)stft, n_bins, n_frames, n_ch, W, window, tw, scale, frames};
    // ct neighbouring channels per workgroup (runs of 8 ct bytes of the channel-fastest spectrogram) wherever more
    // than one image fits; DSPTOOLBOX_AMD_ISTFT_CT=1 keeps one channel per workgroup
    const bool one_ch = c->cfg.istft_one_ch;
    int ct = 1;
    size_t lds = 0;
    int threads = 0;
    DISPATCH_N(nfft, {
        ct = std::min<int>(stft_max_teams<NN>(), n_ch);
        while (ct & (ct - 1)) ct &= ct - 1;
        lds = (size_t)stft_ch_stride<NN>() * sizeof(float2) * ct;
        threads = ct * Cfg<NN>::NT;
    });
    if (ct > 1 && !one_ch) {
        const int n_fp = (n_frames + 1) / 2, n_ct = (n_ch + ct - 1) / ct;
        a.ct = ct;
        a.fpw = std::max(1, std::min(16, (int)(((int64_t)n_fp * n_ct + 511) / 512)));
        DISPATCH_N(nfft, {
            if constexpr (stft_max_teams<NN>() > 1)
                CHK(launch(c, "istft@ct", k_istft_ct<NN>, dim3((unsigned)((n_fp + a.fpw - 1) / a.fpw), (unsigned)n_ct), threads,
                           lds, a));
        });
    } else {
        DISPATCH_N(nfft, CHK(launch(c, "istft@generic", k_istft<NN>, dim3((n_frames + 1) / 2, n_ch), Cfg<NN>::NT,
                                    Cfg<NN>::LDS_BYTES, a)));
    }
    IstftOlaArgs o{frames, n_frames, n_ch, W, step, frame_offset, n_frames_total, window, total_length, ld_out, out};
    // four samples per thread where every row and frame boundary is a multiple of four samples (and 16-byte aligned)
    if (W % 4 == 0 && step % 4 == 0 && total_length % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
        ((uintptr_t)window & 15) == 0)
        CHK(launch(c, "istft_ola", k_istft_ola4, dim3((unsigned)((total_length / 4 + 255) / 256), n_ch), 256, 0, o));
    else
        CHK(launch(c, "istft_ola", k_istft_ola, dim3((unsigned)((total_length + 255) / 256), n_ch), 256, 0, o));
    return DS_OK;
}

// ---- band powers of a spectrogram (mel spectrogram / MFCC) ----------------------------------
extern "C" int ds_band_power_dev(ds_ctx* c, const ds_c32* stft, int n_bins, int64_t n_fc, const float* weights,
                                 const int* band_start, const int* band_stop, int n_bands, int to_db,
                                 int dct_abs, float* out) {
    if (!c || !stft || !weights || !band_start || !band_stop || !out)
        return fail(c, DS_ERR_ARG, "ds_band_power: null argument");
    if (n_bins <= 0 || n_fc <= 0 || n_bands <= 0 || n_bands > 65535)
        return fail(c, DS_ERR_ARG, "ds_band_power: bad shape");
    float* dst = out;
    if (dct_abs) {
        CHK(reserve(c, &c->ws, &c->ws_bytes, sizeof(float) * (size_t)n_bands * n_fc));
        dst = (float*)c->ws;
    }
    BandPowerArgs a{(const float2*)stft, weights, band_start, band_stop, n_bins, n_bands, n_fc, to_db, dst};
    const unsigned gx = (unsigned)((n_fc + 255) / 256);
    CHK(launch(c, "band_power", k_band_power, dim3(gx, n_bands), 256, 0, a));
    if (dct_abs) {
        DctArgs d{dst, n_bands, n_fc, out};
        CHK(launch(c, "dct2_abs", k_dct2_abs, dim3(gx, n_bands), 256, 0, d));
    }
    return DS_OK;
}

extern "C" int ds_band_power(ds_ctx* c, const ds_c32* stft, int n_bins, int64_t n_fc, const float* weights,
                             const int* band_start, const int* band_stop, int n_bands, int to_db, int dct_abs,
                             float* out) {
    if (!c || !stft || !weights || !band_start || !band_stop || !out)
        return fail(c, DS_ERR_ARG, "ds_band_power: null argument");
    if (n_bins <= 0 || n_fc <= 0 || n_bands <= 0) return fail(c, DS_ERR_ARG, "ds_band_power: bad shape");
    const size_t ns = (size_t)n_bins * n_fc, nw = (size_t)n_bands * n_bins, no = (size_t)n_bands * n_fc;
    CHK(reserve(c, &c->io, &c->io_bytes,
                Carver::pad(ns * 8) + Carver::pad(nw * 4) + 2 * Carver::pad((size_t)n_bands * 4) + Carver::pad(no * 4) + 4096));
    Carver cv(c->io);
    float2* dx = cv.take<float2>(ns);
    float* dw = cv.take<float>(nw);
    int* d0 = cv.take<int>(n_bands);
    int* d1 = cv.take<int>(n_bands);
    float* dout = cv.take<float>(no);
    CHK(ds_upload(c, dx, stft, ns * 8));
    CHK(ds_upload(c, dw, weights, nw * 4));
    CHK(ds_upload(c, d0, band_start, (size_t)n_bands * 4));
    CHK(ds_upload(c, d1, band_stop, (size_t)n_bands * 4));
    CHK(ds_band_power_dev(c, (const ds_c32*)dx, n_bins, n_fc, dw, d0, d1, n_bands, to_db, dct_abs, dout));
    return ds_download(c, out, dout, no * 4);
}

// ---- Welch -----------------------------------------------------------------
struct WelchPlan {
    int n_chunks, fpc;
};
static WelchPlan plan_welch(int n_frames, int units) {
    // >= ~1024 workgroups when there is enough work, <= 32 frames per fp32 chain
    int by_len = (n_frames + 31) / 32;
    int by_fill = (1024 + units - 1) / units;
    int n_chunks = std::max(by_len, std::min(by_fill, (n_frames + 1) / 2));
    n_chunks = std::max(1, n_chunks);
    int fpc = (n_frames + n_chunks - 1) / n_chunks;
    fpc = (fpc + 1) & ~1;  // even: frames travel in pairs in the input-spectra kernel
    n_chunks = (n_frames + fpc - 1) / fpc;
    return {n_chunks, fpc};
}

// kind 0: tf+coh, 1: psd of x, 2: csd of (x[c], y[c])
// Median kernels keep `series` float series of n_frames values (padded to a power of two) per bin in LDS: the number of bins
// per workgroup (8, 4, 2 or 1) that fits 150 KB; 0 if not even one does.
static int median_bins_per_block(int series, int n_frames, size_t* lds) {
    for (int bpb = 8; bpb >= 1; bpb >>= 1) {
        const size_t need = ((size_t)bpb * series * median_stride(n_frames) + (size_t)bpb * series * 2) * sizeof(float);
        if (need <= 150 * 1024) {
            *lds = need;
            return bpb;
        }
    }
    return 0;
}

static int welch_common(ds_ctx* c, int kind, const float* x, int n_cx, int64_t ldx, const float* y,
                        int n_cy, int64_t ldy, int64_t n_samples, int W, int hop, int n_frames,
                        const float* window, int detrend, int average, int mode, int amp_sqrt,
                        double norm_scale, double factor, int halve_edges, float2* out_c,
                        float* out_r) {
    if (!c || !x || !window) return fail(c, DS_ERR_ARG, "welch: null argument");
    if (average != DS_AVG_MEAN && average != DS_AVG_MEDIAN)
        return fail(c, DS_ERR_ARG, "welch: average must be mean (0) or median (1)");
    if (kind != 1 && !y) return fail(c, DS_ERR_ARG, "welch: null output-signal pointer");
    if (n_cx <= 0 || n_samples <= 0 || hop <= 0 || hop > W || n_frames <= 0 || ldx < n_samples)
        return fail(c, DS_ERR_ARG, "welch: bad shape");
    if (kind != 1 && (n_cy <= 0 || ldy < n_samples || !(n_cx == 1 || n_cx == n_cy)))
        return fail(c, DS_ERR_ARG, "welch: input must have 1 channel or as many as the output");
    if (kind == 0 && (mode < DS_TF_H1 || mode > DS_TF_H3))
        return fail(c, DS_ERR_ARG, "welch: unsupported transfer function type");
    if (W > kMaxFft && is_pow2(W))
        return welch_big(c, kind, x, n_cx, ldx, y, n_cy, ldy, n_samples, W, hop, n_frames, window, detrend,
                         average, mode, amp_sqrt, norm_scale, factor, halve_edges, out_c, out_r);
    CHK(check_fft_len(c, W, "welch window length"));
    const float2* tw;
    CHK(get_twiddles(c, W, &tw));
    const int nb = W / 2 + 1;
    const int units = kind == 1 ? n_cx : (n_cy + 1) / 2;
    WelchPlan pl = plan_welch(n_frames, units);
    if (average == DS_AVG_MEDIAN) {
        // frame spectra of x (and y) -> per-bin medians -> the usual finish with bias n
        // (n = F or F-1, odd; the reference's `csd /= sum((-1)**(n+1)/n)` multiplies by n)
        const int nyc = kind == 1 ? 0 : n_cy;
        size_t lds = 0;
        const int bpb = median_bins_per_block(3, n_frames, &lds);
        if (!bpb)
            return fail(c, DS_ERR_UNSUP, "welch: median averaging over more than 12 799 frames is not built yet");
        size_t mb = Carver::pad(sizeof(float2) * (size_t)n_cx * n_frames * nb) +
                    Carver::pad(sizeof(float2) * (size_t)nyc * n_frames * nb) +
                    Carver::pad(sizeof(float) * (size_t)pl.n_chunks * std::max(n_cx, nyc) * nb) +
                    Carver::pad(sizeof(float) * (size_t)n_cx * nb) + Carver::pad(sizeof(float2) * (size_t)std::max(1, nyc) * nb) +
                    Carver::pad(sizeof(float) * (size_t)std::max(1, nyc) * nb);
        CHK(reserve(c, &c->ws, &c->ws_bytes, mb));
        Carver cv(c->ws);
        float2* xsp = cv.take<float2>((size_t)n_cx * n_frames * nb);
        float2* ysp = nyc ? cv.take<float2>((size_t)nyc * n_frames * nb) : nullptr;
        float* scratch = cv.take<float>((size_t)pl.n_chunks * std::max(n_cx, nyc) * nb);
        float* mxx = cv.take<float>((size_t)n_cx * nb);
        float2* mxy = cv.take<float2>((size_t)std::max(1, nyc) * nb);
        float* myy = cv.take<float>((size_t)std::max(1, nyc) * nb);
        {
            XspecArgs ax{x, n_samples, ldx, n_cx, W, hop, n_frames, detrend, pl.fpc, window, tw, xsp, scratch};
            DISPATCH_N(W, CHK(launch(c, "welch_xspec", k_xspec<NN>, dim3(pl.n_chunks, n_cx), Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, ax)));
        }
        if (nyc) {
            XspecArgs ay{y, n_samples, ldy, nyc, W, hop, n_frames, detrend, pl.fpc, window, tw, ysp, scratch};
            DISPATCH_N(W, CHK(launch(c, "welch_xspec", k_xspec<NN>, dim3(pl.n_chunks, nyc), Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, ay)));
        }
        MedianArgs m{xsp, ysp, n_cx, nyc, n_frames, nb, kind, bpb, mxx, mxy, myy};
        CHK(launch(c, "welch_median", k_welch_median, dim3((nb + bpb - 1) / bpb, kind == 1 ? n_cx : n_cy), 256, lds, m));
        const int nbias = (n_frames & 1) ? n_frames : n_frames - 1;
        WelchFinArgs f{mxx, mxy, myy, 1, 1, n_cx, n_cy, kind, mode,
                       FinishPar{norm_scale * (double)std::max(1, nbias), factor, halve_edges, amp_sqrt, nb},
                       out_c, out_r};
        int64_t total = (int64_t)nb * (kind == 1 ? n_cx : n_cy);
        CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
        return DS_OK;
    }
    const bool need_xs = kind != 1;
    size_t bytes = Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * nb);
    if (need_xs) {
        bytes += Carver::pad(sizeof(float2) * (size_t)n_cx * n_frames * nb);
        bytes += Carver::pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * nb);
        bytes += Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_cy * nb);
    }
    CHK(reserve(c, &c->ws, &c->ws_bytes, bytes));
    Carver cv(c->ws);
    float* pxx = cv.take<float>((size_t)pl.n_chunks * n_cx * nb);
    float2* xs = nullptr;
    float2* pxy = nullptr;
    float* pyy = nullptr;
    if (need_xs) {
        xs = cv.take<float2>((size_t)n_cx * n_frames * nb);
        pxy = cv.take<float2>((size_t)pl.n_chunks * n_cy * nb);
        pyy = cv.take<float>((size_t)pl.n_chunks * n_cy * nb);
    }
    {
        XspecArgs a{x, n_samples, ldx, n_cx, W, hop, n_frames, detrend, pl.fpc, window, tw, xs, pxx};
        dim3 grid(pl.n_chunks, n_cx);
        DISPATCH_N(W, CHK(launch(c, "welch_xspec", k_xspec<NN>, grid, Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, a)));
    }
    if (need_xs) {
        YaccArgs a{y, n_samples, ldy, n_cy, n_cx, W, hop, n_frames, detrend, pl.fpc, window, tw, xs, pxy, pyy};
        dim3 grid(pl.n_chunks, (n_cy + 1) / 2);
        DISPATCH_N(W, CHK(launch(c, "welch_yacc", k_yacc<NN>, grid, Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, a)));
    }
    WelchFinArgs f{pxx, pxy, pyy, pl.n_chunks, pl.n_chunks, n_cx, n_cy, kind, mode,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb},
                   out_c, out_r};
    int64_t total = (int64_t)nb * (kind == 1 ? n_cx : n_cy);
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

// Frames the kernels have to visit: a frame that starts at or past the end of the signal is all
// zeros (include/dsptoolbox_amd.h: "zero padded") and adds nothing to any sum, so the pair loops
// stop at the last pair that still overlaps the signal -- no loader ever forms an address from a
// start beyond the data.  The normalisation keeps the caller's frame count.
static int frames_to_visit(int64_t n_samples, int hop, int n_frames) {
    const int64_t pairs = (n_samples + 2 * (int64_t)hop - 1) / (2 * (int64_t)hop);
    return (int)std::min<int64_t>(n_frames, 2 * pairs);
}

// nfft 4096, one input channel: register-resident radix-16 FFT path (kernels_welch4096.hpp)
static int welch4096_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, const float* y, int n_cy, int64_t ldy,
                         int64_t n_samples, int hop, int n_frames, const float* window, int detrend,
                         int mode, int amp_sqrt, double norm_scale, double factor, int halve_edges,
                         float2* tf, float* coh, int kind = 0) {  // kind 2: tf = cross spectra, no coh
    namespace w4 = welch4096;
    if (!x || !y || !window) return fail(c, DS_ERR_ARG, "ds_welch_tf: null argument");
    if (n_cx != 1 && n_cx != n_cy) return fail(c, DS_ERR_ARG, "ds_welch_tf: one input channel, or one per output channel");
    if (n_cy <= 0 || n_samples <= 0 || hop <= 0 || hop > 4096 || n_frames <= 0 || ldx < n_samples ||
        ldy < n_samples)
        return fail(c, DS_ERR_ARG, "ds_welch_tf: bad shape");
    if (mode < DS_TF_H1 || mode > DS_TF_H3) return fail(c, DS_ERR_ARG, "welch: unsupported transfer function type");
    if (!c->w4_tables) {
        std::vector<float2> h;
        w4::host_tables(h);
        HIPCHK(c, hipMalloc((void**)&c->w4_tables, sizeof(float2) * h.size()));
        HIPCHK(c, hipMemcpyAsync(c->w4_tables, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const int nf = frames_to_visit(n_samples, hop, n_frames);
    // 50 % overlap: three workgroups per CU (kernels_welch4096w.hpp); any other hop: two
    const bool half = hop == 2048;
    const bool three = half && !c->cfg.w4_two_per_cu && w4::fits3(n_samples, nf);
    w4::Plan pl = three ? w4::plan3(nf, n_cy, c->cfg.welch_chunks) : w4::plan(nf, n_cy, c->cfg.welch_chunks);
    CHK(reserve(c, &c->ws, &c->ws_bytes, pl.bytes + (size_t)(n_cx - 1) * (Carver::pad(sizeof(float2) * (size_t)pl.n_pairs * w4::N) +
                                                                  Carver::pad(sizeof(float) * (size_t)pl.n_pairs * w4::NB) +
                                                                  Carver::pad(sizeof(float) * (size_t)pl.n_chunks * w4::NB)) + 4096));
    Carver cv(c->ws);
    float2* xs = cv.take<float2>((size_t)n_cx * pl.n_pairs * w4::N);
    float* px = cv.take<float>((size_t)n_cx * pl.n_pairs * w4::NB);
    float* psx = cv.take<float>((size_t)pl.n_chunks * n_cx * w4::NB);
    float2* pxy = cv.take<float2>((size_t)pl.n_chunks * n_cy * w4::NB);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_cy * w4::NB);
    w4::Args ax{x, n_samples, ldx, 1, hop, nf, pl.n_pairs, detrend, pl.n_chunks, pl.ppc, window,
                c->w4_tables, xs, px, pxy, pyy, psx};
    ax.n_cx = n_cx;
    w4::Args ay = ax;
    ay.sig = y;
    ay.ld = ldy;
    ay.n_ch = n_cy;
    if (three) w4::place_remainder(ay, n_cy);
    if (three) {
        CHK(launch(c, "welch4096_x", w4::k_x3, dim3(pl.n_pairs * n_cx), w4::NT, w4::LDS3_BYTES, ax));
        if (n_cx > 1) CHK(launch(c, "welch4096_pxsum", w4::k_px_sum, dim3(pl.n_chunks, n_cx), 256, 0, ay));
        CHK(launch(c, "welch4096_main@3", w4::k_y3<false>, dim3(pl.n_chunks * n_cy), w4::NT, w4::LDS3_BYTES, ay));
    } else {
        auto kx = half ? w4::k_x<true> : w4::k_x<false>;
        auto ky = half ? w4::k_y<true> : w4::k_y<false>;
        CHK(launch(c, "welch4096_x", kx, dim3(pl.n_pairs, n_cx), w4::NT, w4::LDS_BYTES, ax));
        if (n_cx > 1) CHK(launch(c, "welch4096_pxsum", w4::k_px_sum, dim3(pl.n_chunks, n_cx), 256, 0, ay));
        CHK(launch(c, "welch4096_main@2", ky, dim3(pl.n_chunks * n_cy), w4::NT, w4::LDS_BYTES_2, ay));
    }
    WelchFinArgs f{psx, pxy, pyy, pl.n_chunks, pl.n_chunks, n_cx, n_cy, kind, mode,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, w4::NB},
                   tf, coh};
    int64_t total = (int64_t)w4::NB * n_cy;
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

// window 2048 at 50 % overlap: two 2048-point pair transforms per pass of the 4096-point register machine
// (kernels_welch2048h.hpp).  y == nullptr: auto spectra of x only (psd in `coh`).
static bool welch2048h_applies(const ds_ctx* c, int W, int hop, int average, int64_t n_samples, int n_frames) {
    return c && W == 2048 && hop == 1024 && average == DS_AVG_MEAN && !c->cfg.welch_generic && !c->cfg.w2048_wave &&
           welch2048h::fits(n_samples, n_frames);
}
static int welch2048h_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, const float* y, int n_cy, int64_t ldy,
                          int64_t n_samples, int n_frames, const float* window, int detrend, int mode, int amp_sqrt,
                          double norm_scale, double factor, int halve_edges, float2* tf, float* coh, int kind = 0) {
    namespace wh = welch2048h;
    namespace w4 = welch4096;
    const bool auto_only = kind == 1;
    if (!x || !window || (!auto_only && !y)) return fail(c, DS_ERR_ARG, "ds_welch: null argument");
    if (!auto_only && n_cx != 1 && n_cx != n_cy) return fail(c, DS_ERR_ARG, "ds_welch_tf: one input channel, or one per output channel");
    if (n_cx <= 0 || n_samples <= 0 || n_frames <= 0 || ldx < n_samples || (!auto_only && (n_cy <= 0 || ldy < n_samples)))
        return fail(c, DS_ERR_ARG, "ds_welch: bad shape");
    if (kind == 0 && (mode < DS_TF_H1 || mode > DS_TF_H3)) return fail(c, DS_ERR_ARG, "welch: unsupported transfer function type");
    if (!c->w4_tables) {
        std::vector<float2> h;
        w4::host_tables(h);
        CHK(upload_table_fwd(c, &c->w4_tables, h));
    }
    const int nf = frames_to_visit(n_samples, wh::HOP, n_frames);
    const int n_out = auto_only ? n_cx : n_cy;
    wh::Plan pl = wh::plan(nf, n_out, auto_only ? 0 : n_cx, c->cfg.welch_chunks);
    CHK(reserve(c, &c->ws, &c->ws_bytes, pl.bytes + 4096));
    Carver cv(c->ws);
    float2* xs = auto_only ? nullptr : cv.take<float2>((size_t)n_cx * pl.n_passes * wh::PASS);
    float* px = auto_only ? nullptr : cv.take<float>((size_t)n_cx * pl.n_passes * wh::NBW);
    float* psx = auto_only ? nullptr : cv.take<float>((size_t)pl.n_chunks * n_cx * wh::NBW);
    float2* pxy = auto_only ? nullptr : cv.take<float2>((size_t)pl.n_chunks * n_cy * wh::NBW);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_out * wh::NBW);
    // (Args::n_pairs counts passes of four frames here)
    w4::Args ax{x, n_samples, ldx, 1, wh::HOP, nf, pl.n_passes, detrend, pl.n_chunks, 0, window, c->w4_tables, xs, px, pxy, pyy, psx};
    ax.n_cx = n_cx;
    if (auto_only) {
        ax.n_ch = n_cx;
        w4::place_remainder(ax, n_cx);
        CHK(launch(c, "welch2048_main@4k", wh::k_y2h<true>, dim3(pl.n_chunks * n_cx), w4::NT, wh::LDS_BYTES, ax));
        WelchFinArgs f{pyy, nullptr, nullptr, pl.n_chunks, pl.n_chunks, n_cx, 0, 1, 0,
                       FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, wh::NBW}, nullptr, coh};
        return launch_finish(c, dim3((unsigned)(((int64_t)wh::NBW * n_cx + 63) / 64)), f);
    }
    w4::Args ay = ax;
    ay.sig = y;
    ay.ld = ldy;
    ay.n_ch = n_cy;
    w4::place_remainder(ay, n_cy);
    CHK(launch(c, "welch2048_x", wh::k_x2h, dim3(pl.n_passes * n_cx), w4::NT, wh::LDS_BYTES, ax));
    if (n_cx > 1) CHK(launch(c, "welch2048_pxsum", wh::k_px_sum, dim3(pl.n_chunks, n_cx), 256, 0, ay));
    CHK(launch(c, "welch2048_main@4k", wh::k_y2h<false>, dim3(pl.n_chunks * n_cy), w4::NT, wh::LDS_BYTES, ay));
    WelchFinArgs f{psx, pxy, pyy, pl.n_chunks, pl.n_chunks, n_cx, n_cy, kind, mode,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, wh::NBW}, tf, coh};
    return launch_finish(c, dim3((unsigned)(((int64_t)wh::NBW * n_cy + 63) / 64)), f);
}

// window 8192, one input channel: two 4096-point register transforms per frame pair
// (kernels_welch8192.hpp)
static int welch8192_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, const float* y, int n_cy, int64_t ldy,
                         int64_t n_samples, int hop, int n_frames, const float* window, int detrend,
                         int mode, int amp_sqrt, double norm_scale, double factor, int halve_edges,
                         float2* tf, float* coh, int kind = 0) {  // kind 2: tf = cross spectra, no coh
    namespace w8 = welch8k;
    if (!x || !y || !window) return fail(c, DS_ERR_ARG, "ds_welch_tf: null argument");
    if (n_cy <= 0 || n_samples <= 0 || hop <= 0 || hop > 8192 || n_frames <= 0 || ldx < n_samples ||
        ldy < n_samples)
        return fail(c, DS_ERR_ARG, "ds_welch_tf: bad shape");
    if (mode < DS_TF_H1 || mode > DS_TF_H3) return fail(c, DS_ERR_ARG, "welch: unsupported transfer function type");
    if (!c->w4_tables) {
        std::vector<float2> h;
        welch4096::host_tables(h);
        CHK(upload_table_fwd(c, &c->w4_tables, h));
    }
    if (!c->deconv8k_tables) {
        std::vector<float2> h;
        deconv8k::host_tables(h);
        CHK(upload_table_fwd(c, &c->deconv8k_tables, h));
    }
    const int nf = frames_to_visit(n_samples, hop, n_frames);
    if (n_cx != 1 && n_cx != n_cy) return fail(c, DS_ERR_ARG, "ds_welch_tf: one input channel, or one per output channel");
    w8::Plan pl = w8::plan(nf, n_cy, n_cx);
    CHK(reserve(c, &c->ws, &c->ws_bytes, pl.bytes));
    Carver cv(c->ws);
    float2* xs = cv.take<float2>((size_t)n_cx * pl.n_pairs * w8::N);
    float* px = cv.take<float>((size_t)n_cx * pl.n_pairs * w8::NB);
    float* psx = cv.take<float>((size_t)pl.n_chunks * n_cx * w8::NB);
    float2* pxy = cv.take<float2>((size_t)pl.n_chunks * n_cy * w8::NB);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_cy * w8::NB);
    const bool half = hop == 4096;
    w8::Args ax{x, n_samples, ldx, n_cx, hop, nf, pl.n_pairs, detrend, pl.n_chunks, window,
                c->w4_tables, c->deconv8k_tables, (float4*)xs, px, pxy, pyy, psx, n_cx};
    auto kx = half ? w8::k_x<true> : w8::k_x<false>;
    CHK(launch(c, "welch8192_x", kx, dim3(pl.n_pairs, n_cx), w8::NTB, w8::LDS_BYTES, ax));
    if (n_cx > 1) CHK(launch(c, "welch8192_pxsum", w8::k_px_sum, dim3(pl.n_chunks, n_cx), 256, 0, ax));
    w8::Args ay = ax;
    ay.sig = y;
    ay.ld = ldy;
    ay.n_ch = n_cy;
    // window in LDS + one exchange buffer per group (0.226 ms; the global-window / two-buffer
    // variant measured 0.280 ms and spilled: removed)
    {
        auto kyw = half ? w8::k_y<true, true> : w8::k_y<false, true>;
        CHK(launch(c, "welch8192_main", kyw, dim3(pl.n_chunks * n_cy), w8::NTB, w8::LDS_BYTES_WINLDS, ay));
    }
    WelchFinArgs f{psx, pxy, pyy, pl.n_chunks, pl.n_chunks, n_cx, n_cy, kind, mode,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, w8::NB},
                   tf, coh};
    int64_t total = (int64_t)w8::NB * n_cy;
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

// auto spectra of every channel with an 8192-sample window (AUTO variant of welch8k::k_y)
static int welch8192_psd_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, int64_t n_samples, int hop,
                             int n_frames, const float* window, int detrend, int amp_sqrt, double norm_scale,
                             double factor, int halve_edges, float* psd) {
    namespace w8 = welch8k;
    if (!x || !window) return fail(c, DS_ERR_ARG, "ds_welch_psd: null argument");
    if (n_cx <= 0 || n_samples <= 0 || hop <= 0 || hop > 8192 || n_frames <= 0 || ldx < n_samples)
        return fail(c, DS_ERR_ARG, "ds_welch_psd: bad shape");
    if (!c->w4_tables) {
        std::vector<float2> h;
        welch4096::host_tables(h);
        CHK(upload_table_fwd(c, &c->w4_tables, h));
    }
    if (!c->deconv8k_tables) {
        std::vector<float2> h;
        deconv8k::host_tables(h);
        CHK(upload_table_fwd(c, &c->deconv8k_tables, h));
    }
    const int nf = frames_to_visit(n_samples, hop, n_frames);
    w8::Plan pl = w8::plan(nf, n_cx);
    CHK(reserve(c, &c->ws, &c->ws_bytes, Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * w8::NB)));
    Carver cv(c->ws);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_cx * w8::NB);
    w8::Args a{x, n_samples, ldx, n_cx, hop, nf, pl.n_pairs, detrend, pl.n_chunks, window,
               c->w4_tables, c->deconv8k_tables, nullptr, nullptr, nullptr, pyy, nullptr};
    auto ky = hop == 4096 ? w8::k_y<true, true, true> : w8::k_y<false, true, true>;
    CHK(launch(c, "welch8192_main", ky, dim3(pl.n_chunks * n_cx), w8::NTB, w8::LDS_BYTES_WINLDS, a));
    WelchFinArgs f{pyy, nullptr, nullptr, pl.n_chunks, pl.n_chunks, n_cx, 0, 1, 0,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, w8::NB},
                   nullptr, psd};
    int64_t total = (int64_t)w8::NB * n_cx;
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

// window 16384: four 4096-point register transforms per frame pair, two per slot of 256 threads
// (kernels_welch16384.hpp)
static int welch16k_tables(ds_ctx* c) {
    if (!c->w4_tables) {
        std::vector<float2> h;
        welch4096::host_tables(h);
        CHK(upload_table_fwd(c, &c->w4_tables, h));
    }
    if (!c->fir16k_tables) {
        std::vector<float2> h;
        fir16k::host_tables(h);
        CHK(upload_table_fwd(c, &c->fir16k_tables, h));
    }
    return DS_OK;
}
static int welch16384_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, const float* y, int n_cy, int64_t ldy,
                          int64_t n_samples, int hop, int n_frames, const float* window, int detrend,
                          int mode, int amp_sqrt, double norm_scale, double factor, int halve_edges,
                          float2* tf, float* coh, int kind = 0) {  // kind 2: tf = cross spectra, no coh
    namespace w16 = welch16k;
    if (!x || !y || !window) return fail(c, DS_ERR_ARG, "ds_welch_tf: null argument");
    if (n_cx != 1 && n_cx != n_cy) return fail(c, DS_ERR_ARG, "ds_welch_tf: one input channel, or one per output channel");
    if (n_cy <= 0 || n_samples <= 0 || hop <= 0 || hop > 16384 || n_frames <= 0 || ldx < n_samples ||
        ldy < n_samples)
        return fail(c, DS_ERR_ARG, "ds_welch_tf: bad shape");
    if (mode < DS_TF_H1 || mode > DS_TF_H3) return fail(c, DS_ERR_ARG, "welch: unsupported transfer function type");
    CHK(welch16k_tables(c));
    const int nf = frames_to_visit(n_samples, hop, n_frames);
    w16::Plan pl = w16::plan(nf, n_cy, n_cx);
    CHK(reserve(c, &c->ws, &c->ws_bytes, pl.bytes));
    Carver cv(c->ws);
    float2* xs = cv.take<float2>((size_t)n_cx * pl.n_pairs * w16::N);
    float* pxu = cv.take<float>((size_t)n_cx * pl.n_pairs * w16::N);
    float* psx = cv.take<float>((size_t)pl.n_chunks * n_cx * w16::NB);
    float2* pxy = cv.take<float2>((size_t)pl.n_chunks * n_cy * w16::NB);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_cy * w16::NB);
    float2* tu = cv.take<float2>((size_t)pl.n_chunks * n_cy * w16::N);
    float* pu = cv.take<float>((size_t)pl.n_chunks * n_cy * w16::N);
    w16::Args ax{x, n_samples, ldx, n_cx, hop, nf, pl.n_pairs, detrend, pl.n_chunks, window,
                 c->w4_tables, c->fir16k_tables, (float4*)xs, pxu, pxy, pyy, psx, n_cx, tu, pu};
    CHK(launch(c, "welch16384_x", w16::k_x, dim3(pl.n_pairs, n_cx, 4), w16::NTB, w16::LDS_BYTES, ax));
    CHK(launch(c, "welch16384_pxsum", w16::k_px_sum, dim3((w16::NB + 255) / 256, pl.n_chunks, n_cx), 256, 0, ax));
    w16::Args ay = ax;
    ay.sig = y;
    ay.ld = ldy;
    ay.n_ch = n_cy;
    CHK(launch(c, "welch16384_main", w16::k_y<false>, dim3(pl.n_chunks * n_cy, 1, 4), w16::NTB, w16::LDS_BYTES, ay));
    CHK(launch(c, "welch16384_fold", w16::k_fold<false>, dim3((w16::NB + 255) / 256, pl.n_chunks * n_cy), 256, 0, ay));
    WelchFinArgs f{psx, pxy, pyy, pl.n_chunks, pl.n_chunks, n_cx, n_cy, kind, mode,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, w16::NB},
                   tf, coh};
    int64_t total = (int64_t)w16::NB * n_cy;
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}
// Windows of 2^15 ... 2^18 samples: decimation in frequency into R = W / 4096 class sequences (k_dif), the headline
// kernel's loop on them (k_xc / k_yc), fold across the classes, finish (kernels_welch_long.hpp).
// y == nullptr: auto spectra of x only (ds_welch_psd), the result in `coh`.
static bool welch_long_applies(const ds_ctx* c, int W, int n_ch_total, int64_t n_samples, int n_frames, int hop, int average) {
    if (!c || c->cfg.welch_generic || average != DS_AVG_MEAN || !welchl::classes_of(W) || W < c->cfg.welch_long_min) return false;
    if (hop <= 0 || hop > W || !welchl::buf_fits(n_samples, n_frames, hop, W)) return false;
    // the class sequences: one complex value per sample of every frame pair (8 bytes per sample at 50 % overlap)
    const int64_t pairs = ((int64_t)frames_to_visit(n_samples, hop, n_frames) + 1) / 2;
    // launch grids: k_dif puts the frame pairs on grid.y and the channels on grid.z, k_fold (chunk, channel) units on
    // grid.y (chunks <= max(768 / R, pairs / 64), kernels_welch_long.hpp plan()); shapes beyond 65535 there (99 %
    // overlap on 2^25 samples, ...) fall through to the routes behind this one
    const int64_t chunks_max = std::max<int64_t>(768 / welchl::classes_of(W) + 1, (pairs + 63) / 64);
    if (pairs > 65535 || n_ch_total > 65535 || chunks_max * n_ch_total > 65535) return false;
    return (int64_t)n_ch_total * pairs * W * 8 <= ((int64_t)16 << 30);
}
static int welch_long_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, const float* y, int n_cy, int64_t ldy,
                          int64_t n_samples, int W, int hop, int n_frames, const float* window, int detrend, int mode,
                          int amp_sqrt, double norm_scale, double factor, int halve_edges, float2* tf, float* coh,
                          int kind = 0) {  // kind 0: tf + coherence, 1: auto spectra of x (y null), 2: cross spectra in tf
    namespace wl = welchl;
    const bool auto_only = kind == 1;
    if (!x || !window || (!auto_only && !y)) return fail(c, DS_ERR_ARG, "ds_welch: null argument");
    if (!auto_only && n_cx != 1 && n_cx != n_cy) return fail(c, DS_ERR_ARG, "ds_welch_tf: one input channel, or one per output channel");
    if (n_cx <= 0 || n_samples <= 0 || n_frames <= 0 || ldx < n_samples || (!auto_only && (n_cy <= 0 || ldy < n_samples)))
        return fail(c, DS_ERR_ARG, "ds_welch: bad shape");
    if (!auto_only && kind == 0 && (mode < DS_TF_H1 || mode > DS_TF_H3)) return fail(c, DS_ERR_ARG, "welch: unsupported transfer function type");
    const int R = wl::classes_of(W);
    int lgR = 0;
    while ((1 << lgR) < R) ++lgR;
    if (!c->w4_tables) {
        std::vector<float2> h;
        welch4096::host_tables(h);
        CHK(upload_table_fwd(c, &c->w4_tables, h));
    }
    float2** slot = &c->wl_tables[lgR - 1];
    if (!*slot) {
        std::vector<float2> h;
        wl::host_tables(R, h);
        CHK(upload_table_fwd(c, slot, h));
    }
    const int nf = frames_to_visit(n_samples, hop, n_frames), nb = W / 2 + 1;
    const int n_out = auto_only ? n_cx : n_cy;  // channels that are accumulated
    wl::Plan pl = wl::plan(nf, n_out, R);
    const size_t seq = (size_t)pl.n_pairs * W;  // complex values per channel
    CHK(reserve(c, &c->ws, &c->ws_bytes,
                Carver::pad(sizeof(float2) * seq * n_cx) + (auto_only ? 0 : Carver::pad(sizeof(float2) * seq * n_cy)) +
                    (auto_only ? 0 : Carver::pad(sizeof(float2) * seq * n_cx) + Carver::pad(sizeof(float) * seq * n_cx) +
                                         Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * nb) +
                                         Carver::pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * nb) +
                                         Carver::pad(sizeof(float2) * (size_t)pl.n_chunks * n_cy * W)) +
                    Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_out * nb) + Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_out * W)));
    Carver cv(c->ws);
    float2* bx = cv.take<float2>(seq * n_cx);
    float2* by = auto_only ? nullptr : cv.take<float2>(seq * n_cy);
    float2* xs = auto_only ? nullptr : cv.take<float2>(seq * n_cx);
    float* pxu = auto_only ? nullptr : cv.take<float>(seq * n_cx);
    float* psx = auto_only ? nullptr : cv.take<float>((size_t)pl.n_chunks * n_cx * nb);
    float2* pxy = auto_only ? nullptr : cv.take<float2>((size_t)pl.n_chunks * n_cy * nb);
    float2* tu = auto_only ? nullptr : cv.take<float2>((size_t)pl.n_chunks * n_cy * W);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_out * nb);
    float* pu = cv.take<float>((size_t)pl.n_chunks * n_out * W);
    wl::Args ax{x, n_samples, ldx, n_cx, hop, nf, pl.n_pairs, detrend, pl.n_chunks, R, lgR, window, c->w4_tables, *slot,
                bx, (float4*)xs, pxu, pxy, pyy, psx, n_cx, tu, pu};
    auto dif = [&](const wl::Args& a, int n_ch) {
        const dim3 grid(wl::M / wl::NT, pl.n_pairs, n_ch);
        switch (R) {
            case 4: return launch(c, "welch_long_dif", wl::k_dif<4>, grid, wl::NT, 0, a);
            case 8: return launch(c, "welch_long_dif", wl::k_dif<8>, grid, wl::NT, 0, a);
            case 16: return launch(c, "welch_long_dif", wl::k_dif<16>, grid, wl::NT, 0, a);
            case 32: return launch(c, "welch_long_dif", wl::k_dif<32>, grid, wl::NT, 0, a);
            default: return launch(c, "welch_long_dif", wl::k_dif<64>, grid, wl::NT, 0, a);
        }
    };
    CHK(dif(ax, n_cx));
    const dim3 fold_grid((nb + 255) / 256, pl.n_chunks * n_out);
    if (auto_only) {
        CHK(launch(c, "welch_long_main", wl::k_yc<true>, dim3((unsigned)(pl.n_chunks * n_cx * R)), wl::NT, wl::LDS_BYTES, ax));
        CHK(launch(c, "welch_long_fold", wl::k_fold<true>, fold_grid, 256, 0, ax));
        WelchFinArgs f{pyy, nullptr, nullptr, pl.n_chunks, pl.n_chunks, n_cx, 0, 1, 0,
                       FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb}, nullptr, coh};
        return launch_finish(c, dim3((unsigned)(((int64_t)nb * n_cx + 63) / 64)), f);
    }
    CHK(launch(c, "welch_long_x", wl::k_xc, dim3((unsigned)(pl.n_pairs * R * n_cx)), wl::NT, wl::LDS_BYTES, ax));
    CHK(launch(c, "welch_long_pxsum", wl::k_px_sum, dim3((nb + 255) / 256, pl.n_chunks, n_cx), 256, 0, ax));
    wl::Args ay = ax;
    ay.sig = y;
    ay.ld = ldy;
    ay.n_ch = n_cy;
    ay.b = by;
    CHK(dif(ay, n_cy));
    if (c->cfg.welch_long_3percu)
        CHK(launch(c, "welch_long_main@jit", wl::k_yc<false, true>, dim3((unsigned)(pl.n_chunks * n_cy * R)), wl::NT, wl::LDS_BYTES, ay));
    else
        CHK(launch(c, "welch_long_main", wl::k_yc<false>, dim3((unsigned)(pl.n_chunks * n_cy * R)), wl::NT, wl::LDS_BYTES, ay));
    CHK(launch(c, "welch_long_fold", wl::k_fold<false>, fold_grid, 256, 0, ay));
    WelchFinArgs f{psx, pxy, pyy, pl.n_chunks, pl.n_chunks, n_cx, n_cy, kind, mode,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb}, tf, coh};
    return launch_finish(c, dim3((unsigned)(((int64_t)nb * n_cy + 63) / 64)), f);
}

static int welch16384_psd_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, int64_t n_samples, int hop,
                              int n_frames, const float* window, int detrend, int amp_sqrt, double norm_scale,
                              double factor, int halve_edges, float* psd) {
    namespace w16 = welch16k;
    if (!x || !window) return fail(c, DS_ERR_ARG, "ds_welch_psd: null argument");
    if (n_cx <= 0 || n_samples <= 0 || hop <= 0 || hop > 16384 || n_frames <= 0 || ldx < n_samples)
        return fail(c, DS_ERR_ARG, "ds_welch_psd: bad shape");
    CHK(welch16k_tables(c));
    const int nf = frames_to_visit(n_samples, hop, n_frames);
    w16::Plan pl = w16::plan(nf, n_cx);
    CHK(reserve(c, &c->ws, &c->ws_bytes, Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * w16::NB) +
                                             Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * w16::N)));
    Carver cv(c->ws);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_cx * w16::NB);
    float* pu = cv.take<float>((size_t)pl.n_chunks * n_cx * w16::N);
    w16::Args a{x, n_samples, ldx, n_cx, hop, nf, pl.n_pairs, detrend, pl.n_chunks, window,
                c->w4_tables, c->fir16k_tables, nullptr, nullptr, nullptr, pyy, nullptr, 1, nullptr, pu};
    CHK(launch(c, "welch16384_main", w16::k_y<true>, dim3(pl.n_chunks * n_cx, 1, 4), w16::NTB, w16::LDS_BYTES, a));
    CHK(launch(c, "welch16384_fold", w16::k_fold<true>, dim3((w16::NB + 255) / 256, pl.n_chunks * n_cx), 256, 0, a));
    WelchFinArgs f{pyy, nullptr, nullptr, pl.n_chunks, pl.n_chunks, n_cx, 0, 1, 0,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, w16::NB},
                   nullptr, psd};
    int64_t total = (int64_t)w16::NB * n_cx;
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

// auto spectra of every channel with a 4096-sample window on the headline kernel (AUTO variant)
static int welch4096_psd_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, int64_t n_samples, int hop,
                             int n_frames, const float* window, int detrend, int amp_sqrt, double norm_scale,
                             double factor, int halve_edges, float* psd) {
    namespace w4 = welch4096;
    if (!x || !window) return fail(c, DS_ERR_ARG, "ds_welch_psd: null argument");
    if (n_cx <= 0 || n_samples <= 0 || hop <= 0 || hop > 4096 || n_frames <= 0 || ldx < n_samples)
        return fail(c, DS_ERR_ARG, "ds_welch_psd: bad shape");
    if (!c->w4_tables) {
        std::vector<float2> h;
        w4::host_tables(h);
        CHK(upload_table_fwd(c, &c->w4_tables, h));
    }
    const int nf = frames_to_visit(n_samples, hop, n_frames);
    const bool three = hop == 2048 && !c->cfg.w4_two_per_cu && w4::fits3(n_samples, nf);
    w4::Plan pl = three ? w4::plan3(nf, n_cx, c->cfg.welch_chunks) : w4::plan(nf, n_cx, c->cfg.welch_chunks);
    CHK(reserve(c, &c->ws, &c->ws_bytes, Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * w4::NB)));
    Carver cv(c->ws);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_cx * w4::NB);
    w4::Args a{x, n_samples, ldx, n_cx, hop, nf, pl.n_pairs, detrend, pl.n_chunks, pl.ppc, window,
               c->w4_tables, nullptr, nullptr, nullptr, pyy, nullptr};
    if (three) {
        w4::place_remainder(a, n_cx);
        CHK(launch(c, "welch4096_main@3", w4::k_y3<true>, dim3(pl.n_chunks * n_cx), w4::NT, w4::LDS3_BYTES, a));
    } else {
        auto ky = hop == 2048 ? w4::k_y<true, true> : w4::k_y<false, true>;
        CHK(launch(c, "welch4096_main@2", ky, dim3(pl.n_chunks * n_cx), w4::NT, w4::LDS_BYTES_2, a));
    }
    WelchFinArgs f{pyy, nullptr, nullptr, pl.n_chunks, pl.n_chunks, n_cx, 0, 1, 0,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, w4::NB},
                   nullptr, psd};
    int64_t total = (int64_t)w4::NB * n_cx;
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

// twiddle tables of the wave-level transforms (stft1k::host_tables<N>), cached per context
template <int NN>
static int wave_tables(ds_ctx* c, const float2** out) {
    float2** tab = NN == 1024 ? &c->stft1k_tables : &c->stft_wave_tables[NN == 512 ? 0 : (NN == 256 ? 1 : 2)];
    if (!*tab) {
        std::vector<float2> h;
        stft1k::host_tables<NN>(h);
        CHK(upload_table_fwd(c, tab, h));
    }
    *out = *tab;
    return DS_OK;
}

// windows of 256 / 512 / 1024 samples (1024 = the reference's default), one input channel:
// wave-level register transforms (kernels_welch1024.hpp)
template <int NN>
static int welch_wave_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, const float* y, int n_cy, int64_t ldy,
                          int64_t n_samples, int hop, int n_frames, const float* window, int detrend,
                          int mode, int amp_sqrt, double norm_scale, double factor, int halve_edges,
                          float2* tf, float* coh, int kind = 0, int decim = 1) {  // kind 2: tf = cross spectra, no coh
    // decim = D > 1: windows of NN / D samples (`window` holds that many values): the frames are transformed zero-padded
    // to NN points and every D-th bin is kept (removing a frame's mean still only clears bin 0 of the kept bins)
    namespace w1 = welch1k;
    using W = w1::WG<NN>;
    if (!x || !y || !window) return fail(c, DS_ERR_ARG, "ds_welch_tf: null argument");
    if (n_cy <= 0 || n_samples <= 0 || hop <= 0 || hop > NN || n_frames <= 0 || ldx < n_samples ||
        ldy < n_samples)
        return fail(c, DS_ERR_ARG, "ds_welch_tf: bad shape");
    if (mode < DS_TF_H1 || mode > DS_TF_H3) return fail(c, DS_ERR_ARG, "welch: unsupported transfer function type");
    const float2* tab;
    CHK(wave_tables<NN>(c, &tab));
    const int nf = frames_to_visit(n_samples, hop, n_frames);
    if (n_cx != 1 && n_cx != n_cy) return fail(c, DS_ERR_ARG, "ds_welch_tf: one input channel, or one per output channel");
    w1::Plan pl = w1::plan<NN>(nf, n_cy, n_cx, c->cfg.welch1k_chunks);
    CHK(reserve(c, &c->ws, &c->ws_bytes, pl.bytes + Carver::pad(sizeof(float) * NN)));
    Carver cv(c->ws);
    if (decim > 1) {  // the window, zero-padded to the transform length
        float* wz = cv.take<float>(NN);
        HIPCHK(c, hipMemsetAsync(wz, 0, sizeof(float) * NN, c->stream));
        HIPCHK(c, hipMemcpyAsync(wz, window, sizeof(float) * (NN / decim), hipMemcpyDeviceToDevice, c->stream));
        window = wz;
    }
    float2* xs = cv.take<float2>((size_t)n_cx * pl.n_pairs * NN);
    float* px = cv.take<float>((size_t)n_cx * pl.n_pairs * W::NB);
    float* psx = cv.take<float>((size_t)pl.n_chunks * n_cx * W::NB);
    float2* pxy = cv.take<float2>((size_t)pl.n_chunks * n_cy * W::NB);
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_cy * W::NB);
    const bool half = hop == NN / 2;
    w1::Args ax{x, n_samples, ldx, n_cx, hop, nf, pl.n_pairs, detrend, pl.n_chunks, pl.ppc, window,
                tab, (float4*)xs, px, pxy, pyy, psx, n_cx};
    auto kx = half ? w1::k_x<NN, true> : w1::k_x<NN, false>;
    auto ky = half ? w1::k_y<NN, true> : w1::k_y<NN, false>;
    CHK(launch(c, "welch1024_x", kx, dim3((pl.n_pairs + W::TPB - 1) / W::TPB, n_cx), w1::NTB, W::LDS_BYTES, ax));
    if (n_cx > 1) CHK(launch(c, "welch1024_pxsum", w1::k_px_sum<NN>, dim3(pl.n_chunks, n_cx), 256, 0, ax));
    w1::Args ay = ax;
    ay.sig = y;
    ay.ld = ldy;
    ay.n_ch = n_cy;
    const int n_grp = (n_cy + W::TPB - 1) / W::TPB;
    CHK(launch(c, "welch1024_main", ky, dim3(pl.n_chunks * n_grp), w1::NTB, W::LDS_BYTES, ay));
    const int nb_out = NN / decim / 2 + 1;
    WelchFinArgs f{psx, pxy, pyy, pl.n_chunks, pl.n_chunks, n_cx, n_cy, kind, mode,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb_out},
                   tf, coh, W::NB, decim};
    int64_t total = (int64_t)nb_out * n_cy;
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

// auto spectra of every channel (Signal.get_spectrum's default parameters: window 1024)
template <int NN>
static int welch_wave_psd_run(ds_ctx* c, const float* x, int n_cx, int64_t ldx, int64_t n_samples, int hop,
                              int n_frames, const float* window, int detrend, int amp_sqrt, double norm_scale,
                              double factor, int halve_edges, float* psd, int decim = 1) {
    namespace w1 = welch1k;
    using W = w1::WG<NN>;
    if (!x || !window) return fail(c, DS_ERR_ARG, "ds_welch_psd: null argument");
    if (n_cx <= 0 || n_samples <= 0 || hop <= 0 || hop > NN || n_frames <= 0 || ldx < n_samples)
        return fail(c, DS_ERR_ARG, "ds_welch_psd: bad shape");
    const float2* tab;
    CHK(wave_tables<NN>(c, &tab));
    const int nf = frames_to_visit(n_samples, hop, n_frames);
    w1::Plan pl = w1::plan<NN>(nf, n_cx, 1, c->cfg.welch1k_chunks);
    CHK(reserve(c, &c->ws, &c->ws_bytes,
                Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_cx * W::NB) + Carver::pad(sizeof(float) * NN)));
    Carver cv(c->ws);
    if (decim > 1) {  // the window, zero-padded to the transform length (see welch_wave_run)
        float* wz = cv.take<float>(NN);
        HIPCHK(c, hipMemsetAsync(wz, 0, sizeof(float) * NN, c->stream));
        HIPCHK(c, hipMemcpyAsync(wz, window, sizeof(float) * (NN / decim), hipMemcpyDeviceToDevice, c->stream));
        window = wz;
    }
    float* pyy = cv.take<float>((size_t)pl.n_chunks * n_cx * W::NB);
    w1::Args a{x, n_samples, ldx, n_cx, hop, nf, pl.n_pairs, detrend, pl.n_chunks, pl.ppc, window,
               tab, nullptr, nullptr, nullptr, pyy, nullptr, 1};
    auto ky = hop == NN / 2 ? w1::k_y<NN, true, true> : w1::k_y<NN, false, true>;
    const int n_grp = (n_cx + W::TPB - 1) / W::TPB;
    CHK(launch(c, "welch1024_main", ky, dim3(pl.n_chunks * n_grp), w1::NTB, W::LDS_BYTES, a));
    const int nb_out = NN / decim / 2 + 1;
    WelchFinArgs f{pyy, nullptr, nullptr, pl.n_chunks, pl.n_chunks, n_cx, 0, 1, 0,
                   FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb_out},
                   nullptr, psd, W::NB, decim};
    int64_t total = (int64_t)nb_out * n_cx;
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

extern "C" int ds_welch_tf_dev(ds_ctx* c, const float* x, int n_cx, int64_t ldx, const float* y,
                               int n_cy, int64_t ldy, int64_t n_samples, int W, int hop, int n_frames,
                               const float* window, int detrend, int average, int mode, int amp_sqrt,
                               double norm_scale, double factor, int halve_edges, ds_c32* tf,
                               float* coh) {
    if (!tf || !coh) return fail(c, DS_ERR_ARG, "ds_welch_tf: null output");
    // one input channel, or one per output channel (three-per-CU kernels at 50 % overlap, two-per-CU otherwise)
    if (c && W == 4096 && average == DS_AVG_MEAN && !c->cfg.no_welch4096 && (n_cx == 1 || n_cx == n_cy))
        return welch4096_run(c, x, n_cx, ldx, y, n_cy, ldy, n_samples, hop, n_frames, window, detrend, mode,
                             amp_sqrt, norm_scale, factor, halve_edges, (float2*)tf, coh);
    const bool no1k = c && c->cfg.welch_generic;
    if ((n_cx == 1 || n_cx == n_cy) && welch_long_applies(c, W, n_cx + n_cy, n_samples, n_frames, hop, average))
        return welch_long_run(c, x, n_cx, ldx, y, n_cy, ldy, n_samples, W, hop, n_frames, window, detrend, mode, amp_sqrt,
                              norm_scale, factor, halve_edges, (float2*)tf, coh);
    if (c && W == 16384 && (n_cx == 1 || n_cx == n_cy) && average == DS_AVG_MEAN && !no1k &&
        welch16k::buf_fits(n_samples, n_frames, hop))
        return welch16384_run(c, x, n_cx, ldx, y, n_cy, ldy, n_samples, hop, n_frames, window, detrend, mode,
                              amp_sqrt, norm_scale, factor, halve_edges, (float2*)tf, coh);
    if (c && W == 8192 && (n_cx == 1 || n_cx == n_cy) && average == DS_AVG_MEAN && !no1k &&
        welch8k::buf_fits(n_samples, n_frames, hop))
        return welch8192_run(c, x, n_cx, ldx, y, n_cy, ldy, n_samples, hop, n_frames, window, detrend, mode,
                             amp_sqrt, norm_scale, factor, halve_edges, (float2*)tf, coh);
    if ((n_cx == 1 || n_cx == n_cy) && welch2048h_applies(c, W, hop, average, n_samples, n_frames))
        return welch2048h_run(c, x, n_cx, ldx, y, n_cy, ldy, n_samples, n_frames, window, detrend, mode, amp_sqrt, norm_scale,
                              factor, halve_edges, (float2*)tf, coh);
    // 256 ... 2048-sample windows (1024: the reference's default): one input channel or one per output channel
    if (c && (W == 2048 || W == 1024 || W == 512 || W == 256) && (n_cx == 1 || n_cx == n_cy) && average == DS_AVG_MEAN &&
        !no1k && welch1k::buf_fits(n_samples, n_cy, ldy)) {
        auto run = W == 2048 ? welch_wave_run<2048>
                             : (W == 1024 ? welch_wave_run<1024> : (W == 512 ? welch_wave_run<512> : welch_wave_run<256>));
        return run(c, x, n_cx, ldx, y, n_cy, ldy, n_samples, hop, n_frames, window, detrend, mode, amp_sqrt,
                   norm_scale, factor, halve_edges, (float2*)tf, coh, 0, 1);
    }
    // 128 / 64 / 32-sample windows: every 2nd / 4th / 8th bin of the 256-point kernels on zero-padded frames
    if (c && (W == 128 || W == 64 || W == 32) && (n_cx == 1 || n_cx == n_cy) && average == DS_AVG_MEAN && !no1k &&
        welch1k::buf_fits(n_samples, n_cy, ldy))
        return welch_wave_run<256>(c, x, n_cx, ldx, y, n_cy, ldy, n_samples, hop, n_frames, window, detrend, mode, amp_sqrt,
                                   norm_scale, factor, halve_edges, (float2*)tf, coh, 0, 256 / W);
    return welch_common(c, 0, x, n_cx, ldx, y, n_cy, ldy, n_samples, W, hop, n_frames, window, detrend,
                        average, mode, amp_sqrt, norm_scale, factor, halve_edges, (float2*)tf, coh);
}
// float64 frame spectra of a device-resident (samples, channels) float64 array: W <= 16384 one workgroup per
// (frame, channel); 2^15 ... 2^18 one per (frame, class, channel) + the split (class spectra in the workspace)
static const int kMaxX64Window = 262144;
static int x64_launch_frames(ds_ctx* c, const double* dsig, int n_ch, int64_t n_samples, int W, int hop, int n_frames,
                             int detrend, const double* dw, const double2* tw, double2* spec) {
    int lg = 0;
    while ((1 << lg) < W) ++lg;
    w64::FrameArgs fa{dsig, n_samples, n_ch, W, lg, hop, n_frames, detrend, dw, tw, spec, n_ch, 1};
    if (W <= 16384) {
        // four channels or more: read a planar copy (one 8-byte value per 32-byte sector otherwise)
        if (n_ch >= 4 && (n_samples + 31) / 32 <= 0x7fffffff) {
            CHK(reserve(c, &c->ws, &c->ws_bytes, Carver::pad(sizeof(double) * (size_t)n_ch * n_samples)));
            double* planar = (double*)c->ws;
            hipLaunchKernelGGL(w64::k_planar, dim3((unsigned)((n_samples + 31) / 32), (unsigned)((n_ch + 31) / 32)), dim3(256), 0,
                               c->stream, dsig, n_samples, n_ch, planar);
            HIPCHK(c, hipGetLastError());
            fa.sig = planar;
            fa.s_stride = 1;
            fa.c_stride = n_samples;
        }
        const bool packed = W > 8192;  // the real frame as a W/2-point complex sequence: 128 KB of LDS either way
        const size_t lds = (size_t)(packed ? W / 2 : W) * 16 + 256 * 8;
        auto frames = packed ? w64::k_frames<true> : w64::k_frames<false>;
        return launch(c, "welch_f64_frames", frames, dim3(n_frames, n_ch), 256, lds, fa);
    }
    const int rc = (W / 2) / w64::LONG_M;
    int lg_rc = 0;
    while ((1 << lg_rc) < rc) ++lg_rc;
    if ((int64_t)n_frames * rc > 0x7fffffff || n_ch > 65535 || n_frames > 65535)
        return fail(c, DS_ERR_UNSUP, "float64 Welch route: too many frames / channels for the long-window kernels");
    CHK(reserve(c, &c->ws, &c->ws_bytes, Carver::pad(sizeof(double2) * (size_t)n_ch * n_frames * (W / 2)) +
                                             Carver::pad(sizeof(double) * (size_t)n_ch * n_samples)));
    Carver cvl(c->ws);
    double2* zc = cvl.take<double2>((size_t)n_ch * n_frames * (W / 2));
    double* planar = cvl.take<double>((size_t)n_ch * n_samples);
    if ((n_samples + 31) / 32 > 0x7fffffff) return fail(c, DS_ERR_UNSUP, "float64 Welch route: signal too long for the long-window kernels");
    hipLaunchKernelGGL(w64::k_planar, dim3((unsigned)((n_samples + 31) / 32), (unsigned)((n_ch + 31) / 32)), dim3(256), 0, c->stream,
                       dsig, n_samples, n_ch, planar);
    HIPCHK(c, hipGetLastError());
    w64::LongArgs la{fa, rc, lg_rc, zc, planar};
    CHK(launch(c, "welch_f64_frames@long", w64::k_frames_cls, dim3((unsigned)(n_frames * rc), n_ch), 256,
               (size_t)w64::LONG_M * 16 + 256 * 8, la));
    return launch(c, "welch_f64_split", w64::k_split, dim3((W / 2 + 1 + 255) / 256, n_frames, n_ch), 256, 0, la);
}

// float64 end to end (kernels_welch_f64.hpp): host arrays in the reference's own layout
extern "C" int ds_welch_tf_x64(ds_ctx* c, const double* x, int n_cx, const double* y, int n_cy,
                               int64_t n_samples, int W, int hop, int n_frames, const double* window,
                               int detrend, int average, int mode, int amp_sqrt, double norm_scale,
                               double factor, int halve_edges, double* tf, double* coh) {
    if (!c || !x || !y || !window || !tf || !coh) return fail(c, DS_ERR_ARG, "ds_welch_tf_x64: null argument");
    if (average != DS_AVG_MEAN && average != DS_AVG_MEDIAN)
        return fail(c, DS_ERR_ARG, "welch: average must be mean (0) or median (1)");
    if (average == DS_AVG_MEDIAN && n_frames > 4096)
        return fail(c, DS_ERR_UNSUP, "ds_welch_tf_x64: median averaging over more than 4096 frames (use ds_welch_tf)");
    if (n_cy <= 0 || (n_cx != 1 && n_cx != n_cy) || n_samples <= 0 || hop <= 0 || hop > W || n_frames <= 0)
        return fail(c, DS_ERR_ARG, "ds_welch_tf_x64: bad shape");
    if (!is_pow2(W) || W < 8 || W > kMaxX64Window)
        return fail(c, DS_ERR_UNSUP, "ds_welch_tf_x64: window length must be a power of two in [8, 262144]");
    if (mode < DS_TF_H1 || mode > DS_TF_H3) return fail(c, DS_ERR_ARG, "welch: unsupported transfer function type");
    const int nb = W / 2 + 1;
    const size_t spec_x = (size_t)n_cx * n_frames * nb, spec_y = (size_t)n_cy * n_frames * nb;
    if ((spec_x + spec_y) * sizeof(double2) > ((size_t)2 << 30))
        return fail(c, DS_ERR_UNSUP, "ds_welch_tf_x64: problem too large for the float64 route (use ds_welch_tf)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bx = (size_t)n_samples * n_cx * 8, by = (size_t)n_samples * n_cy * 8;
    const size_t bout = (size_t)nb * n_cy;
    CHK(reserve(c, &c->io, &c->io_bytes,
                Carver::pad(bx) + Carver::pad(by) + Carver::pad((size_t)W * 8) + Carver::pad((size_t)W * 8) +
                    Carver::pad((spec_x + spec_y) * 16) + Carver::pad(bout * 16) + Carver::pad(bout * 8)));
    Carver cv(c->io);
    double* dx = cv.take<double>((size_t)n_samples * n_cx);
    double* dy = cv.take<double>((size_t)n_samples * n_cy);
    double* dw = cv.take<double>(W);
    double2* tw = cv.take<double2>(W / 2);
    double2* xs = cv.take<double2>(spec_x);
    double2* ys = cv.take<double2>(spec_y);
    double2* dtf = cv.take<double2>(bout);
    double* dcoh = cv.take<double>(bout);
    HIPCHK(c, hipMemcpyAsync(dx, x, bx, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dy, y, by, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dw, window, (size_t)W * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(w64::k_twiddles, dim3((W / 2 + 255) / 256), dim3(256), 0, c->stream, tw, W / 2);
    HIPCHK(c, hipGetLastError());
    CHK(x64_launch_frames(c, dx, n_cx, n_samples, W, hop, n_frames, detrend, dw, tw, xs));
    CHK(x64_launch_frames(c, dy, n_cy, n_samples, W, hop, n_frames, detrend, dw, tw, ys));
    if (average == DS_AVG_MEDIAN) {
        const int nbias = (n_frames & 1) ? n_frames : n_frames - 1;
        w64::TfArgs ta{xs, ys, n_cx, n_cy, n_frames, mode,
                       FinishPar{norm_scale * (double)std::max(1, nbias), factor, halve_edges, amp_sqrt, nb}, dtf, dcoh};
        CHK(launch(c, "welch_f64_tf_median", w64::k_tf_median, dim3(nb, n_cy), 256,
                   sizeof(double) * (4 * (size_t)n_frames + 8), ta));
    } else {
        w64::TfArgs ta{xs, ys, n_cx, n_cy, n_frames, mode,
                       FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb}, dtf, dcoh};
        CHK(launch(c, "welch_f64_tf", w64::k_tf, dim3((nb + 255) / 256, n_cy), 256, 0, ta));
    }
    HIPCHK(c, hipMemcpyAsync(tf, dtf, bout * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(coh, dcoh, bout * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DS_OK;
}

// float64 frame spectra of a host (samples, channels) float64 array: upload, window / twiddle tables, k_frames.
// The caller has reserved c->io and carves `dsig` (n_samples * n_ch doubles) and `spec` out of it.
struct X64Tables {
    double* dw = nullptr;
    double2* tw = nullptr;
};
static int x64_tables(ds_ctx* c, Carver& cv, const double* window, int W, X64Tables* t) {
    t->dw = cv.take<double>(W);
    t->tw = cv.take<double2>(W / 2);
    HIPCHK(c, hipMemcpyAsync(t->dw, window, (size_t)W * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(w64::k_twiddles, dim3((W / 2 + 255) / 256), dim3(256), 0, c->stream, t->tw, W / 2);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}
static int x64_frames(ds_ctx* c, const X64Tables& t, const double* sig, double* dsig, int n_ch, int64_t n_samples, int W,
                      int hop, int n_frames, int detrend, double2* spec) {
    HIPCHK(c, hipMemcpyAsync(dsig, sig, (size_t)n_samples * n_ch * 8, hipMemcpyHostToDevice, c->stream));
    return x64_launch_frames(c, dsig, n_ch, n_samples, W, hop, n_frames, detrend, t.dw, t.tw, spec);
}
static int x64_shape_ok(ds_ctx* c, const char* who, int n_ch, int64_t n_samples, int W, int hop, int n_frames, int average) {
    if (average != DS_AVG_MEAN && average != DS_AVG_MEDIAN)
        return fail(c, DS_ERR_ARG, "welch: average must be mean (0) or median (1)");
    if (average == DS_AVG_MEDIAN && n_frames > 4096)
        return fail(c, DS_ERR_UNSUP, std::string(who) + ": median averaging over more than 4096 frames (use the fp32 entry point)");
    if (n_ch <= 0 || n_samples <= 0 || hop <= 0 || hop > W || n_frames <= 0) return fail(c, DS_ERR_ARG, std::string(who) + ": bad shape");
    if (!is_pow2(W) || W < 8 || W > kMaxX64Window)
        return fail(c, DS_ERR_UNSUP, std::string(who) + ": window length must be a power of two in [8, 262144]");
    return DS_OK;
}

// _welch in float64 end to end (auto spectra: y = NULL; cross spectra conj(X_i) Y_i otherwise): the route
// backend._welch takes for SHORT estimates.  out: [nb][n_ch] complex128 (auto spectra: imaginary part 0).
extern "C" int ds_welch_spec_x64(ds_ctx* c, const double* x, const double* y, int n_ch, int64_t n_samples, int W,
                                 int hop, int n_frames, const double* window, int detrend, int average, int amp_sqrt,
                                 double norm_scale, double factor, int halve_edges, double* out) {
    if (!c || !x || !window || !out) return fail(c, DS_ERR_ARG, "ds_welch_spec_x64: null argument");
    CHK(x64_shape_ok(c, "ds_welch_spec_x64", n_ch, n_samples, W, hop, n_frames, average));
    const int nb = W / 2 + 1, n_in = y ? 2 : 1;
    const size_t spec = (size_t)n_ch * n_frames * nb;
    if (spec * n_in * sizeof(double2) > ((size_t)2 << 30))
        return fail(c, DS_ERR_UNSUP, "ds_welch_spec_x64: problem too large for the float64 route (use ds_welch_psd / ds_welch_csd)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bsig = (size_t)n_samples * n_ch * 8, bout = (size_t)nb * n_ch;
    CHK(reserve(c, &c->io, &c->io_bytes, n_in * (Carver::pad(bsig) + Carver::pad(spec * 16)) + 2 * Carver::pad((size_t)W * 8) +
                                             Carver::pad(bout * 16)));
    Carver cv(c->io);
    X64Tables t;
    CHK(x64_tables(c, cv, window, W, &t));
    double* dx = cv.take<double>((size_t)n_samples * n_ch);
    double2* xs = cv.take<double2>(spec);
    CHK(x64_frames(c, t, x, dx, n_ch, n_samples, W, hop, n_frames, detrend, xs));
    double2* ys = nullptr;
    if (y) {
        double* dy = cv.take<double>((size_t)n_samples * n_ch);
        ys = cv.take<double2>(spec);
        CHK(x64_frames(c, t, y, dy, n_ch, n_samples, W, hop, n_frames, detrend, ys));
    }
    double2* dout = cv.take<double2>(bout);
    if (average == DS_AVG_MEDIAN) {
        const int nbias = (n_frames & 1) ? n_frames : n_frames - 1;
        w64::SpecArgs sa{xs, ys, n_ch, n_frames, FinishPar{norm_scale * (double)std::max(1, nbias), factor, halve_edges, amp_sqrt, nb}, dout};
        CHK(launch(c, "welch_f64_spec_median", w64::k_spec_median, dim3(nb, n_ch), 256, sizeof(double) * (2 * (size_t)n_frames + 4), sa));
    } else {
        w64::SpecArgs sa{xs, ys, n_ch, n_frames, FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb}, dout};
        CHK(launch(c, "welch_f64_spec", w64::k_spec, dim3((nb + 255) / 256, n_ch), 256, 0, sa));
    }
    HIPCHK(c, hipMemcpyAsync(out, dout, bout * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DS_OK;
}

// _csm_welch in float64 end to end (up to 1024 channels; median averaging: up to 128 frames): csm [nb][n_ch][n_ch] complex128
extern "C" int ds_csm_x64(ds_ctx* c, const double* x, int n_ch, int64_t n_samples, int W, int hop, int n_frames,
                          const double* window, int detrend, int average, int amp_sqrt, double norm_scale, double factor,
                          int halve_edges, double* csm) {
    if (!c || !x || !window || !csm) return fail(c, DS_ERR_ARG, "ds_csm_x64: null argument");
    CHK(x64_shape_ok(c, "ds_csm_x64", n_ch, n_samples, W, hop, n_frames, average));
    if (n_ch > w64::CSM_MAX_CH) return fail(c, DS_ERR_UNSUP, "ds_csm_x64: more than 1024 channels (use ds_csm)");
    if (average == DS_AVG_MEDIAN && n_frames > w64::CSM_MEDIAN_MAX_FRAMES)
        return fail(c, DS_ERR_UNSUP, "ds_csm_x64: median averaging over more than 128 frames (use ds_csm)");
    const int nb = W / 2 + 1;
    const size_t spec = (size_t)n_ch * n_frames * nb, bout = (size_t)nb * n_ch * n_ch;
    if (spec * sizeof(double2) > ((size_t)2 << 30))
        return fail(c, DS_ERR_UNSUP, "ds_csm_x64: problem too large for the float64 route (use ds_csm)");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(reserve(c, &c->io, &c->io_bytes, Carver::pad((size_t)n_samples * n_ch * 8) + Carver::pad(spec * 16) +
                                             2 * Carver::pad((size_t)W * 8) + Carver::pad(bout * 16)));
    Carver cv(c->io);
    X64Tables t;
    CHK(x64_tables(c, cv, window, W, &t));
    double* dx = cv.take<double>((size_t)n_samples * n_ch);
    double2* xs = cv.take<double2>(spec);
    CHK(x64_frames(c, t, x, dx, n_ch, n_samples, W, hop, n_frames, detrend, xs));
    double2* dcsm = cv.take<double2>(bout);
    if (average == DS_AVG_MEDIAN) {
        const int nbias = (n_frames & 1) ? n_frames : n_frames - 1;  // as ds_welch_spec_x64
        w64::CsmArgs ca{xs, n_ch, n_frames, FinishPar{norm_scale * (double)std::max(1, nbias), factor, halve_edges, amp_sqrt, nb}, dcsm};
        CHK(launch(c, "csm_f64_median", w64::k_csm_median, dim3(nb, w64::csm_median_tile_pairs(n_ch)), 256,
                   w64::csm_median_lds(n_ch, n_frames), ca));
    } else {
        const int tile = std::max(1, std::min(n_frames, 4096 / n_ch));  // <= 64 KB of frame values per workgroup
        w64::CsmArgs ca{xs, n_ch, n_frames, FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb}, dcsm};
        hipLaunchKernelGGL(w64::k_csm, dim3(nb, w64::csm_pair_groups(n_ch)), dim3(256), (size_t)n_ch * tile * 16, c->stream, ca, tile);
        HIPCHK(c, hipGetLastError());
        c->routes.insert("csm_f64");
    }
    HIPCHK(c, hipMemcpyAsync(csm, dcsm, bout * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DS_OK;
}

extern "C" int ds_welch_psd_dev(ds_ctx* c, const float* x, int n_cx, int64_t ldx, int64_t n_samples,
                                int W, int hop, int n_frames, const float* window, int detrend,
                                int average, int amp_sqrt, double norm_scale, double factor,
                                int halve_edges, float* psd) {
    if (!psd) return fail(c, DS_ERR_ARG, "ds_welch_psd: null output");
    const bool no1k = c && c->cfg.welch_generic;
    if (welch_long_applies(c, W, n_cx, n_samples, n_frames, hop, average))
        return welch_long_run(c, x, n_cx, ldx, nullptr, 0, 0, n_samples, W, hop, n_frames, window, detrend, 0, amp_sqrt,
                              norm_scale, factor, halve_edges, nullptr, psd, 1);
    if (c && W == 16384 && average == DS_AVG_MEAN && !no1k && welch16k::buf_fits(n_samples, n_frames, hop))
        return welch16384_psd_run(c, x, n_cx, ldx, n_samples, hop, n_frames, window, detrend, amp_sqrt,
                                  norm_scale, factor, halve_edges, psd);
    if (c && W == 8192 && average == DS_AVG_MEAN && !no1k && welch8k::buf_fits(n_samples, n_frames, hop))
        return welch8192_psd_run(c, x, n_cx, ldx, n_samples, hop, n_frames, window, detrend, amp_sqrt,
                                 norm_scale, factor, halve_edges, psd);
    if (c && W == 4096 && average == DS_AVG_MEAN && !c->cfg.no_welch4096 && !no1k)
        return welch4096_psd_run(c, x, n_cx, ldx, n_samples, hop, n_frames, window, detrend, amp_sqrt,
                                 norm_scale, factor, halve_edges, psd);
    if (welch2048h_applies(c, W, hop, average, n_samples, n_frames))
        return welch2048h_run(c, x, n_cx, ldx, nullptr, 0, 0, n_samples, n_frames, window, detrend, 0, amp_sqrt, norm_scale,
                              factor, halve_edges, nullptr, psd, 1);
    if (c && (W == 2048 || W == 1024 || W == 512 || W == 256) && average == DS_AVG_MEAN && !no1k &&
        welch1k::buf_fits(n_samples, n_cx, ldx)) {
        auto run = W == 2048 ? welch_wave_psd_run<2048>
                             : (W == 1024 ? welch_wave_psd_run<1024>
                                          : (W == 512 ? welch_wave_psd_run<512> : welch_wave_psd_run<256>));
        return run(c, x, n_cx, ldx, n_samples, hop, n_frames, window, detrend, amp_sqrt, norm_scale, factor,
                   halve_edges, psd, 1);
    }
    if (c && (W == 128 || W == 64 || W == 32) && average == DS_AVG_MEAN && !no1k && welch1k::buf_fits(n_samples, n_cx, ldx))
        return welch_wave_psd_run<256>(c, x, n_cx, ldx, n_samples, hop, n_frames, window, detrend, amp_sqrt, norm_scale,
                                       factor, halve_edges, psd, 256 / W);
    return welch_common(c, 1, x, n_cx, ldx, nullptr, 0, 0, n_samples, W, hop, n_frames, window, detrend,
                        average, 0, amp_sqrt, norm_scale, factor, halve_edges, nullptr, psd);
}
static int welch_csd_dev(ds_ctx* c, const float* x, const float* y, int n_ch, int64_t ld,
                         int64_t n_samples, int W, int hop, int n_frames, const float* window,
                         int detrend, int average, int amp_sqrt, double norm_scale, double factor,
                         int halve_edges, ds_c32* csd) {
    // csd_i = mean_f conj(X_i) Y_i is the cross sum a transfer function with one input channel per
    // output channel accumulates: the register kernels with the finish of kind 2
    const bool generic = c && c->cfg.welch_generic;
    if (c && x && y && window && csd && average == DS_AVG_MEAN && !generic && n_ch > 0 && n_samples > 0 &&
        n_frames > 0 && hop > 0 && hop <= W && ld >= n_samples) {
        if (W == 4096 && !c->cfg.no_welch4096)
            return welch4096_run(c, x, n_ch, ld, y, n_ch, ld, n_samples, hop, n_frames, window, detrend, DS_TF_H1,
                                 amp_sqrt, norm_scale, factor, halve_edges, (float2*)csd, nullptr, 2);
        if (welch_long_applies(c, W, 2 * n_ch, n_samples, n_frames, hop, average))
            return welch_long_run(c, x, n_ch, ld, y, n_ch, ld, n_samples, W, hop, n_frames, window, detrend, DS_TF_H1, amp_sqrt,
                                  norm_scale, factor, halve_edges, (float2*)csd, nullptr, 2);
        if (W == 16384 && welch16k::buf_fits(n_samples, n_frames, hop))
            return welch16384_run(c, x, n_ch, ld, y, n_ch, ld, n_samples, hop, n_frames, window, detrend, DS_TF_H1,
                                  amp_sqrt, norm_scale, factor, halve_edges, (float2*)csd, nullptr, 2);
        if (W == 8192 && welch8k::buf_fits(n_samples, n_frames, hop))
            return welch8192_run(c, x, n_ch, ld, y, n_ch, ld, n_samples, hop, n_frames, window, detrend, DS_TF_H1,
                                 amp_sqrt, norm_scale, factor, halve_edges, (float2*)csd, nullptr, 2);
        if (welch2048h_applies(c, W, hop, average, n_samples, n_frames))
            return welch2048h_run(c, x, n_ch, ld, y, n_ch, ld, n_samples, n_frames, window, detrend, DS_TF_H1, amp_sqrt,
                                  norm_scale, factor, halve_edges, (float2*)csd, nullptr, 2);
        if ((W == 2048 || W == 1024 || W == 512 || W == 256) && welch1k::buf_fits(n_samples, n_ch, ld)) {
            auto run = W == 2048 ? welch_wave_run<2048>
                                 : (W == 1024 ? welch_wave_run<1024> : (W == 512 ? welch_wave_run<512> : welch_wave_run<256>));
            return run(c, x, n_ch, ld, y, n_ch, ld, n_samples, hop, n_frames, window, detrend, DS_TF_H1, amp_sqrt,
                       norm_scale, factor, halve_edges, (float2*)csd, nullptr, 2, 1);
        }
        if ((W == 128 || W == 64 || W == 32) && welch1k::buf_fits(n_samples, n_ch, ld))
            return welch_wave_run<256>(c, x, n_ch, ld, y, n_ch, ld, n_samples, hop, n_frames, window, detrend, DS_TF_H1,
                                       amp_sqrt, norm_scale, factor, halve_edges, (float2*)csd, nullptr, 2, 256 / W);
    }
    return welch_common(c, 2, x, n_ch, ld, y, n_ch, ld, n_samples, W, hop, n_frames, window, detrend,
                        average, 0, amp_sqrt, norm_scale, factor, halve_edges, (float2*)csd, nullptr);
}

// ---- CSM -------------------------------------------------------------------
// frame f of a chunk that starts at frame f0 is frame f0 + f of the signal: the chunk's transform reads x + f0 hop
static bool pad_ok_for_chunks(int64_t n_samples, int hop, int n_frames) {
    return (int64_t)(n_frames - 1) * hop < n_samples;  // every chunk starts inside the signal
}
static int csm_chunked(ds_ctx* c, const float* x, int n_ch, int64_t ld, int64_t n_samples, int W, int hop, int n_frames,
                       const float* window, int detrend, int amp_sqrt, double norm_scale, double factor, int halve_edges,
                       int K, float2* csm) {
    const int nb = W / 2 + 1;
    const size_t part_elems = (size_t)nb * n_ch * n_ch;
    CHK(reserve(c, &c->ws, &c->ws_bytes, Carver::pad(sizeof(float2) * (size_t)nb * n_frames * n_ch) + Carver::pad(sizeof(float2) * part_elems)));
    Carver cv(c->ws);
    float2* X = cv.take<float2>((size_t)nb * n_frames * n_ch);
    float2* part = cv.take<float2>(part_elems);
    if (c->ev_chunk[0] == nullptr)
        for (auto& e : c->ev_chunk) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t main_stream = c->stream;
    // the side stream starts behind everything already queued on the main one (the previous call's product reads X)
    HIPCHK(c, hipEventRecord(c->ev_fork, main_stream));
    HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
    int rc = DS_OK;
    for (int k = 0; k < K && rc == DS_OK; ++k) {
        const int f0 = (int)((int64_t)k * n_frames / K), f1 = (int)((int64_t)(k + 1) * n_frames / K), fk = f1 - f0;
        float2* Xk = X + (size_t)nb * f0 * n_ch;  // chunk k: [nb][fk][n_ch]
        rc = ds_stft_r2c_dev(c, x + (int64_t)f0 * hop, n_samples - (int64_t)f0 * hop, n_ch, ld, W, hop, W, 0, fk, window, detrend,
                             1.0f, 1.0f, 0, (ds_c32*)Xk);
        if (rc != DS_OK) break;
        HIPCHK(c, hipEventRecord(c->ev_chunk[k], main_stream));
        HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_chunk[k], 0));
        CsmArgs a{Xk, n_ch, fk, FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb}, csm, 0};
        a.part_in = k > 0 ? part : nullptr;
        a.part_out = k + 1 < K ? part : nullptr;
        c->stream = c->side;  // launch() enqueues on the context's stream
        rc = launch(c, "csm_gemm@b3", csmb3::k_csm_gemm64_b3, dim3(nb - 1), 256, 0, a);
        c->stream = main_stream;
    }
    // the main stream goes on behind the last product
    HIPCHK(c, hipEventRecord(c->ev_join, c->side));
    HIPCHK(c, hipStreamWaitEvent(main_stream, c->ev_join, 0));
    return rc;
}

static int csm_dev(ds_ctx* c, const float* x, int n_ch, int64_t ld, int64_t n_samples, int W, int hop,
                   int n_frames, const float* window, int detrend, int average, int amp_sqrt,
                   double norm_scale, double factor, int halve_edges, int bin_start, int bin_count,
                   ds_c32* csm) {
    if (!c || !x || !window || !csm) return fail(c, DS_ERR_ARG, "ds_csm: null argument");
    if (n_ch < 1 || n_samples <= 0 || hop <= 0 || hop > W || n_frames <= 0 || ld < n_samples)
        return fail(c, DS_ERR_ARG, "ds_csm: bad shape");
    if (average != DS_AVG_MEAN && average != DS_AVG_MEDIAN)
        return fail(c, DS_ERR_ARG, "ds_csm: average must be mean (0) or median (1)");
    const bool big = W > kMaxFft && is_pow2(W);
    if (!big) CHK(check_fft_len(c, W, "ds_csm window length"));
    const int nb = W / 2 + 1;
    const bool all_bins = bin_start == 0 && bin_count == nb;
    if (bin_start < 0 || bin_count <= 0 || bin_start + bin_count > nb)
        return fail(c, DS_ERR_ARG, "ds_csm: bad bin range");
    if (average == DS_AVG_MEDIAN && !all_bins)
        return fail(c, DS_ERR_UNSUP, "ds_csm: a bin range with median averaging is not built yet");
    if (average == DS_AVG_MEDIAN) {
        // spectra of every frame [c][F][nb] -> per-pair, per-bin medians
        size_t lds = 0;
        const int bpb = median_bins_per_block(2, n_frames, &lds);
        if (!bpb)
            return fail(c, DS_ERR_UNSUP, "ds_csm: median averaging over more than 19 199 frames is not built yet");
        size_t bytes = Carver::pad(sizeof(float2) * (size_t)n_ch * n_frames * nb);
        WelchPlan pl = plan_welch(n_frames, n_ch);
        bytes += big ? stft_big_ws(n_ch, n_frames, W) : Carver::pad(sizeof(float) * (size_t)pl.n_chunks * n_ch * nb);
        CHK(reserve(c, &c->ws, &c->ws_bytes, bytes));
        Carver cv(c->ws);
        float2* xsp = cv.take<float2>((size_t)n_ch * n_frames * nb);
        if (big) {
            CHK(stft_big(c, cv, x, n_ch, ld, n_samples, W, hop, W, 0, n_frames, window, detrend, 1.0f, 1.0f, 0, 0, xsp));
        } else {
            const float2* tw;
            CHK(get_twiddles(c, W, &tw));
            float* scratch = cv.take<float>((size_t)pl.n_chunks * n_ch * nb);
            XspecArgs ax{x, n_samples, ld, n_ch, W, hop, n_frames, detrend, pl.fpc, window, tw, xsp, scratch};
            DISPATCH_N(W, CHK(launch(c, "welch_xspec", k_xspec<NN>, dim3(pl.n_chunks, n_ch), Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, ax)));
        }
        const int nbias = (n_frames & 1) ? n_frames : n_frames - 1;
        CsmMedianArgs m{xsp, n_ch, n_frames, bpb,
                        FinishPar{norm_scale * (double)std::max(1, nbias), factor, halve_edges, amp_sqrt, nb},
                        (float2*)csm};
        CHK(launch(c, "csm_median", k_csm_median, dim3((nb + bpb - 1) / bpb, n_ch * (n_ch + 1) / 2), 256, lds, m));
        return DS_OK;
    }
    // Round 5: the 64-microphone shape in frame chunks on two streams.  Transform and product are both streams of the
    // spectrogram X (written once, read once: 4.9 x the algorithmic bytes of the step, and each kernel alone reaches
    // 0.4 of the HBM roofline); with the frames cut into chunks the transform of chunk k + 1 (main stream) runs beside
    // the product of chunk k (side stream), the products carrying their raw fp32 sums from chunk to chunk
    // (CsmArgs::part_in / part_out: 8.5 MB per hand-over against 66 MB of spectrogram per chunk).
    {
        const int K = std::min(c->cfg.csm_chunks, n_frames / 64);  // >= 64 frames per chunk
        if (K >= 2 && !big && all_bins && n_ch <= 64 && nb >= 3 && !c->cfg.csm_generic && !c->cfg.csm_f32 &&
            csmb3::fits(n_ch, n_frames) && pad_ok_for_chunks(n_samples, hop, n_frames))
            return csm_chunked(c, x, n_ch, ld, n_samples, W, hop, n_frames, window, detrend, amp_sqrt, norm_scale, factor,
                               halve_edges, K, (float2*)csm);
    }
    // the STFT buffer X[b][f][c] (+ the four-step scratch for long windows) in the workspace
    size_t bytes = Carver::pad(sizeof(float2) * (size_t)nb * n_frames * n_ch);
    if (big) bytes += stft_big_ws(n_ch, n_frames, W);
    CHK(reserve(c, &c->ws, &c->ws_bytes, bytes));
    Carver cv(c->ws);
    float2* X = cv.take<float2>((size_t)nb * n_frames * n_ch);
    if (big)
        CHK(stft_big(c, cv, x, n_ch, ld, n_samples, W, hop, W, 0, n_frames, window, detrend, 1.0f, 1.0f, 0, 1, X));
    else
        CHK(ds_stft_r2c_dev(c, x, n_samples, n_ch, ld, W, hop, W, 0, n_frames, window, detrend, 1.0f, 1.0f,
                            0, (ds_c32*)X));
    const int nt = (n_ch + 31) / 32;
    CsmArgs a{X, n_ch, n_frames,
              FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, nb},
              (float2*)csm, bin_start};
    // up to 64 channels: one workgroup per bin shares the operand loads between the three tile
    // pairs (the spectra of an even-length real transform are purely real at both edge bins,
    // which the kernel relies on)
    const bool no64 = c->cfg.csm_generic;
    // the same product from bf16 triples on the 16 x faster bf16 matrix pipe
    // (kernels_csm_b3.hpp; DSPTOOLBOX_AMD_CSM_F32=1 keeps the fp32 matrix instructions)
    const bool f32_only = c->cfg.csm_f32;
    const bool one_wg_per_bin = all_bins && n_ch <= 64 && n_frames >= 8 && nb >= 3 && !no64;
    if (one_wg_per_bin && !f32_only && csmb3::fits(n_ch, n_frames))
        CHK(launch(c, "csm_gemm@b3", csmb3::k_csm_gemm64_b3, dim3(nb - 1), 256, 0, a));
    else if (one_wg_per_bin)
        CHK(launch(c, "csm_gemm@f32", k_csm_gemm64, dim3(nb - 1), 256, 0, a));
    else if (!all_bins && n_ch <= 64 && n_frames >= 8 && !no64 && !f32_only && csmb3::fits(n_ch, n_frames))
        CHK(launch(c, "csm_gemm@b3_range", csmb3::k_csm_gemm64_b3_range, dim3(bin_count), 256, 0, a));
    else if (n_ch > 64 && n_frames >= 8 && !no64 && !f32_only && csmb3::fits_groups(n_ch, n_frames)) {
        // groups of 64 channels: the diagonal blocks, then the blocks below the diagonal (two workgroups each)
        const int ng = (n_ch + 63) / 64;
        a.n_groups = ng;
        a.n_groups_bins = bin_count;
        CHK(launch(c, "csm_gemm@group_b3", csmb3::k_csm_group_b3, dim3(bin_count, ng), 256, 0, a));
        CHK(launch(c, "csm_gemm_offdiag", csmb3::k_csm_offdiag_b3, dim3(16 * ((bin_count + 7) / 8), ng * (ng - 1) / 2),
                   256, 0, a));
    }
    else
        CHK(launch(c, "csm_gemm@generic", k_csm_gemm, dim3(bin_count, nt * (nt + 1) / 2), 256, 0, a));
    return DS_OK;
}

extern "C" int ds_csm_dev(ds_ctx* c, const float* x, int n_ch, int64_t ld, int64_t n_samples, int W,
                          int hop, int n_frames, const float* window, int detrend, int average,
                          int amp_sqrt, double norm_scale, double factor, int halve_edges, ds_c32* csm) {
    return csm_dev(c, x, n_ch, ld, n_samples, W, hop, n_frames, window, detrend, average, amp_sqrt, norm_scale,
                   factor, halve_edges, 0, W / 2 + 1, csm);
}

// bins [bin_start, bin_start + bin_count) only (csm_dev[0] = matrix of bin_start): the multi-GPU
// split of the CSM -- every rank transforms all channels and keeps its own bin range
extern "C" int ds_csm_bins_dev(ds_ctx* c, const float* x, int n_ch, int64_t ld, int64_t n_samples, int W,
                               int hop, int n_frames, const float* window, int detrend, int amp_sqrt,
                               double norm_scale, double factor, int halve_edges, int bin_start,
                               int bin_count, ds_c32* csm) {
    return csm_dev(c, x, n_ch, ld, n_samples, W, hop, n_frames, window, detrend, DS_AVG_MEAN, amp_sqrt,
                   norm_scale, factor, halve_edges, bin_start, bin_count, csm);
}

extern "C" int ds_csm_spec_dev(ds_ctx* c, const ds_c32* X, int n_bins, int n_frames, int n_ch,
                               int amp_sqrt, double norm_scale, double factor, int halve_edges,
                               ds_c32* csm) {
    if (!c || !X || !csm) return fail(c, DS_ERR_ARG, "ds_csm_spec: null argument");
    if (n_bins <= 0 || n_frames <= 0 || n_ch <= 0) return fail(c, DS_ERR_ARG, "ds_csm_spec: bad shape");
    const int nt = (n_ch + 31) / 32;
    CsmArgs a{(const float2*)X, n_ch, n_frames,
              FinishPar{norm_scale / (double)n_frames, factor, halve_edges, amp_sqrt, n_bins},
              (float2*)csm, 0};
    CHK(launch(c, "csm_gemm@generic", k_csm_gemm, dim3(n_bins, nt * (nt + 1) / 2), 256, 0, a));
    return DS_OK;
}

extern "C" int ds_csm_spec(ds_ctx* c, const ds_c32* X, int n_bins, int n_frames, int n_ch, int amp_sqrt,
                           double norm_scale, double factor, int halve_edges, ds_c32* csm) {
    if (!c || !X || !csm) return fail(c, DS_ERR_ARG, "ds_csm_spec: null argument");
    if (n_bins <= 0 || n_frames <= 0 || n_ch <= 0) return fail(c, DS_ERR_ARG, "ds_csm_spec: bad shape");
    size_t nx = (size_t)n_bins * n_frames * n_ch, no = (size_t)n_bins * n_ch * n_ch;
    CHK(reserve(c, &c->io, &c->io_bytes, Carver::pad(nx * 8) + Carver::pad(no * 8) + 4096));
    Carver cv(c->io);
    float2* dx = cv.take<float2>(nx);
    float2* dc = cv.take<float2>(no);
    CHK(ds_upload(c, dx, X, nx * 8));
    CHK(ds_csm_spec_dev(c, (const ds_c32*)dx, n_bins, n_frames, n_ch, amp_sqrt, norm_scale, factor,
                        halve_edges, (ds_c32*)dc));
    return ds_download(c, csm, dc, no * 8);
}

// ---- delay-and-sum beamformer map ---------------------------------------------------
extern "C" int ds_das_map_dev(ds_ctx* c, const ds_c32* csm, const ds_c32* h, int n_bins, int n_ch,
                              int n_grid, float* map) {
    if (!c || !csm || !h || !map) return fail(c, DS_ERR_ARG, "ds_das_map: null argument");
    if (n_bins <= 0 || n_ch <= 0 || n_grid <= 0) return fail(c, DS_ERR_ARG, "ds_das_map: bad shape");
    if (n_bins > 65535) return fail(c, DS_ERR_UNSUP, "ds_das_map: more than 65535 bins per call is not built yet");
    DasArgs a{(const float2*)csm, (const float2*)h, n_bins, n_ch, n_grid, map};
    CHK(launch(c, "das_map", k_das_map, dim3((n_grid + 127) / 128, n_bins), 256, 0, a));
    return DS_OK;
}

__global__ void k_csm_das_prepare(const float2* in, int64_t total, int n_ch, float scale, int zero_diag, float2* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t e = i % ((int64_t)n_ch * n_ch);
    const bool diag = zero_diag && (e / n_ch == e % n_ch);
    const float2 v = in[i];
    out[i] = diag ? make_float2(0.f, 0.f) : make_float2(v.x * scale, v.y * scale);
}
extern "C" int ds_csm_das_prepare_dev(ds_ctx* c, const ds_c32* csm, int n_bins, int n_ch, double scale,
                                      int zero_diagonal, ds_c32* out) {
    if (!c || !csm || !out) return fail(c, DS_ERR_ARG, "ds_csm_das_prepare: null argument");
    if (n_bins <= 0 || n_ch <= 0) return fail(c, DS_ERR_ARG, "ds_csm_das_prepare: bad shape");
    const int64_t total = (int64_t)n_bins * n_ch * n_ch;
    hipLaunchKernelGGL(k_csm_das_prepare, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, (const float2*)csm,
                       total, n_ch, (float)scale, zero_diagonal, (float2*)out);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}
extern "C" int ds_das_map(ds_ctx* c, const ds_c32* csm, const ds_c32* h, int n_bins, int n_ch, int n_grid,
                          float* map) {
    if (!c || !csm || !h || !map) return fail(c, DS_ERR_ARG, "ds_das_map: null argument");
    if (n_bins <= 0 || n_ch <= 0 || n_grid <= 0) return fail(c, DS_ERR_ARG, "ds_das_map: bad shape");
    size_t nc = (size_t)n_bins * n_ch * n_ch, nh = (size_t)n_bins * n_ch * n_grid, nm = (size_t)n_grid * n_bins;
    CHK(reserve(c, &c->io, &c->io_bytes, Carver::pad(nc * 8) + Carver::pad(nh * 8) + Carver::pad(nm * 4) + 4096));
    Carver cv(c->io);
    float2* dc = cv.take<float2>(nc);
    float2* dh = cv.take<float2>(nh);
    float* dm = cv.take<float>(nm);
    CHK(ds_upload(c, dc, csm, nc * 8));
    CHK(ds_upload(c, dh, h, nh * 8));
    CHK(ds_das_map_dev(c, (const ds_c32*)dc, (const ds_c32*)dh, n_bins, n_ch, n_grid, dm));
    return ds_download(c, map, dm, nm * 4);
}

// ---- four-step FFT for 2^15 .. 2^24 points (kernels_bigfft.hpp) -------------------
static int big_rows_ct(int n2) {
    int nt = dsfft::threads_for(n2);
    size_t per = (size_t)(((n2 + n2 / 16 + 2 + 30) / 32) * 32 + 1) * sizeof(float2);
    int ct = std::min<int>({16, 1024 / nt, std::max<int>(1, (int)((70 * 1024) / per))});
    while (ct & (ct - 1)) ct &= ct - 1;  // power of two: must divide N1
    return ct;
}

// cols stage: source = complex `zin` (may equal `zout`) or real channel pairs
static int big_cols(ds_ctx* c, const float2* zin, const float* xreal, int n_ch, int64_t ld_real,
                    int64_t n_samples, float2* zout, int64_t N, int batch) {
    constexpr int N1 = 1024;
    const int n2 = (int)(N / N1);
    const float2* tw;
    CHK(get_twiddles(c, N1, &tw));
    const int ct = std::min(8, n2);
    dsbig::ColsArgs a{zin, xreal, nullptr, n_samples, 0, zout, N, n2, ct, 0, ld_real, n_ch, tw};
    size_t lds = (size_t)ct * dsbig::ch_stride<N1>() * sizeof(float2);
    CHK(launch(c, "bigfft_cols", dsbig::k_big_cols<N1>, dim3(n2 / ct, batch), ct * Cfg<N1>::NT, lds, a));
    return DS_OK;
}

static int big_rows(ds_ctx* c, const float2* zin, float2* zout, int64_t N, int batch) {
    constexpr int N1 = 1024;
    const int n2 = (int)(N / N1);
    const float2* tw;
    CHK(get_twiddles(c, n2, &tw));
    const int ct = big_rows_ct(n2);
    dsbig::RowsArgs a{zin, zout, N, N1, ct, tw};
    size_t lds = 0;
    DISPATCH_N(n2, lds = (size_t)ct * dsbig::ch_stride<NN>() * sizeof(float2));
    DISPATCH_N(n2, CHK(launch(c, "bigfft_rows", dsbig::k_big_rows<NN>, dim3(N1 / ct, batch), ct * Cfg<NN>::NT, lds, a)));
    return DS_OK;
}

static int check_big_len(ds_ctx* c, int64_t n, const char* what) {
    if (!is_pow2(n)) return fail(c, DS_ERR_ARG, std::string(what) + ": internal: not a power of two");
    if (n > kMaxBigFft) return fail(c, DS_ERR_UNSUP, std::string(what) + ": lengths above 2^24 are not built yet");
    return DS_OK;
}

static int rfft_big(ds_ctx* c, const float* x, int n_ch, int64_t ld, int64_t n_samples, int64_t N,
                    float scale, float2* spec) {
    const int npair = (n_ch + 1) / 2;
    CHK(reserve(c, &c->ws, &c->ws_bytes, 2 * Carver::pad(sizeof(float2) * (size_t)npair * N)));
    Carver cv(c->ws);
    float2* P = cv.take<float2>((size_t)npair * N);
    float2* Q = cv.take<float2>((size_t)npair * N);
    CHK(big_cols(c, nullptr, x, n_ch, ld, n_samples, P, N, npair));
    CHK(big_rows(c, P, Q, N, npair));
    dsbig::UnpackArgs u{Q, N, n_ch, scale, spec, n_ch, 1};
    CHK(launch(c, "bigfft_unpack", dsbig::k_big_unpack, dim3(1024, npair), 256, 0, u));
    return DS_OK;
}

static int deconv_big(ds_ctx* c, const float* y, int n_items, int n_ch, int64_t ld, int64_t n_samples,
                      int64_t N, const float2* r, int r_per_channel, int64_t n_out, int64_t ld_out,
                      float* ir) {
    const int npair = (n_ch + 1) / 2, batch = n_items * npair;
    CHK(reserve(c, &c->ws, &c->ws_bytes, 2 * Carver::pad(sizeof(float2) * (size_t)batch * N)));
    Carver cv(c->ws);
    float2* P = cv.take<float2>((size_t)batch * N);
    float2* Q = cv.take<float2>((size_t)batch * N);
    CHK(big_cols(c, nullptr, y, n_ch, ld, n_samples, P, N, batch));
    CHK(big_rows(c, P, Q, N, batch));
    dsbig::MulArgs m{Q, N, n_ch, r_per_channel, r};
    CHK(launch(c, "bigfft_mul", dsbig::k_big_mul, dim3(1024, batch), 256, 0, m));
    CHK(big_cols(c, Q, nullptr, n_ch, 0, 0, Q, N, batch));
    CHK(big_rows(c, Q, P, N, batch));
    dsbig::StoreArgs st{P, N, n_out, ld_out, n_ch, ir};
    CHK(launch(c, "bigfft_store", dsbig::k_big_store, dim3(1024, batch), 256, 0, st));
    return DS_OK;
}

// ---- framed transforms beyond the LDS-resident FFT (window / FFT length 2^15 .. 2^24) ------
// Frame pairs of every channel are one batch of four-step complex FFTs, processed in groups
// of <= 2^25 complex points per scratch buffer.
static int64_t stft_big_group(int64_t nfft, int64_t batch) {
    return std::max<int64_t>(1, std::min<int64_t>({batch, ((int64_t)1 << 25) / nfft, (int64_t)32768}));
}
static size_t stft_big_ws(int n_ch, int n_frames, int64_t nfft) {
    const int64_t batch = (int64_t)n_ch * ((n_frames + 1) / 2);
    const int64_t g = stft_big_group(nfft, batch);
    return 2 * Carver::pad(sizeof(float2) * (size_t)g * nfft) + Carver::pad(sizeof(float) * (size_t)n_ch * n_frames);
}
// layout 0: out[(c*F + f)*nb + k] (unscaled spectra for the Welch sums), 1: out[(k*F + f)*C + c]
static int stft_big(ds_ctx* c, Carver& cv, const float* x, int n_ch, int64_t ld, int64_t n_samples,
                    int W, int hop, int64_t nfft, int64_t pad_front, int n_frames, const float* window,
                    int detrend, float scale, float edge_scale, int power, int layout, float2* out) {
    CHK(check_big_len(c, nfft, "framed transform length"));
    constexpr int N1 = 1024;
    const int n2 = (int)(nfft / N1);
    const int64_t batch = (int64_t)n_ch * ((n_frames + 1) / 2);
    const int64_t grp = stft_big_group(nfft, batch);
    float2* P = cv.take<float2>((size_t)grp * nfft);
    float2* Q = cv.take<float2>((size_t)grp * nfft);
    float* means = cv.take<float>((size_t)n_ch * n_frames);
    if (detrend) {
        dsbig::FrameMeansArgs m{x, n_samples, ld, pad_front, n_ch, W, hop, n_frames, window, means};
        CHK(launch(c, "bigfft_means", dsbig::k_frame_means, dim3(n_frames, n_ch), 256, 0, m));
    }
    const float2* tw;
    CHK(get_twiddles(c, N1, &tw));
    const int ct = std::min(8, n2);
    const size_t lds = (size_t)ct * dsbig::ch_stride<N1>() * sizeof(float2);
    for (int64_t b0 = 0; b0 < batch; b0 += grp) {
        const int nb = (int)std::min<int64_t>(grp, batch - b0);
        dsbig::ColsArgs a{nullptr, x, nullptr, n_samples, 0, P, nfft, n2, ct, 0, ld, n_ch, tw,
                          window, detrend ? means : nullptr, W, hop, n_frames, pad_front, b0};
        CHK(launch(c, "bigfft_cols", dsbig::k_big_cols<N1>, dim3(n2 / ct, nb), ct * Cfg<N1>::NT, lds, a));
        CHK(big_rows(c, P, Q, nfft, nb));
        dsbig::UnpackFramesArgs u{Q, nfft, b0, n_ch, n_frames, layout, power, scale, edge_scale, out};
        CHK(launch(c, "bigfft_unpack", dsbig::k_big_unpack_frames, dim3(64, nb), 256, 0, u));
    }
    return DS_OK;
}

// Welch for window lengths beyond the LDS-resident FFT: spectra of every frame -> frame sums
// (or per-bin medians) -> the usual finish
static int welch_big(ds_ctx* c, int kind, const float* x, int n_cx, int64_t ldx, const float* y,
                     int n_cy, int64_t ldy, int64_t n_samples, int W, int hop, int n_frames,
                     const float* window, int detrend, int average, int mode, int amp_sqrt,
                     double norm_scale, double factor, int halve_edges, float2* out_c, float* out_r) {
    const int nb = W / 2 + 1;
    const int nyc = kind == 1 ? 0 : n_cy;
    const int nmax = std::max(n_cx, nyc);
    size_t med_lds = 0;
    const int med_bpb = median_bins_per_block(3, n_frames, &med_lds);
    if (average == DS_AVG_MEDIAN && !med_bpb)
        return fail(c, DS_ERR_UNSUP, "welch: median averaging over more than 12 799 frames is not built yet");
    size_t bytes = stft_big_ws(nmax, n_frames, W) + Carver::pad(sizeof(float2) * (size_t)n_cx * n_frames * nb) +
                   Carver::pad(sizeof(float2) * (size_t)std::max(1, nyc) * n_frames * nb) +
                   Carver::pad(sizeof(float) * (size_t)n_cx * nb) +
                   Carver::pad(sizeof(float2) * (size_t)std::max(1, nyc) * nb) +
                   Carver::pad(sizeof(float) * (size_t)std::max(1, nyc) * nb);
    CHK(reserve(c, &c->ws, &c->ws_bytes, bytes));
    Carver cv(c->ws);
    float2* xsp = cv.take<float2>((size_t)n_cx * n_frames * nb);
    float2* ysp = cv.take<float2>((size_t)std::max(1, nyc) * n_frames * nb);
    float* pxx = cv.take<float>((size_t)n_cx * nb);
    float2* pxy = cv.take<float2>((size_t)std::max(1, nyc) * nb);
    float* pyy = cv.take<float>((size_t)std::max(1, nyc) * nb);
    Carver scratch = cv;  // the FFT scratch is reused by both signals
    CHK(stft_big(c, scratch, x, n_cx, ldx, n_samples, W, hop, W, 0, n_frames, window, detrend, 1.0f, 1.0f, 0,
                 0, xsp));
    if (nyc) {
        scratch = cv;
        CHK(stft_big(c, scratch, y, nyc, ldy, n_samples, W, hop, W, 0, n_frames, window, detrend, 1.0f, 1.0f,
                     0, 0, ysp));
    }
    double count = (double)n_frames;
    if (average == DS_AVG_MEDIAN) {
        MedianArgs m{xsp, nyc ? ysp : nullptr, n_cx, nyc, n_frames, nb, kind, med_bpb, pxx, pxy, pyy};
        CHK(launch(c, "welch_median", k_welch_median, dim3((nb + med_bpb - 1) / med_bpb, kind == 1 ? n_cx : n_cy), 256,
                   med_lds, m));
        const int nbias = (n_frames & 1) ? n_frames : n_frames - 1;
        count = 1.0 / (double)std::max(1, nbias);
    } else {
        dsbig::SpecSumArgs sa{xsp, nyc ? ysp : nullptr, n_cx, nyc, n_frames, nb, kind, pxx, pxy, pyy};
        CHK(launch(c, "welch_specsum", dsbig::k_spec_sum, dim3((nb + 255) / 256, kind == 1 ? n_cx : n_cy), 256, 0, sa));
    }
    WelchFinArgs f{pxx, pxy, pyy, 1, 1, n_cx, n_cy, kind, mode,
                   FinishPar{norm_scale / count, factor, halve_edges, amp_sqrt, nb}, out_c, out_r};
    int64_t total = (int64_t)nb * (kind == 1 ? n_cx : n_cy);
    CHK(launch_finish(c, dim3((unsigned)((total + 63) / 64)), f));
    return DS_OK;
}

// ---- arbitrary lengths: Bluestein on top of the four-step FFT ------------------------
static int64_t blue_len(int64_t L) {
    int64_t m = (int64_t)1 << 15;  // smallest four-step length
    while (m < 2 * L - 1) m <<= 1;
    return m;
}

static int blue_filter(ds_ctx* c, int64_t L, int64_t M, const float2** out) {
    auto key = std::make_pair(L, M);
    auto it = c->blue.find(key);
    if (it != c->blue.end()) {
        it->second.stamp = ++c->blue_clock;
        *out = it->second.ptr;
        return DS_OK;
    }
    const size_t bytes = sizeof(float2) * (size_t)M;
    // make room: drop the least recently used tables (hipFree waits for the device, so a table
    // still referenced by queued kernels of an earlier call is never pulled from under them)
    while (!c->blue.empty() && c->blue_bytes + bytes > c->cfg.bluestein_cache_bytes) {
        auto lru = c->blue.begin();
        for (auto jt = c->blue.begin(); jt != c->blue.end(); ++jt)
            if (jt->second.stamp < lru->second.stamp) lru = jt;
        HIPCHK(c, hipFree(lru->second.ptr));
        c->blue_bytes -= lru->second.bytes;
        c->blue.erase(lru);
    }
    float2 *bt = nullptr, *bf = nullptr;
    HIPCHK(c, hipMalloc((void**)&bt, bytes));
    if (hipMalloc((void**)&bf, bytes) != hipSuccess) {
        (void)hipFree(bt);
        return fail(c, DS_ERR_NOMEM, "Bluestein filter table: hipMalloc failed");
    }
    int rc = DS_OK;
    do {
        hipLaunchKernelGGL(dsblue::k_filter, dim3(1024), dim3(256), 0, c->stream, bt, L, M);
        if (hipGetLastError() != hipSuccess) { rc = fail(c, DS_ERR_HIP, "Bluestein filter kernel launch failed"); break; }
        if ((rc = big_cols(c, bt, nullptr, 0, 0, 0, bt, M, 1)) != DS_OK) break;
        if ((rc = big_rows(c, bt, bf, M, 1)) != DS_OK) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(c, DS_ERR_HIP, "Bluestein filter: stream sync failed"); break; }
    } while (0);
    (void)hipFree(bt);
    if (rc != DS_OK) {
        (void)hipFree(bf);
        return rc;
    }
    c->blue[key] = ds_ctx::BlueEntry{bf, bytes, ++c->blue_clock};
    c->blue_bytes += bytes;
    *out = bf;
    return DS_OK;
}

// X[batch][L] = DFT_L of (real channel pairs | complex zin[batch][L]); P, Q: [batch][M] scratch
static int blue_dft(ds_ctx* c, const float* xreal, int n_ch, int64_t ld_real, int64_t n_samples,
                    const float2* zin, int batch, int64_t L, int64_t M, float2* P, float2* Q, float2* X) {
    const float2* bf;
    CHK(blue_filter(c, L, M, &bf));
    dsblue::PreArgs pa{xreal, ld_real, n_samples, n_ch, zin, P, L, M};
    CHK(launch(c, "blue_pre", dsblue::k_pre, dim3(512, batch), 256, 0, pa));
    CHK(big_cols(c, P, nullptr, 0, 0, 0, P, M, batch));
    CHK(big_rows(c, P, Q, M, batch));
    hipLaunchKernelGGL(dsblue::k_mul_filter, dim3(512, batch), dim3(256), 0, c->stream, Q, bf, M);
    HIPCHK(c, hipGetLastError());
    CHK(big_cols(c, Q, nullptr, 0, 0, 0, Q, M, batch));
    CHK(big_rows(c, Q, P, M, batch));
    hipLaunchKernelGGL(dsblue::k_post, dim3(512, batch), dim3(256), 0, c->stream, (const float2*)P, X, L, M);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}

static int check_blue_len(ds_ctx* c, int64_t L, const char* what) {
    if (L < 2) return fail(c, DS_ERR_ARG, std::string(what) + ": length must be >= 2");
    if (2 * L - 1 > kMaxBigFft) return fail(c, DS_ERR_UNSUP, std::string(what) + ": non-power-of-two lengths above 2^23 are not built yet");
    return DS_OK;
}

static int rfft_blue(ds_ctx* c, const float* x, int n_ch, int64_t ld, int64_t n_samples, int64_t L,
                     float scale, float2* spec) {
    const int npair = (n_ch + 1) / 2;
    const int64_t M = blue_len(L);
    CHK(reserve(c, &c->ws, &c->ws_bytes, 2 * Carver::pad(sizeof(float2) * (size_t)npair * M) +
                                             Carver::pad(sizeof(float2) * (size_t)npair * L)));
    Carver cv(c->ws);
    float2* P = cv.take<float2>((size_t)npair * M);
    float2* Q = cv.take<float2>((size_t)npair * M);
    float2* X = cv.take<float2>((size_t)npair * L);
    CHK(blue_dft(c, x, n_ch, ld, n_samples, nullptr, npair, L, M, P, Q, X));
    hipLaunchKernelGGL(dsblue::k_unpack, dim3(512, npair), dim3(256), 0, c->stream, (const float2*)X, L, n_ch,
                       scale, spec);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}

static int deconv_blue(ds_ctx* c, const float* y, int n_items, int n_ch, int64_t ld, int64_t n_samples,
                       int64_t L, const float2* r, int r_per_channel, int64_t n_out, int64_t ld_out,
                       float* ir) {
    const int npair = (n_ch + 1) / 2, batch = n_items * npair;
    const int64_t M = blue_len(L);
    CHK(reserve(c, &c->ws, &c->ws_bytes, 2 * Carver::pad(sizeof(float2) * (size_t)batch * M) +
                                             2 * Carver::pad(sizeof(float2) * (size_t)batch * L)));
    Carver cv(c->ws);
    float2* P = cv.take<float2>((size_t)batch * M);
    float2* Q = cv.take<float2>((size_t)batch * M);
    float2* X = cv.take<float2>((size_t)batch * L);
    float2* Y = cv.take<float2>((size_t)batch * L);
    CHK(blue_dft(c, y, n_ch, ld, n_samples, nullptr, batch, L, M, P, Q, X));
    hipLaunchKernelGGL(dsblue::k_mul_r, dim3(512, batch), dim3(256), 0, c->stream, X, L, n_ch, r_per_channel, r);
    HIPCHK(c, hipGetLastError());
    CHK(blue_dft(c, nullptr, n_ch, 0, 0, X, batch, L, M, P, Q, Y));
    hipLaunchKernelGGL(dsblue::k_store, dim3(512, batch), dim3(256), 0, c->stream, (const float2*)Y, L, n_out,
                       ld_out, n_ch, ir);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}

// ---- whole-signal rFFT, deconvolution ---------------------------------------
extern "C" int ds_rfft_dev(ds_ctx* c, const float* x, int n_ch, int64_t ld, int64_t n_samples,
                           int n_fft, float scale, ds_c32* spec) {
    if (!c || !x || !spec) return fail(c, DS_ERR_ARG, "ds_rfft: null argument");
    if (n_ch <= 0 || n_samples <= 0 || ld < n_samples || n_samples > n_fft)
        return fail(c, DS_ERR_ARG, "ds_rfft: bad shape (n_samples must be <= n_fft)");
    if (!is_pow2(n_fft)) {
        CHK(check_blue_len(c, n_fft, "ds_rfft n_fft"));
        return rfft_blue(c, x, n_ch, ld, n_samples, n_fft, scale, (float2*)spec);
    }
    if (n_fft > kMaxFft) {
        CHK(check_big_len(c, n_fft, "ds_rfft n_fft"));
        return rfft_big(c, x, n_ch, ld, n_samples, n_fft, scale, (float2*)spec);
    }
    CHK(check_fft_len(c, n_fft, "ds_rfft n_fft"));
    const float2* tw;
    CHK(get_twiddles(c, n_fft, &tw));
    RfftArgs a{x, n_samples, ld, n_ch, tw, scale, (float2*)spec};
    DISPATCH_N(n_fft, CHK(launch(c, "rfft", k_rfft<NN>, dim3((n_ch + 1) / 2), Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, a)));
    return DS_OK;
}

extern "C" int ds_deconv_inverse_dev(ds_ctx* c, const ds_c32* xspec, int n_ch, int n_bins,
                                     const float* eps, ds_c32* r) {
    if (!c || !xspec || !r || n_ch <= 0 || n_bins <= 0)
        return fail(c, DS_ERR_ARG, "ds_deconv_inverse: bad argument");
    int64_t total = (int64_t)n_ch * n_bins;
    hipLaunchKernelGGL(k_deconv_inverse, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       (const float2*)xspec, n_ch, n_bins, eps, (float2*)r);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}

extern "C" int ds_deconv_dev(ds_ctx* c, const float* y, int n_items, int n_ch, int64_t ld,
                             int64_t n_samples, int n_fft, const ds_c32* r, int r_per_channel,
                             int64_t n_out, int64_t ld_out, float* ir) {
    if (!c || !y || !r || !ir) return fail(c, DS_ERR_ARG, "ds_deconv: null argument");
    if (n_items <= 0 || n_ch <= 0 || n_samples <= 0 || ld < n_samples || n_samples > n_fft ||
        n_out <= 0 || n_out > n_fft || ld_out < n_out)
        return fail(c, DS_ERR_ARG, "ds_deconv: bad shape");
    if (!is_pow2(n_fft)) {
        CHK(check_blue_len(c, n_fft, "ds_deconv n_fft"));
        return deconv_blue(c, y, n_items, n_ch, ld, n_samples, n_fft, (const float2*)r, r_per_channel, n_out,
                           ld_out, ir);
    }
    if (n_fft > kMaxFft) {
        CHK(check_big_len(c, n_fft, "ds_deconv n_fft"));
        return deconv_big(c, y, n_items, n_ch, ld, n_samples, n_fft, (const float2*)r, r_per_channel,
                          n_out, ld_out, ir);
    }
    CHK(check_fft_len(c, n_fft, "ds_deconv n_fft"));
    const bool no8k = c->cfg.deconv_generic;
    if (n_fft == deconv8k::N && !r_per_channel && !no8k) {
        // 8192 points, one inverse spectrum for all channels: two register-resident 4096-point
        // transforms per channel pair, the packed spectrum multiplied directly (kernels_deconv8k.hpp)
        if (!c->w4_tables) {
            std::vector<float2> h;
            welch4096::host_tables(h);
            CHK(upload_table_fwd(c, &c->w4_tables, h));
        }
        if (!c->deconv8k_tables) {
            std::vector<float2> h;
            deconv8k::host_tables(h);
            CHK(upload_table_fwd(c, &c->deconv8k_tables, h));
        }
        deconv8k::Args a8{y, n_samples, ld, n_out, ld_out, n_ch, c->w4_tables, c->deconv8k_tables,
                          (const float2*)r, ir};
        // one 256-thread group per channel pair, its two sub-spectra one after the other: three independent
        // workgroups per CU (k_deconv3); DSPTOOLBOX_AMD_DECONV_2PERCU=1 keeps the 512-thread kernel (A/B)
        // default since round 5: two persistent workgroups per CU, the next unit's samples in flight during this unit's
        // transforms, the inverse spectrum from a permuted copy (k_rperm + k_deconv_p); DSPTOOLBOX_AMD_DECONV_PERSIST=0
        // keeps the one-unit-per-workgroup kernels below
        const int64_t n_units = (int64_t)((n_ch + 1) / 2) * n_items;
        // (n_units + grid stays an int inside the kernel: u + gridDim.x is formed for the prefetch past the last unit)
        if (c->cfg.deconv_persist && !c->cfg.deconv_2percu && n_units < ((int64_t)1 << 31) - 4096 && c->n_cu > 0) {
            if (!c->deconv_rperm) HIPCHK(c, hipMalloc((void**)&c->deconv_rperm, sizeof(float2) * deconv8k::RPERM_LEN));
            CHK(launch(c, "deconv_rperm", deconv8k::k_rperm, dim3(32), 256, 0, deconv8k::RpArgs{(const float2*)r, c->deconv_rperm}));
            deconv8k::PArgs pa{a8, c->deconv_rperm, (int)n_units};
            const int grid = (int)std::min<int64_t>(n_units, 2 * (int64_t)c->n_cu);
            return launch(c, "deconv@8k_persist", deconv8k::k_deconv_p, dim3((unsigned)grid), 256, deconv8k::LDS_BYTES_3, pa);
        }
        const bool two = c->cfg.deconv_2percu;
        // four workgroups per CU (k_deconv3q: all 1024 pairs of the benchmark resident at once) unless
        // DSPTOOLBOX_AMD_DECONV_4PERCU=0 (k_deconv3: three, 168 registers)
        const bool four = c->cfg.deconv_4percu;
        if (!two && four && (int64_t)((n_ch + 1) / 2) * n_items < ((int64_t)1 << 31))
            return launch(c, "deconv@8k_4percu", deconv8k::k_deconv3q, dim3((unsigned)(((n_ch + 1) / 2) * n_items)), 256,
                          deconv8k::LDS_BYTES_3, a8);
        if (!two && (int64_t)((n_ch + 1) / 2) * n_items < ((int64_t)1 << 31))
            return launch(c, "deconv@8k_3percu", deconv8k::k_deconv3, dim3((unsigned)(((n_ch + 1) / 2) * n_items)), 256,
                          deconv8k::LDS_BYTES_3, a8);
        CHK(launch(c, "deconv@8k_512", deconv8k::k_deconv, dim3((n_ch + 1) / 2, n_items), deconv8k::NTB,
                   deconv8k::LDS_BYTES, a8));
        return DS_OK;
    }
    const float2* tw;
    CHK(get_twiddles(c, n_fft, &tw));
    DeconvArgs a{y, n_samples, ld, n_out, ld_out, n_ch, r_per_channel, tw, (const float2*)r, ir};
    dim3 grid((n_ch + 1) / 2, n_items);
    DISPATCH_N(n_fft, CHK(launch(c, "deconv@generic", k_deconv<NN>, grid, Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, a)));
    return DS_OK;
}

// ---- FIR ---------------------------------------------------------------------
static int upload_table_fwd(ds_ctx* c, float2** slot, const std::vector<float2>& h) {
    if (*slot) return DS_OK;
    HIPCHK(c, hipMalloc((void**)slot, sizeof(float2) * h.size()));
    HIPCHK(c, hipMemcpyAsync(*slot, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DS_OK;
}

static int fir_block_len(int n_taps, int v = 0) {  // v: forced block length (DSPTOOLBOX_AMD_FIR_BLOCK), 0 = none
    int n = 1024;
    while (n < 4 * n_taps && n < kMaxFft) n <<= 1;
    // 1025 .. 2048 taps would take generic 8192-point blocks (75 % of every block new samples): the
    // register kernel for 16384-point blocks with its whole-group stores is faster (1025 taps:
    // 2.06 -> 1.57 ms on the 32-band x 8 x 2^22 shape) when the discarded length is a multiple of 4
    if (n == 8192 && ((n_taps - 1) & 3) == 0) n = 16384;
    if (v >= 1024 && v <= kMaxFft && is_pow2(v) && n_taps - 1 <= v / 2) n = v;  // DSPTOOLBOX_AMD_FIR_BLOCK
    return n;
}

// Long filters (more taps than half the largest LDS-resident block): overlap-save on the
// four-step FFT with blocks of 2^15 .. 2^24 points.  Per block: one forward transform of the
// channel pairs, then groups of filters: multiply by the tap spectra, inverse, store.
static int fir_long(ds_ctx* c, const float* x, int n_ch, int64_t ldx, int64_t n_samples,
                    const float* taps, int n_filt, int n_taps, float* y, int64_t ld_y) {
    if ((int64_t)n_taps - 1 > kMaxBigFft / 2)
        return fail(c, DS_ERR_UNSUP, "ds_fir_ola: more than 2^23 + 1 taps is not built yet");
    int64_t L = (int64_t)1 << 15;
    while (L < 4 * (int64_t)n_taps && L < kMaxBigFft) L <<= 1;
    while (L > ((int64_t)1 << 15) && L / 2 >= n_samples + n_taps - 1) L >>= 1;  // short signals: one block
    const int64_t nb = L / 2 + 1, step = L - (n_taps - 1);
    const int npair = (n_ch + 1) / 2;
    const int G = (int)std::max<int64_t>(1, std::min<int64_t>(n_filt, ((int64_t)1 << 25) / (npair * L)));
    const size_t scratch = (size_t)G * npair * L;
    CHK(reserve(c, &c->ws, &c->ws_bytes,
                Carver::pad(sizeof(float2) * (size_t)n_filt * nb) + Carver::pad(sizeof(float2) * (size_t)npair * L) +
                    2 * Carver::pad(sizeof(float2) * scratch)));
    Carver cv(c->ws);
    float2* R = cv.take<float2>((size_t)n_filt * nb);
    float2* Qs = cv.take<float2>((size_t)npair * L);
    float2* P = cv.take<float2>(scratch);
    float2* S = cv.take<float2>(scratch);
    constexpr int N1 = 1024;
    const int n2 = (int)(L / N1);
    const float2* tw;
    CHK(get_twiddles(c, N1, &tw));
    const int ct = std::min(8, n2);
    const size_t lds = (size_t)ct * dsbig::ch_stride<N1>() * sizeof(float2);
    // tap spectra R[k][nb], 2*G*npair filters per pass through the scratch buffers
    for (int k0 = 0; k0 < n_filt; k0 += 2 * G * npair) {
        const int nf = std::min(2 * G * npair, n_filt - k0), bt = (nf + 1) / 2;
        CHK(big_cols(c, nullptr, taps + (int64_t)k0 * n_taps, nf, n_taps, n_taps, P, L, bt));
        CHK(big_rows(c, P, S, L, bt));
        dsbig::UnpackArgs u{S, L, nf, 1.0f, R + (int64_t)k0 * nb, 1, nb};
        CHK(launch(c, "bigfft_unpack", dsbig::k_big_unpack, dim3(1024, bt), 256, 0, u));
    }
    for (int64_t b0 = 0; b0 < n_samples; b0 += step) {
        dsbig::ColsArgs a{nullptr, x, nullptr, n_samples, 0, P, L, n2, ct, 0, ldx, n_ch, tw,
                          nullptr, nullptr, 0, 0, 0, 0, 0, b0 - (n_taps - 1)};
        CHK(launch(c, "bigfft_cols", dsbig::k_big_cols<N1>, dim3(n2 / ct, npair), ct * Cfg<N1>::NT, lds, a));
        CHK(big_rows(c, P, Qs, L, npair));  // spectrum of the block, kept while the filter groups reuse P / S
        const int64_t n_out = std::min<int64_t>(step, n_samples - b0);
        for (int k0 = 0; k0 < n_filt; k0 += G) {
            const int g = std::min(G, n_filt - k0), bt = g * npair;
            dsbig::MulBankArgs m{Qs, R + (int64_t)k0 * nb, P, L, npair};
            CHK(launch(c, "bigfft_mul", dsbig::k_big_mul_bank, dim3(1024, bt), 256, 0, m));
            CHK(big_cols(c, P, nullptr, n_ch, 0, 0, P, L, bt));
            CHK(big_rows(c, P, S, L, bt));
            dsbig::StoreArgs st{S, L, n_out, ld_y, n_ch, y + (int64_t)k0 * n_ch * ld_y + b0, (int64_t)n_taps - 1};
            CHK(launch(c, "bigfft_store", dsbig::k_big_store, dim3(1024, bt), 256, 0, st));
        }
    }
    return DS_OK;
}

// Up to 4097 taps: uniformly partitioned overlap-save on the 4096-point register transform, three independent
// workgroups per CU (kernels_fir4k.hpp; two partitions: k_fir3 since round 5, DSPTOOLBOX_AMD_FIR_3PERCU=0 keeps the
// two-per-CU k_fir<2> with its tap-spectrum prefetch).
// DSPTOOLBOX_AMD_FIR_4K=0 keeps the block kernels below (A/B); =1 also sends the short filters here.
static int fir4k_run(ds_ctx* c, const float* x, int n_ch, int64_t ldx, int64_t n_samples, const float* taps,
                     int n_filt, int n_taps, float* y, int64_t ld_y) {
    namespace f4 = fir4k;
    if (!c->w4_tables) {
        std::vector<float2> h;
        welch4096::host_tables(h);
        HIPCHK(c, hipMalloc((void**)&c->w4_tables, sizeof(float2) * h.size()));
        HIPCHK(c, hipMemcpyAsync(c->w4_tables, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const int P = f4::partitions(n_taps);
    CHK(reserve(c, &c->ws, &c->ws_bytes, Carver::pad(sizeof(float4) * (size_t)n_filt * P * 8 * 256)));
    Carver cv(c->ws);
    float4* hp = cv.take<float4>((size_t)n_filt * P * 8 * 256);
    f4::TapArgs ta{taps, n_filt, n_taps, P, c->w4_tables, hp};
    CHK(launch(c, "fir_taps", f4::k_taps, dim3((unsigned)(n_filt * P)), f4::NT, f4::LDS_BYTES, ta));
    const int pairs = (n_ch + 1) / 2;
    const int n_blocks = (int)((n_samples + f4::HOP - 1) / f4::HOP);
    // every workgroup resident at once (3 or 2 per CU): a run of blocks each, one extra forward
    // transform per run with two partitions
    const bool three = P == 2 && c->cfg.fir_3percu && !c->cfg.fir_stage;
    int chunks = std::max(1, ((P == 1 || three) ? 768 : 512) / pairs);
    if (c->cfg.fir_chunks > 0) chunks = c->cfg.fir_chunks;
    chunks = std::min(chunks, n_blocks);
    f4::Args a{x, n_samples, ldx, ld_y, n_ch, n_filt, n_blocks, chunks, c->w4_tables, hp, y};
    if (c->cfg.fir_stage) {  // round-5 experiment: stores through a per-wave LDS strip (profiles/r05_fir_staged_stores.txt)
        if (P == 1)
            return launch(c, "fir@4k_p1_staged", f4::k_fir<1, true>, dim3((unsigned)(pairs * chunks)), f4::NT, f4::LDS_BYTES + f4::STAGE_BYTES, a);
        return launch(c, "fir@4k_p2_staged", f4::k_fir<2, true>, dim3((unsigned)(pairs * chunks)), f4::NT, f4::LDS_BYTES + f4::STAGE_BYTES, a);
    }
    if (P == 1) return launch(c, "fir@4k_p1", f4::k_fir<1>, dim3((unsigned)(pairs * chunks)), f4::NT, f4::LDS_BYTES, a);
    if (three) return launch(c, "fir@4k_p2_3percu", f4::k_fir3<0>, dim3((unsigned)(pairs * chunks)), f4::NT, f4::LDS_BYTES, a);
    return launch(c, "fir@4k_p2", f4::k_fir<2>, dim3((unsigned)(pairs * chunks)), f4::NT, f4::LDS_BYTES, a);
}

static int fir_once(ds_ctx* c, const float* x, int n_ch, int64_t ldx, int64_t n_samples,
                    const float* taps, int n_filt, int n_taps, float* y, int64_t ld_y) {
    // a signal shorter than the filter (or a tiny one): the direct sum in float64 -- no rounding floor set by the block's
    // peak, which is what an FFT convolution leaves on an output far below (peak of the block) x (size of the taps)
    // (kernels_freqz.hpp; DESIGN section 2, limit (x)).  At most 2^28 multiply-adds.
    // Every output sample is one thread's sum over min(n_samples, n_taps) products: at most 16384 of them (a 2^20-tap
    // filter on 256 samples would be 256 threads of a million dependent steps each).  DSPTOOLBOX_AMD_FIR_DIRECT=0 turns
    // the route off (A/B against the FFT routes on the same shape).
    if (c->cfg.fir_direct && (n_samples < n_taps || n_samples <= 512) && std::min<int64_t>(n_samples, n_taps) <= 16384 &&
        n_samples * std::min<int64_t>(n_samples, n_taps) * n_ch * n_filt <= ((int64_t)1 << 28) && n_filt <= 65535 && n_ch <= 65535) {
        freqz::DirectArgs a{x, taps, n_samples, ldx, ld_y, n_ch, n_taps, y};
        return launch(c, "fir@direct_f64", freqz::k_fir_direct, dim3((unsigned)((n_samples + 255) / 256), n_ch, n_filt), 256, 0, a);
    }
    if (n_taps >= c->cfg.fir4k_min_taps && fir4k::partitions(n_taps) <= 2 && fir4k::fits(n_samples) && n_filt <= 16384)
        return fir4k_run(c, x, n_ch, ldx, n_samples, taps, n_filt, n_taps, y, ld_y);
    const int N = fir_block_len(n_taps, c->cfg.fir_block);
    if (n_taps - 1 > N / 2) return fir_long(c, x, n_ch, ldx, n_samples, taps, n_filt, n_taps, y, ld_y);
    const float2* tw;
    CHK(get_twiddles(c, N, &tw));
    const bool no16k = c->cfg.fir_generic;
    const bool use16k = N == fir16k::NBIG && !no16k;
    CHK(reserve(c, &c->ws, &c->ws_bytes, (use16k ? 2 : 1) * Carver::pad(sizeof(float2) * (size_t)n_filt * N)));
    Carver cvw(c->ws);
    float2* hs = cvw.take<float2>((size_t)n_filt * N);
    {
        FirTapsArgs a{taps, n_filt, n_taps, tw, hs};
        DISPATCH_N(N, CHK(launch(c, "fir_taps", k_fir_taps<NN>, dim3((n_filt + 1) / 2), Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, a)));
    }
    const int L = N - (n_taps - 1);
    const int64_t n_blocks = (n_samples + L - 1) / L;
    if (use16k) {
        // 16384-point blocks: four 4096-point register transforms per block (kernels_fir16k.hpp)
        if (!c->w4_tables) {
            std::vector<float2> h;
            welch4096::host_tables(h);
            HIPCHK(c, hipMalloc((void**)&c->w4_tables, sizeof(float2) * h.size()));
            HIPCHK(c, hipMemcpyAsync(c->w4_tables, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (!c->fir16k_tables) {
            std::vector<float2> h;
            fir16k::host_tables(h);
            HIPCHK(c, hipMalloc((void**)&c->fir16k_tables, sizeof(float2) * h.size()));
            HIPCHK(c, hipMemcpyAsync(c->fir16k_tables, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        float2* hperm = cvw.take<float2>((size_t)n_filt * N);
        fir16k::PermArgs pa{hs, n_filt, hperm};
        CHK(launch(c, "fir_taps", fir16k::k_permute, dim3((unsigned)(((int64_t)n_filt * N + 255) / 256)), 256, 0, pa));
        fir16k::Args a{x, n_samples, ldx, ld_y, n_ch, n_filt, n_taps, c->w4_tables, c->fir16k_tables, hperm, y, 0};
        // interior blocks of a 4097-tap filter with 16-byte aligned rows: the store-everything variant
        int64_t n_plain = 0;
        if (((n_taps - 1) & 3) == 0 && (ld_y & 3) == 0 && (((uintptr_t)y) & 15) == 0) n_plain = n_samples / L;
        // The few ragged blocks go to the side stream so they run beside the main grid's last,
        // partly filled round instead of after it (fork after the tap spectra, join at the end).
        const bool ragged = n_blocks > n_plain;
        if (ragged && n_plain > 0) {
            HIPCHK(c, hipFuncSetAttribute((const void*)fir16k::k_fir<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)fir16k::LDS_BYTES));
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
            fir16k::Args ar = a;
            ar.block0 = (int)n_plain;
            hipLaunchKernelGGL(fir16k::k_fir<false>, dim3((unsigned)(n_blocks - n_plain), (n_ch + 1) / 2),
                               dim3(fir16k::NTB), fir16k::LDS_BYTES, c->side, ar);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(c->ev_join, c->side));
        }
        if (n_plain > 0) {
            // one workgroup per CU: split the filter loop over 1, 2 or 4 workgroups per block when that
            // fills the last round better (cost ~ rounds x (filters per slice + 1 forward transform))
            int split = 1;
            {
                const int64_t wgs = n_plain * ((n_ch + 1) / 2);
                int64_t best = -1;
                for (int s = 1; s <= 4 && s <= n_filt; s *= 2) {
                    const int64_t cost = ((wgs * s + 255) / 256) * ((n_filt + s - 1) / s + 1);
                    if (best < 0 || cost < best) {
                        best = cost;
                        split = s;
                    }
                }
                if (c->cfg.fir_split > 0) split = std::min(c->cfg.fir_split, n_filt);
            }
            CHK(launch(c, "fir@16k", fir16k::k_fir<true>, dim3((unsigned)n_plain, (n_ch + 1) / 2, split), fir16k::NTB,
                       fir16k::LDS_BYTES, a));
        }
        if (ragged && n_plain > 0) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        } else if (ragged) {
            CHK(launch(c, "fir@16k_ragged", fir16k::k_fir<false>, dim3((unsigned)n_blocks, (n_ch + 1) / 2), fir16k::NTB,
                       fir16k::LDS_BYTES, a));
        }
        return DS_OK;
    }
    FirArgs a{x, n_samples, ldx, ld_y, n_ch, n_filt, n_taps, tw, hs, y};
    dim3 grid((unsigned)n_blocks, (n_ch + 1) / 2);
    DISPATCH_N(N, CHK(launch(c, "fir@generic", k_fir<NN>, grid, Cfg<NN>::NT, Cfg<NN>::LDS_BYTES, a)));
    return DS_OK;
}

__global__ void k_sum_taps(const float* taps, int n_filt, int n_taps, float* out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_taps) return;
    double s = 0.0;
    for (int k = 0; k < n_filt; ++k) s += (double)taps[(int64_t)k * n_taps + t];
    out[t] = (float)s;
}

__global__ void k_taps_to_f64(const float* a, int n, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)a[i];
}
__global__ void k_f64_to_taps(const double* a, int n, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)a[i];
}
// out[n] = sum_k a[k] b[n-k], fp64 accumulate
__global__ void k_conv_taps(const double* a, int na, const float* b, int nb, double* out) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= na + nb - 1) return;
    int k0 = max(0, n - (nb - 1)), k1 = min(na - 1, n);
    double s = 0.0;
    for (int k = k0; k <= k1; ++k) s += a[k] * (double)b[n - k];
    out[n] = s;
}

extern "C" int ds_fir_ola_dev(ds_ctx* c, const float* x, int n_ch, int64_t ldx, int64_t n_samples,
                              const float* taps, int n_filt, int n_taps, int mode, float* y,
                              int64_t ld_y) {
    if (!c || !x || !taps || !y) return fail(c, DS_ERR_ARG, "ds_fir_ola: null argument");
    if (n_ch <= 0 || n_samples <= 0 || n_filt <= 0 || n_taps <= 0 || ldx < n_samples || ld_y < n_samples)
        return fail(c, DS_ERR_ARG, "ds_fir_ola: bad shape");
    if (mode == DS_FB_PARALLEL) return fir_once(c, x, n_ch, ldx, n_samples, taps, n_filt, n_taps, y, ld_y);
    // (combined taps and the cascade's intermediate signal live in the context's aux scratch: no
    // allocation, free or synchronisation per call; everything stays ordered on the stream)
    if (mode == DS_FB_SUMMED) {
        // sum_k (x * b_k) = x * (sum_k b_k): one filter with the summed taps
        CHK(reserve(c, &c->aux, &c->aux_bytes, sizeof(float) * (size_t)n_taps));
        float* bs = (float*)c->aux;
        hipLaunchKernelGGL(k_sum_taps, dim3((n_taps + 255) / 256), dim3(256), 0, c->stream, taps, n_filt, n_taps, bs);
        HIPCHK(c, hipGetLastError());
        return fir_once(c, x, n_ch, ldx, n_samples, bs, 1, n_taps, y, ld_y);
    }
    if (mode == DS_FB_SEQUENTIAL) {
        // ((x*b1)[:N]*b2)[:N]... = (x*(b1*b2*...))[:N] for causal filters: when the combined
        // response fits one block transform, convolve the taps (fp64, on the device) and
        // filter once -- no fp32 round trip of the intermediate signals through HBM.
        const int64_t n_comb = (int64_t)n_filt * (n_taps - 1) + 1;
        if (n_filt > 1 && n_comb - 1 <= kMaxBigFft / 2 && (double)n_comb * n_taps <= 1.0e10) {
            CHK(reserve(c, &c->aux, &c->aux_bytes,
                        2 * Carver::pad(sizeof(double) * (size_t)n_comb) + Carver::pad(sizeof(float) * (size_t)n_comb)));
            Carver cv(c->aux);
            double* pa = cv.take<double>((size_t)n_comb);
            double* pb = cv.take<double>((size_t)n_comb);
            float* bf = cv.take<float>((size_t)n_comb);
            hipLaunchKernelGGL(k_taps_to_f64, dim3((n_taps + 255) / 256), dim3(256), 0, c->stream, taps, n_taps, pa);
            int len = n_taps;
            for (int k = 1; k < n_filt; ++k) {
                int nl = len + n_taps - 1;
                hipLaunchKernelGGL(k_conv_taps, dim3((nl + 255) / 256), dim3(256), 0, c->stream, pa, len,
                                   taps + (int64_t)k * n_taps, n_taps, pb);
                std::swap(pa, pb);
                len = nl;
            }
            hipLaunchKernelGGL(k_f64_to_taps, dim3((len + 255) / 256), dim3(256), 0, c->stream, pa, len, bf);
            HIPCHK(c, hipGetLastError());
            return fir_once(c, x, n_ch, ldx, n_samples, bf, 1, len, y, ld_y);
        }
        // long cascades: stage by stage, each truncated to n_samples like the reference loop
        float* tmp = nullptr;
        if (n_filt > 1) {
            CHK(reserve(c, &c->aux, &c->aux_bytes, sizeof(float) * (size_t)n_ch * n_samples));
            tmp = (float*)c->aux;
        }
        const float* src = x;
        int64_t lds = ldx;
        for (int k = 0; k < n_filt; ++k) {
            float* dst = ((n_filt - 1 - k) % 2 == 0) ? y : tmp;  // ping-pong, last stage lands in y
            int64_t ldd = (dst == y) ? ld_y : n_samples;
            CHK(fir_once(c, src, n_ch, lds, n_samples, taps + (int64_t)k * n_taps, 1, n_taps, dst, ldd));
            src = dst;
            lds = ldd;
        }
        return DS_OK;
    }
    return fail(c, DS_ERR_ARG, "ds_fir_ola: invalid filter bank apply mode");
}

#if W4_TIMING
// dev only (built with -DW4_TIMING=1): read and reset the per-phase cycle stamps
extern "C" int ds_debug_welch_timing(unsigned long long out[16]) {
    unsigned long long z[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(welch4096::w4_timing), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(welch4096::w4_timing), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif

// ---- host-pointer entry points -------------------------------------------------
struct Stage {  // device staging out of ctx->io
    ds_ctx* c;
    Carver cv;
    explicit Stage(ds_ctx* ctx) : c(ctx), cv(ctx->io) {}
};
static int stage_reserve(ds_ctx* c, size_t bytes) { return reserve(c, &c->io, &c->io_bytes, bytes + 4096); }

extern "C" int ds_stft_r2c(ds_ctx* c, const float* x, int64_t n_samples, int n_ch, int W, int hop,
                           int nfft, int64_t pad_front, int n_frames, const float* window, int detrend,
                           float scale, float edge_scale, int power, ds_c32* out) {
    if (!c || !x || !window || !out) return fail(c, DS_ERR_ARG, "ds_stft_r2c: null argument");
    if (n_ch <= 0 || n_samples <= 0 || W <= 0 || n_frames <= 0 || nfft <= 0) return fail(c, DS_ERR_ARG, "ds_stft_r2c: bad shape");
    size_t nx = (size_t)n_ch * n_samples, no = (size_t)(nfft / 2 + 1) * n_frames * n_ch;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad((size_t)W * 4) + Carver::pad(no * 8)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dw = cv.take<float>(W);
    float2* dout = cv.take<float2>(no);
    CHK(ds_upload(c, dx, x, nx * 4));
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(ds_stft_r2c_dev(c, dx, n_samples, n_ch, n_samples, W, hop, nfft, pad_front, n_frames, dw, detrend,
                        scale, edge_scale, power, (ds_c32*)dout));
    return ds_download(c, out, dout, no * 8);
}

extern "C" int ds_istft(ds_ctx* c, const ds_c32* stft, int n_bins, int n_frames, int n_ch, int nfft, int W,
                        int step, int frame_offset, int n_frames_total, const float* window, float scale,
                        int64_t total_length, float* out) {
    if (!c || !stft || !window || !out) return fail(c, DS_ERR_ARG, "ds_istft: null argument");
    if (n_bins <= 0 || n_frames <= 0 || n_ch <= 0 || W <= 0 || total_length <= 0)
        return fail(c, DS_ERR_ARG, "ds_istft: bad shape");
    size_t ns = (size_t)n_bins * n_frames * n_ch, no = (size_t)n_ch * total_length;
    CHK(stage_reserve(c, Carver::pad(ns * 8) + Carver::pad((size_t)W * 4) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float2* ds = cv.take<float2>(ns);
    float* dw = cv.take<float>(W);
    float* dout = cv.take<float>(no);
    CHK(ds_upload(c, ds, stft, ns * 8));
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(ds_istft_dev(c, (const ds_c32*)ds, n_bins, n_frames, n_ch, nfft, W, step, frame_offset, n_frames_total,
                     dw, scale, total_length, dout, total_length));
    return ds_download(c, out, dout, no * 4);
}

// Fused boundary upload: (samples, channels) float64 C-order host array -> planar float32 device
// rows.  Chunks of samples are cast + transposed by host threads straight into one of two pinned
// buffers and sent with an asynchronous 2-D copy, so the cast of chunk k+1 overlaps the DMA of
// chunk k (a pageable hipMemcpy of the pre-cast array moves ~13 GB/s; this path is bound by the
// threads' cast, ~35 GB/s of float64 input).
static const size_t kPinBytes = (size_t)32 << 20;
// The transport of the pipelines in host_marshal.hpp: asynchronous copies on the context's stream, one
// event per pinned staging chunk.
struct HipTransport {
    ds_ctx* c;
    hipError_t err = hipSuccess;
    bool ok(hipError_t e) {
        if (e != hipSuccess) err = e;
        return e == hipSuccess;
    }
    bool mark(int b) {
        if (!ok(hipEventRecord(c->pin_ev[b], c->stream))) return false;
        c->pin_busy[b] = true;
        return true;
    }
    bool wait(int b) {
        if (c->pin_busy[b] && !ok(hipEventSynchronize(c->pin_ev[b]))) return false;
        c->pin_busy[b] = false;
        return true;
    }
    bool h2d_2d(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t rows, int b) {
        return ok(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyHostToDevice, c->stream)) && mark(b);
    }
    bool d2h_2d(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t rows, int b) {
        return ok(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToHost, c->stream)) && mark(b);
    }
    bool d2h(float* dst, const float* src, size_t bytes, int b) {
        return ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)) && mark(b);
    }
    bool h2d(float* dst, const float* src, size_t bytes, int b) {
        return ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)) && mark(b);
    }
};
static int pin_ready(ds_ctx* c, bool drain) {
    for (int i = 0; i < 2; ++i) {
        if (!c->pin[i]) {
            HIPCHK(c, hipHostMalloc(&c->pin[i], kPinBytes, hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
        }
        if (drain && c->pin_busy[i]) {
            HIPCHK(c, hipEventSynchronize(c->pin_ev[i]));
            c->pin_busy[i] = false;
        }
    }
    return DS_OK;
}
static int pipe_result(ds_ctx* c, bool ok, const HipTransport& tr, const char* what) {
    if (ok) return DS_OK;
    if (tr.err != hipSuccess) return fail(c, DS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(tr.err));
    return fail(c, DS_ERR_UNSUP, std::string(what) + ": too many channels for the staging chunk");
}
static int upload_planar_f64(ds_ctx* c, const double* src, int64_t n_samples, int n_ch, float* dst_dev,
                             int64_t ld) {
    CHK(pin_ready(c, false));
    HipTransport tr{c};
    float* pin[2] = {(float*)c->pin[0], (float*)c->pin[1]};
    return pipe_result(c, dshost::upload_planar(tr, pin, kPinBytes, src, n_samples, n_ch, dst_dev, ld), tr, "upload_planar_f64");
}
static int upload_narrow_f64(ds_ctx* c, const double* src, int64_t n, float* dst_dev) {
    CHK(pin_ready(c, false));
    HipTransport tr{c};
    float* pin[2] = {(float*)c->pin[0], (float*)c->pin[1]};
    return pipe_result(c, dshost::upload_narrow(tr, pin, kPinBytes, src, n, dst_dev), tr, "upload_narrow_f64");
}
static int download_widen(ds_ctx* c, const float* src_dev, int64_t n, double* dst) {
    CHK(pin_ready(c, true));
    HipTransport tr{c};
    float* pin[2] = {(float*)c->pin[0], (float*)c->pin[1]};
    return pipe_result(c, dshost::download_widen(tr, pin, kPinBytes, src_dev, n, dst), tr, "download_widen");
}
static int download_interleave(ds_ctx* c, const float* src_dev, int64_t n_samples, int n_ch, int64_t ld,
                               double* dst) {
    CHK(pin_ready(c, true));
    HipTransport tr{c};
    float* pin[2] = {(float*)c->pin[0], (float*)c->pin[1]};
    return pipe_result(c, dshost::download_interleave(tr, pin, kPinBytes, src_dev, n_samples, n_ch, ld, dst), tr,
                       "download_interleave");
}

// ds_stft_r2c with the reference's layouts on both sides: x (n_samples, n_ch) float64 C-order in,
// (bins, frames, channels) complex128 out.
extern "C" int ds_stft_r2c_f64(ds_ctx* c, const double* x, int64_t n_samples, int n_ch, int W, int hop,
                               int nfft, int64_t pad_front, int n_frames, const float* window, int detrend,
                               float scale, float edge_scale, int power, double* out_c128) {
    if (!c || !x || !window || !out_c128) return fail(c, DS_ERR_ARG, "ds_stft_r2c_f64: null argument");
    if (n_ch <= 0 || n_samples <= 0 || W <= 0 || n_frames <= 0 || nfft <= 0)
        return fail(c, DS_ERR_ARG, "ds_stft_r2c_f64: bad shape");
    size_t nx = (size_t)n_ch * n_samples, no = (size_t)(nfft / 2 + 1) * n_frames * n_ch;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad((size_t)W * 4) + Carver::pad(no * 8)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dw = cv.take<float>(W);
    float2* dout = cv.take<float2>(no);
    CHK(upload_planar_f64(c, x, n_samples, n_ch, dx, n_samples));
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(ds_stft_r2c_dev(c, dx, n_samples, n_ch, n_samples, W, hop, nfft, pad_front, n_frames, dw, detrend,
                        scale, edge_scale, power, (ds_c32*)dout));
    return download_widen(c, (const float*)dout, (int64_t)no * 2, out_c128);
}

// ds_rfft with the reference's layouts on both sides: x (n_samples, n_ch) float64 C-order in, (bins, channels)
// complex128 out (Signal.get_spectrum with SpectrumMethod.FFT, classes/signal.py:899-911).
extern "C" int ds_rfft_f64(ds_ctx* c, const double* x, int n_ch, int64_t n_samples, int n_fft, float scale,
                           double* spec_c128) {
    if (!c || !x || !spec_c128) return fail(c, DS_ERR_ARG, "ds_rfft_f64: null argument");
    if (n_ch <= 0 || n_samples <= 0 || n_fft < 2) return fail(c, DS_ERR_ARG, "ds_rfft_f64: bad shape");
    const size_t nx = (size_t)n_ch * n_samples, no = (size_t)(n_fft / 2 + 1) * n_ch;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad(no * 8)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float2* ds = cv.take<float2>(no);
    CHK(upload_planar_f64(c, x, n_samples, n_ch, dx, n_samples));
    CHK(ds_rfft_dev(c, dx, n_ch, n_samples, n_samples, n_fft, scale, (ds_c32*)ds));
    return download_widen(c, (const float*)ds, (int64_t)no * 2, spec_c128);
}

// ds_deconv for ONE item in the reference's layouts: y (n_samples, n_ch) float64 in, the impulse responses
// (n_out, n_ch) float64 out (_spectral_deconvolve, transfer_functions/_transfer_functions.py:19-42).
extern "C" int ds_deconv_f64(ds_ctx* c, const double* y, int n_ch, int64_t n_samples, int n_fft, const ds_c32* r,
                             int r_per_channel, int64_t n_out, double* ir) {
    if (!c || !y || !r || !ir) return fail(c, DS_ERR_ARG, "ds_deconv_f64: null argument");
    if (n_ch <= 0 || n_samples <= 0 || n_fft < 2 || n_out <= 0 || n_out > n_fft) return fail(c, DS_ERR_ARG, "ds_deconv_f64: bad shape");
    const size_t ny = (size_t)n_ch * n_samples, nr = (size_t)(r_per_channel ? n_ch : 1) * (n_fft / 2 + 1), no = (size_t)n_ch * n_out;
    CHK(stage_reserve(c, Carver::pad(ny * 4) + Carver::pad(nr * 8) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float* dy = cv.take<float>(ny);
    float2* dr = cv.take<float2>(nr);
    float* dout = cv.take<float>(no);
    CHK(upload_planar_f64(c, y, n_samples, n_ch, dy, n_samples));
    CHK(ds_upload(c, dr, r, nr * 8));
    CHK(ds_deconv_dev(c, dy, 1, n_ch, n_samples, n_samples, n_fft, (const ds_c32*)dr, r_per_channel, n_out, n_out, dout));
    return download_interleave(c, dout, n_out, n_ch, n_out, ir);
}

// ds_istft with the reference's layouts on both sides: the spectrogram (bins, frames, channels) complex128 in,
// the signal (total_length, n_ch) float64 out (transforms.istft, transforms/transforms.py:444-586).
extern "C" int ds_istft_f64(ds_ctx* c, const double* stft_c128, int n_bins, int n_frames, int n_ch, int nfft, int W,
                            int step, int frame_offset, int n_frames_total, const float* window, float scale,
                            int64_t total_length, double* out) {
    if (!c || !stft_c128 || !window || !out) return fail(c, DS_ERR_ARG, "ds_istft_f64: null argument");
    if (n_bins <= 0 || n_frames <= 0 || n_ch <= 0 || W <= 0 || total_length <= 0) return fail(c, DS_ERR_ARG, "ds_istft_f64: bad shape");
    const size_t ns = (size_t)n_bins * n_frames * n_ch, no = (size_t)n_ch * total_length;
    CHK(stage_reserve(c, Carver::pad(ns * 8) + Carver::pad((size_t)W * 4) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float2* dsp = cv.take<float2>(ns);
    float* dw = cv.take<float>(W);
    float* dout = cv.take<float>(no);
    CHK(upload_narrow_f64(c, stft_c128, (int64_t)ns * 2, (float*)dsp));
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(ds_istft_dev(c, (const ds_c32*)dsp, n_bins, n_frames, n_ch, nfft, W, step, frame_offset, n_frames_total, dw, scale,
                     total_length, dout, total_length));
    return download_interleave(c, dout, total_length, n_ch, total_length, out);
}

// ds_fir_ola with the reference's layouts on both sides: x (n_samples, n_ch) float64 in,
// y (bands or 1, n_samples, n_ch) float64 out.
extern "C" int ds_fir_ola_f64(ds_ctx* c, const double* x, int n_ch, int64_t n_samples, const float* taps,
                              int n_filt, int n_taps, int mode, double* y) {
    if (!c || !x || !taps || !y) return fail(c, DS_ERR_ARG, "ds_fir_ola_f64: null argument");
    if (n_ch <= 0 || n_samples <= 0 || n_filt <= 0 || n_taps <= 0) return fail(c, DS_ERR_ARG, "ds_fir_ola_f64: bad shape");
    const int n_out = mode == DS_FB_PARALLEL ? n_filt : 1;
    size_t nx = (size_t)n_ch * n_samples, nt = (size_t)n_filt * n_taps, no = (size_t)n_out * n_ch * n_samples;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad(nt * 4) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dt = cv.take<float>(nt);
    float* dy = cv.take<float>(no);
    CHK(upload_planar_f64(c, x, n_samples, n_ch, dx, n_samples));
    CHK(ds_upload(c, dt, taps, nt * 4));
    CHK(ds_fir_ola_dev(c, dx, n_ch, n_samples, n_samples, dt, n_filt, n_taps, mode, dy, n_samples));
    for (int k = 0; k < n_out; ++k)
        CHK(download_interleave(c, dy + (size_t)k * n_ch * n_samples, n_samples, n_ch, n_samples,
                                y + (size_t)k * n_samples * n_ch));
    return DS_OK;
}

// ds_welch_tf with the reference's own array layout at the boundary: x (n_samples, n_cx) and
// y (n_samples, n_cy) float64 C-order (classes/signal.py:222-301), outputs as ds_welch_tf.
extern "C" int ds_welch_tf_f64(ds_ctx* c, const double* x, int n_cx, const double* y, int n_cy,
                               int64_t n_samples, int W, int hop, int n_frames, const float* window,
                               int detrend, int average, int mode, int amp_sqrt, double norm_scale,
                               double factor, int halve_edges, ds_c32* tf, float* coh) {
    if (!c || !x || !y || !window || !tf || !coh) return fail(c, DS_ERR_ARG, "ds_welch_tf_f64: null argument");
    if (n_cx <= 0 || n_cy <= 0 || n_samples <= 0 || W <= 0) return fail(c, DS_ERR_ARG, "ds_welch_tf_f64: bad shape");
    size_t nx = (size_t)n_cx * n_samples, ny = (size_t)n_cy * n_samples, no = (size_t)(W / 2 + 1) * n_cy;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad(ny * 4) + Carver::pad((size_t)W * 4) +
                             Carver::pad(no * 8) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dy = cv.take<float>(ny);
    float* dw = cv.take<float>(W);
    float2* dtf = cv.take<float2>(no);
    float* dcoh = cv.take<float>(no);
    CHK(upload_planar_f64(c, x, n_samples, n_cx, dx, n_samples));
    CHK(upload_planar_f64(c, y, n_samples, n_cy, dy, n_samples));
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(ds_welch_tf_dev(c, dx, n_cx, n_samples, dy, n_cy, n_samples, n_samples, W, hop, n_frames, dw,
                        detrend, average, mode, amp_sqrt, norm_scale, factor, halve_edges, (ds_c32*)dtf,
                        dcoh));
    CHK(ds_download(c, tf, dtf, no * 8));
    return ds_download(c, coh, dcoh, no * 4);
}

extern "C" int ds_welch_tf(ds_ctx* c, const float* x, int n_cx, const float* y, int n_cy,
                           int64_t n_samples, int W, int hop, int n_frames, const float* window,
                           int detrend, int average, int mode, int amp_sqrt, double norm_scale,
                           double factor, int halve_edges, ds_c32* tf, float* coh) {
    if (!c || !x || !y || !window || !tf || !coh) return fail(c, DS_ERR_ARG, "ds_welch_tf: null argument");
    if (n_cx <= 0 || n_cy <= 0 || n_samples <= 0 || W <= 0) return fail(c, DS_ERR_ARG, "ds_welch_tf: bad shape");
    size_t nx = (size_t)n_cx * n_samples, ny = (size_t)n_cy * n_samples, no = (size_t)(W / 2 + 1) * n_cy;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad(ny * 4) + Carver::pad((size_t)W * 4) +
                             Carver::pad(no * 8) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dy = cv.take<float>(ny);
    float* dw = cv.take<float>(W);
    float2* dtf = cv.take<float2>(no);
    float* dcoh = cv.take<float>(no);
    CHK(ds_upload(c, dx, x, nx * 4));
    CHK(ds_upload(c, dy, y, ny * 4));
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(ds_welch_tf_dev(c, dx, n_cx, n_samples, dy, n_cy, n_samples, n_samples, W, hop, n_frames, dw,
                        detrend, average, mode, amp_sqrt, norm_scale, factor, halve_edges, (ds_c32*)dtf,
                        dcoh));
    CHK(ds_download(c, tf, dtf, no * 8));
    return ds_download(c, coh, dcoh, no * 4);
}

static int welch_psd_host(ds_ctx* c, const float* x, const double* x64, int n_cx, int64_t n_samples, int W,
                          int hop, int n_frames, const float* window, int detrend, int average, int amp_sqrt,
                          double norm_scale, double factor, int halve_edges, float* psd) {
    if (!c || (!x && !x64) || !window || !psd) return fail(c, DS_ERR_ARG, "ds_welch_psd: null argument");
    if (n_cx <= 0 || n_samples <= 0 || W <= 0) return fail(c, DS_ERR_ARG, "ds_welch_psd: bad shape");
    size_t nx = (size_t)n_cx * n_samples, no = (size_t)(W / 2 + 1) * n_cx;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad((size_t)W * 4) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dw = cv.take<float>(W);
    float* dp = cv.take<float>(no);
    if (x64)
        CHK(upload_planar_f64(c, x64, n_samples, n_cx, dx, n_samples));
    else
        CHK(ds_upload(c, dx, x, nx * 4));
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(ds_welch_psd_dev(c, dx, n_cx, n_samples, n_samples, W, hop, n_frames, dw, detrend, average,
                         amp_sqrt, norm_scale, factor, halve_edges, dp));
    return ds_download(c, psd, dp, no * 4);
}
extern "C" int ds_welch_psd(ds_ctx* c, const float* x, int n_cx, int64_t n_samples, int W, int hop,
                            int n_frames, const float* window, int detrend, int average, int amp_sqrt,
                            double norm_scale, double factor, int halve_edges, float* psd) {
    return welch_psd_host(c, x, nullptr, n_cx, n_samples, W, hop, n_frames, window, detrend, average, amp_sqrt,
                          norm_scale, factor, halve_edges, psd);
}
extern "C" int ds_welch_psd_f64(ds_ctx* c, const double* x, int n_cx, int64_t n_samples, int W, int hop,
                                int n_frames, const float* window, int detrend, int average, int amp_sqrt,
                                double norm_scale, double factor, int halve_edges, float* psd) {
    return welch_psd_host(c, nullptr, x, n_cx, n_samples, W, hop, n_frames, window, detrend, average, amp_sqrt,
                          norm_scale, factor, halve_edges, psd);
}

// x / y: planar float32 [n_ch][n_samples], or x64 / y64: the reference's (n_samples, n_ch) float64 arrays as they are
static int welch_csd_host(ds_ctx* c, const float* x, const float* y, const double* x64, const double* y64, int n_ch,
                          int64_t n_samples, int W, int hop, int n_frames, const float* window, int detrend, int average,
                          int amp_sqrt, double norm_scale, double factor, int halve_edges, ds_c32* csd) {
    if (!c || (!x && !x64) || (!y && !y64) || !window || !csd) return fail(c, DS_ERR_ARG, "ds_welch_csd: null argument");
    if (n_ch <= 0 || n_samples <= 0 || W <= 0) return fail(c, DS_ERR_ARG, "ds_welch_csd: bad shape");
    size_t nx = (size_t)n_ch * n_samples, no = (size_t)(W / 2 + 1) * n_ch;
    CHK(stage_reserve(c, 2 * Carver::pad(nx * 4) + Carver::pad((size_t)W * 4) + Carver::pad(no * 8)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dy = cv.take<float>(nx);
    float* dw = cv.take<float>(W);
    float2* dc = cv.take<float2>(no);
    if (x64) {
        CHK(upload_planar_f64(c, x64, n_samples, n_ch, dx, n_samples));
        CHK(upload_planar_f64(c, y64, n_samples, n_ch, dy, n_samples));
    } else {
        CHK(ds_upload(c, dx, x, nx * 4));
        CHK(ds_upload(c, dy, y, nx * 4));
    }
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(welch_csd_dev(c, dx, dy, n_ch, n_samples, n_samples, W, hop, n_frames, dw, detrend, average,
                      amp_sqrt, norm_scale, factor, halve_edges, (ds_c32*)dc));
    return ds_download(c, csd, dc, no * 8);
}
extern "C" int ds_welch_csd(ds_ctx* c, const float* x, const float* y, int n_ch, int64_t n_samples,
                            int W, int hop, int n_frames, const float* window, int detrend,
                            int average, int amp_sqrt, double norm_scale, double factor,
                            int halve_edges, ds_c32* csd) {
    return welch_csd_host(c, x, y, nullptr, nullptr, n_ch, n_samples, W, hop, n_frames, window, detrend, average, amp_sqrt,
                          norm_scale, factor, halve_edges, csd);
}
extern "C" int ds_welch_csd_f64(ds_ctx* c, const double* x, const double* y, int n_ch, int64_t n_samples,
                                int W, int hop, int n_frames, const float* window, int detrend,
                                int average, int amp_sqrt, double norm_scale, double factor,
                                int halve_edges, ds_c32* csd) {
    if (!x || !y) return fail(c, DS_ERR_ARG, "ds_welch_csd_f64: null argument");
    return welch_csd_host(c, nullptr, nullptr, x, y, n_ch, n_samples, W, hop, n_frames, window, detrend, average, amp_sqrt,
                          norm_scale, factor, halve_edges, csd);
}

static int csm_host(ds_ctx* c, const float* x, const double* x64, int n_ch, int64_t n_samples, int W, int hop,
                    int n_frames, const float* window, int detrend, int average, int amp_sqrt,
                    double norm_scale, double factor, int halve_edges, ds_c32* csm) {
    if (!c || (!x && !x64) || !window || !csm) return fail(c, DS_ERR_ARG, "ds_csm: null argument");
    if (n_ch <= 0 || n_samples <= 0 || W <= 0) return fail(c, DS_ERR_ARG, "ds_csm: bad shape");
    size_t nx = (size_t)n_ch * n_samples, no = (size_t)(W / 2 + 1) * n_ch * n_ch;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad((size_t)W * 4) + Carver::pad(no * 8)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dw = cv.take<float>(W);
    float2* dc = cv.take<float2>(no);
    if (x64)
        CHK(upload_planar_f64(c, x64, n_samples, n_ch, dx, n_samples));
    else
        CHK(ds_upload(c, dx, x, nx * 4));
    CHK(ds_upload(c, dw, window, (size_t)W * 4));
    CHK(ds_csm_dev(c, dx, n_ch, n_samples, n_samples, W, hop, n_frames, dw, detrend, average, amp_sqrt,
                   norm_scale, factor, halve_edges, (ds_c32*)dc));
    return ds_download(c, csm, dc, no * 8);
}
extern "C" int ds_csm(ds_ctx* c, const float* x, int n_ch, int64_t n_samples, int W, int hop,
                      int n_frames, const float* window, int detrend, int average, int amp_sqrt,
                      double norm_scale, double factor, int halve_edges, ds_c32* csm) {
    return csm_host(c, x, nullptr, n_ch, n_samples, W, hop, n_frames, window, detrend, average, amp_sqrt,
                    norm_scale, factor, halve_edges, csm);
}
extern "C" int ds_csm_f64(ds_ctx* c, const double* x, int n_ch, int64_t n_samples, int W, int hop,
                          int n_frames, const float* window, int detrend, int average, int amp_sqrt,
                          double norm_scale, double factor, int halve_edges, ds_c32* csm) {
    return csm_host(c, nullptr, x, n_ch, n_samples, W, hop, n_frames, window, detrend, average, amp_sqrt,
                    norm_scale, factor, halve_edges, csm);
}

extern "C" int ds_rfft(ds_ctx* c, const float* x, int n_ch, int64_t n_samples, int n_fft, float scale,
                       ds_c32* spec) {
    if (!c || !x || !spec) return fail(c, DS_ERR_ARG, "ds_rfft: null argument");
    if (n_ch <= 0 || n_samples <= 0 || n_fft <= 0) return fail(c, DS_ERR_ARG, "ds_rfft: bad shape");
    size_t nx = (size_t)n_ch * n_samples, no = (size_t)(n_fft / 2 + 1) * n_ch;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad(no * 8)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float2* ds = cv.take<float2>(no);
    CHK(ds_upload(c, dx, x, nx * 4));
    CHK(ds_rfft_dev(c, dx, n_ch, n_samples, n_samples, n_fft, scale, (ds_c32*)ds));
    return ds_download(c, spec, ds, no * 8);
}

extern "C" int ds_deconv(ds_ctx* c, const float* y, int n_items, int n_ch, int64_t n_samples, int n_fft,
                         const ds_c32* r, int r_per_channel, int64_t n_out, float* ir) {
    if (!c || !y || !r || !ir) return fail(c, DS_ERR_ARG, "ds_deconv: null argument");
    if (n_items <= 0 || n_ch <= 0 || n_samples <= 0 || n_fft <= 0 || n_out <= 0) return fail(c, DS_ERR_ARG, "ds_deconv: bad shape");
    size_t ny = (size_t)n_items * n_ch * n_samples, nr = (size_t)(r_per_channel ? n_ch : 1) * (n_fft / 2 + 1),
           no = (size_t)n_items * n_ch * n_out;
    CHK(stage_reserve(c, Carver::pad(ny * 4) + Carver::pad(nr * 8) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float* dy = cv.take<float>(ny);
    float2* dr = cv.take<float2>(nr);
    float* dir = cv.take<float>(no);
    CHK(ds_upload(c, dy, y, ny * 4));
    CHK(ds_upload(c, dr, r, nr * 8));
    CHK(ds_deconv_dev(c, dy, n_items, n_ch, n_samples, n_samples, n_fft, (const ds_c32*)dr, r_per_channel,
                      n_out, n_out, dir));
    return ds_download(c, ir, dir, no * 4);
}

extern "C" int ds_fir_ola(ds_ctx* c, const float* x, int n_ch, int64_t n_samples, const float* taps,
                          int n_filt, int n_taps, int mode, float* y) {
    if (!c || !x || !taps || !y) return fail(c, DS_ERR_ARG, "ds_fir_ola: null argument");
    if (n_ch <= 0 || n_samples <= 0 || n_filt <= 0 || n_taps <= 0) return fail(c, DS_ERR_ARG, "ds_fir_ola: bad shape");
    size_t nx = (size_t)n_ch * n_samples, nt = (size_t)n_filt * n_taps,
           no = (size_t)(mode == DS_FB_PARALLEL ? n_filt : 1) * n_ch * n_samples;
    CHK(stage_reserve(c, Carver::pad(nx * 4) + Carver::pad(nt * 4) + Carver::pad(no * 4)));
    Carver cv(c->io);
    float* dx = cv.take<float>(nx);
    float* dt = cv.take<float>(nt);
    float* dy = cv.take<float>(no);
    CHK(ds_upload(c, dx, x, nx * 4));
    CHK(ds_upload(c, dt, taps, nt * 4));
    CHK(ds_fir_ola_dev(c, dx, n_ch, n_samples, n_samples, dt, n_filt, n_taps, mode, dy, n_samples));
    return ds_download(c, y, dy, no * 4);
}

// ---- block-streaming FIR classes, state on the device (kernels_fir_stream.hpp) ----------
__global__ void k_stream_ones(float* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = 1.f;
}
static int stream_ones(ds_ctx* c, float* dst, int n) {  // n_call unit impulses of length 1, written on the
                                                         // stream: a block step carries no host synchronisation
    hipLaunchKernelGGL(k_stream_ones, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dst, n);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}

extern "C" int ds_fir_part_step_dev(ds_ctx* c, float* inbuf, const float* block, int bs, int n_ch, int ch0,
                                    int n_call, const ds_c32* h, int n_part, int n_fir_ch, ds_c32* delay,
                                    int ind, float* out) {
    if (!c || !inbuf || !block || !h || !delay || !out) return fail(c, DS_ERR_ARG, "ds_fir_part_step: null argument");
    if (bs < 1 || n_ch < 1 || ch0 < 0 || n_call < 1 || ch0 + n_call > n_ch || n_part < 1 || ind < 0 || ind >= n_part ||
        (n_fir_ch != 1 && n_fir_ch != n_ch))
        return fail(c, DS_ERR_ARG, "ds_fir_part_step: bad shape");
    const int n_fft = 2 * bs, B = bs + 1;
    CHK(reserve(c, &c->aux, &c->aux_bytes,
                2 * Carver::pad(sizeof(float2) * (size_t)B * n_call) + Carver::pad(sizeof(float) * (size_t)n_call * n_fft) +
                    Carver::pad(sizeof(float) * (size_t)n_call)));
    Carver cv(c->aux);
    float2* X = cv.take<float2>((size_t)B * n_call);
    float2* Y = cv.take<float2>((size_t)B * n_call);
    float* full = cv.take<float>((size_t)n_call * n_fft);
    float* ones = cv.take<float>(n_call);
    CHK(stream_ones(c, ones, n_call));
    const int64_t nb = (int64_t)n_call * bs;
    hipLaunchKernelGGL(firstream::k_shift_in, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, inbuf, block,
                       bs, ch0, n_call);
    HIPCHK(c, hipGetLastError());
    CHK(ds_rfft_dev(c, inbuf + (int64_t)ch0 * n_fft, n_call, n_fft, n_fft, n_fft, 1.0f, (ds_c32*)X));
    firstream::AccArgs aa{X, (float2*)delay, (const float2*)h, Y, B, n_part, n_ch, n_fir_ch, ch0, n_call, ind};
    const int64_t na = (int64_t)B * n_call;
    CHK(launch(c, "fir_part_acc", firstream::k_part_acc, dim3((unsigned)((na + 255) / 256)), 256, 0, aa));
    // numpy's irfft of bs + 1 bins without a length: 2 bs points
    CHK(ds_deconv_dev(c, ones, 1, n_call, 1, 1, n_fft, (const ds_c32*)Y, 1, n_fft, n_fft, full));
    hipLaunchKernelGGL(firstream::k_tail, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream,
                       (const float*)full, (int64_t)n_fft, bs, n_call, out);
    HIPCHK(c, hipGetLastError());
    return DS_OK;
}

extern "C" int ds_fir_ols_step_dev(ds_ctx* c, float* row, const float* block, int bs, int64_t L,
                                   const ds_c32* h, float* out) {
    if (!c || !row || !block || !h || !out) return fail(c, DS_ERR_ARG, "ds_fir_ols_step: null argument");
    if (bs < 1 || L < 2 || bs > L || L > ((int64_t)1 << 22)) return fail(c, DS_ERR_ARG, "ds_fir_ols_step: bad shape");
    const int B = (int)(L / 2 + 1);
    const int64_t n_inv = 2 * (int64_t)(B - 1);  // L for even L, L - 1 for odd L: irfft without a length
    if (bs > n_inv) return fail(c, DS_ERR_ARG, "ds_fir_ols_step: block longer than the inverse transform");
    CHK(reserve(c, &c->aux, &c->aux_bytes,
                2 * Carver::pad(sizeof(float2) * (size_t)B) + 2 * Carver::pad(sizeof(float) * (size_t)L) +
                    Carver::pad(sizeof(float))));
    Carver cv(c->aux);
    float2* X = cv.take<float2>(B);
    float2* Y = cv.take<float2>(B);
    float* full = cv.take<float>((size_t)L);
    float* roll = cv.take<float>((size_t)L);
    float* one = cv.take<float>(1);
    CHK(stream_ones(c, one, 1));
    hipLaunchKernelGGL(firstream::k_ols_put, dim3((bs + 255) / 256), dim3(256), 0, c->stream, row, block, L, bs);
    HIPCHK(c, hipGetLastError());
    CHK(ds_rfft_dev(c, row, 1, L, L, (int)L, 1.0f, (ds_c32*)X));
    hipLaunchKernelGGL(firstream::k_cmul, dim3((B + 255) / 256), dim3(256), 0, c->stream, (const float2*)X,
                       (const float2*)h, Y, B);
    HIPCHK(c, hipGetLastError());
    CHK(ds_deconv_dev(c, one, 1, 1, 1, 1, (int)n_inv, (const ds_c32*)Y, 0, n_inv, n_inv, full));
    hipLaunchKernelGGL(firstream::k_tail, dim3((bs + 255) / 256), dim3(256), 0, c->stream, (const float*)full, n_inv, bs,
                       1, out);
    hipLaunchKernelGGL(firstream::k_ols_roll, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, c->stream,
                       (const float*)row, roll, L, bs);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(row, roll, sizeof(float) * (size_t)L, hipMemcpyDeviceToDevice, c->stream));
    return DS_OK;
}

// FIR transfer functions at arbitrary frequencies (host pointers; complex128 taps and result)
extern "C" int ds_fir_freqz(ds_ctx* c, const double* taps, int n_filt, int n_taps, const double* freqs_hz, int n_freq,
                            double fs_hz, double* out) {
    if (!c || !taps || !freqs_hz || !out) return fail(c, DS_ERR_ARG, "ds_fir_freqz: null argument");
    if (n_filt <= 0 || n_taps <= 0 || n_freq <= 0 || !(fs_hz > 0.0)) return fail(c, DS_ERR_ARG, "ds_fir_freqz: bad shape");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bt = (size_t)n_filt * n_taps * 16, bf = (size_t)n_freq * 8, bo = (size_t)n_filt * n_freq * 16;
    CHK(reserve(c, &c->io, &c->io_bytes, Carver::pad(bt) + Carver::pad(bf) + Carver::pad(bo)));
    Carver cv(c->io);
    double2* dt = cv.take<double2>((size_t)n_filt * n_taps);
    double* df = cv.take<double>(n_freq);
    double2* dout = cv.take<double2>((size_t)n_filt * n_freq);
    HIPCHK(c, hipMemcpyAsync(dt, taps, bt, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(df, freqs_hz, bf, hipMemcpyHostToDevice, c->stream));
    freqz::Args a{dt, n_filt, n_taps, df, n_freq, fs_hz, dout};
    CHK(launch(c, "fir_freqz", freqz::k_freqz, dim3((n_freq + 255) / 256, n_filt), 256, 0, a));
    HIPCHK(c, hipMemcpyAsync(out, dout, bo, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DS_OK;
}

// ---- RCCL (resolved at run time so the library loads on machines without it) ----
typedef int (*nccl_getuid_t)(void*);
struct uid128 {
    char b[128];
};
typedef int (*nccl_init_rank_t)(void**, int, uid128, int);
typedef int (*nccl_bcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_destroy_t)(void*);
typedef const char* (*nccl_errstr_t)(int);

static void* rccl_handle() {
    static void* h = nullptr;
    // reuse a copy the host process already mapped (e.g. the one PyTorch links) before loading one
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW);
    if (!h) h = dlopen("librccl.so", RTLD_NOW);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW);
    return h;
}

extern "C" int ds_comm_unique_id(char id_out[128]) {
    if (!id_out) return fail(nullptr, DS_ERR_ARG, "ds_comm_unique_id: null");
    void* h = rccl_handle();
    if (!h) return fail(nullptr, DS_ERR_COMM, "librccl.so not found");
    auto f = (nccl_getuid_t)dlsym(h, "ncclGetUniqueId");
    if (!f) return fail(nullptr, DS_ERR_COMM, "ncclGetUniqueId missing");
    int r = f(id_out);
    return r == 0 ? DS_OK : fail(nullptr, DS_ERR_COMM, "ncclGetUniqueId failed");
}

extern "C" int ds_comm_init(ds_ctx* c, int n_ranks, int rank, const char id[128]) {
    if (!c || !id || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(c, DS_ERR_ARG, "ds_comm_init: bad argument");
    void* h = rccl_handle();
    if (!h) return fail(c, DS_ERR_COMM, "librccl.so not found");
    auto f = (nccl_init_rank_t)dlsym(h, "ncclCommInitRank");
    if (!f) return fail(c, DS_ERR_COMM, "ncclCommInitRank missing");
    HIPCHK(c, hipSetDevice(c->device));
    uid128 u;
    memcpy(u.b, id, 128);
    int r = f(&c->comm, n_ranks, u, rank);
    if (r != 0) {
        auto es = (nccl_errstr_t)dlsym(h, "ncclGetErrorString");
        return fail(c, DS_ERR_COMM, std::string("ncclCommInitRank: ") + (es ? es(r) : "error"));
    }
    c->rccl = h;
    return DS_OK;
}

// ranks of the library's communicator as RCCL itself counts them (ncclCommCount)
extern "C" int ds_comm_count(ds_ctx* c, int* n_ranks) {
    if (!c || !n_ranks) return fail(c, DS_ERR_ARG, "ds_comm_count: null argument");
    if (!c->comm) return fail(c, DS_ERR_COMM, "ds_comm_count: communicator not initialised");
    typedef int (*nccl_count_t)(void*, int*);
    auto f = (nccl_count_t)dlsym(c->rccl, "ncclCommCount");
    if (!f) return fail(c, DS_ERR_COMM, "ncclCommCount missing");
    if (f(c->comm, n_ranks) != 0) return fail(c, DS_ERR_COMM, "ncclCommCount failed");
    return DS_OK;
}

extern "C" int ds_bcast(ds_ctx* c, void* buf, size_t bytes, int root) {
    if (!c || !buf) return fail(c, DS_ERR_ARG, "ds_bcast: null argument");
    if (!c->comm) return fail(c, DS_ERR_COMM, "ds_bcast: communicator not initialised");
    auto f = (nccl_bcast_t)dlsym(c->rccl, "ncclBroadcast");
    if (!f) return fail(c, DS_ERR_COMM, "ncclBroadcast missing");
    int r = f(buf, buf, bytes, /*ncclChar*/ 0, root, c->comm, c->stream);
    if (r != 0) return fail(c, DS_ERR_COMM, "ncclBroadcast failed");
    return DS_OK;
}

extern "C" int ds_allgather(ds_ctx* c, const void* send, void* recv, size_t bytes_per_rank) {
    if (!c || !send || !recv) return fail(c, DS_ERR_ARG, "ds_allgather: null argument");
    if (!c->comm) return fail(c, DS_ERR_COMM, "ds_allgather: communicator not initialised");
    auto f = (nccl_allgather_t)dlsym(c->rccl, "ncclAllGather");
    if (!f) return fail(c, DS_ERR_COMM, "ncclAllGather missing");
    int r = f(send, recv, bytes_per_rank, /*ncclChar*/ 0, c->comm, c->stream);
    if (r != 0) return fail(c, DS_ERR_COMM, "ncclAllGather failed");
    return DS_OK;
}

// ---- measured copy bandwidth (the roofline's second denominator) --------------
__global__ __launch_bounds__(256) void k_copy16(const float4* __restrict__ src, float4* __restrict__ dst, size_t n16) {
    // four independent 16-byte loads per lane in flight before the first store
    const size_t stride = (size_t)gridDim.x * 1024;
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 768 < n16; i += stride) {
        const float4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
        dst[i] = a;
        dst[i + 256] = b;
        dst[i + 512] = c;
        dst[i + 768] = d;
    }
    if (i < n16) {  // the last, partly filled tile
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + 256 * k < n16) dst[i + 256 * k] = src[i + 256 * k];
    }
}
extern "C" int ds_measure_copy(ds_ctx* c, size_t bytes, int reps, double* gb_per_s) {
    if (!c || !gb_per_s || bytes < 16 || reps <= 0) return fail(c, DS_ERR_ARG, "ds_measure_copy: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    float4 *a = nullptr, *b = nullptr;
    const size_t n16 = bytes / 16;
    if (hipMalloc((void**)&a, n16 * 16) != hipSuccess) return fail(c, DS_ERR_NOMEM, "ds_measure_copy: hipMalloc");
    if (hipMalloc((void**)&b, n16 * 16) != hipSuccess) {
        (void)hipFree(a);
        return fail(c, DS_ERR_NOMEM, "ds_measure_copy: hipMalloc");
    }
    int rc = DS_OK;
    float ms = 0.f;
    const unsigned grid = (unsigned)std::min<size_t>((n16 + 1023) / 1024, 256 * 8);  // 8 workgroups per CU
    do {
        if (hipMemsetAsync(a, 1, n16 * 16, c->stream) != hipSuccess) { rc = fail(c, DS_ERR_HIP, "ds_measure_copy: memset"); break; }
        hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, c->stream, a, b, n16);
        if (hipEventRecord(c->ev0, c->stream) != hipSuccess) { rc = fail(c, DS_ERR_HIP, "ds_measure_copy: event"); break; }
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, c->stream, a, b, n16);
        if (hipEventRecord(c->ev1, c->stream) != hipSuccess || hipEventSynchronize(c->ev1) != hipSuccess ||
            hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess || hipGetLastError() != hipSuccess) {
            rc = fail(c, DS_ERR_HIP, "ds_measure_copy: timing");
            break;
        }
        *gb_per_s = 2.0 * (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9;
    } while (0);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(a);
    (void)hipFree(b);
    return rc;
}

extern "C" int ds_mem_info(ds_ctx* c, size_t* free_bytes, size_t* total_bytes) {
    if (!c || !free_bytes || !total_bytes) return fail(c, DS_ERR_ARG, "ds_mem_info: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemGetInfo(free_bytes, total_bytes));
    return DS_OK;
}

extern "C" int ds_comm_destroy(ds_ctx* c) {
    if (!c || !c->comm) return DS_OK;
    auto f = (nccl_destroy_t)dlsym(c->rccl, "ncclCommDestroy");
    if (f) f(c->comm);
    c->comm = nullptr;
    return DS_OK;
}
