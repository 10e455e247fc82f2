// Every DSPTOOLBOX_AMD_* switch of the library in one place.  ds_init reads the environment ONCE
// into the context (ds_ctx::cfg); no launch function calls getenv.  The route switches select
// another kernel family for the same result -- kept for A/B measurement and as fall-backs, and every
// one of them is a tested route (tests/test_gpu_parity.py::test_kernel_selecting_switches creates a
// context under each and runs the golden subset).  The tuning overrides change grid shapes only.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdlib>

struct ds_config {
    // ---- routes ---------------------------------------------------------------------------------
    bool welch_generic = false;   // DSPTOOLBOX_AMD_WELCH_GENERIC: generic LDS kernels instead of the register kernels
                                  //   for windows of 8 ... 16384 other than 4096 (and for cross spectra)
    bool no_welch4096 = false;    // DSPTOOLBOX_AMD_NO_WELCH4096=1: ... and for 4096-sample windows too
    bool w2048_wave = false;      // DSPTOOLBOX_AMD_W2048_WAVE=1: 2048-sample windows at 50 % overlap on the wave / team kernels
                                  //   (kernels_welch1024.hpp) instead of two transforms per pass of the 4096-point machine
    bool w4_two_per_cu = false;   // DSPTOOLBOX_AMD_W4_TWO_PER_CU=1: round 1's welch4096::k_y instead of k_y3
    bool stft_generic = false;    // DSPTOOLBOX_AMD_STFT_GENERIC: k_stft<N> instead of the wave / frame kernels
    bool istft_fused = true;      // DSPTOOLBOX_AMD_ISTFT_FUSED=0: transform and overlap-add as two launches
    bool istft_wave = true;       // DSPTOOLBOX_AMD_ISTFT_WAVE=0: no wave-level inverse transform (256 ... 2048)
    bool istft_one_ch = false;    // DSPTOOLBOX_AMD_ISTFT_CT=1: one channel per workgroup in k_istft
    bool csm_generic = false;     // DSPTOOLBOX_AMD_CSM_GENERIC: one workgroup per bin and tile pair
    int csm_chunks = 1;           // DSPTOOLBOX_AMD_CSM_CHUNKS = 2 ... 8: frame chunks of the <= 64-channel matrix (transform of chunk
                                  //   k + 1 beside the product of chunk k on a second stream).  Measured, not adopted: 0.22 / 0.24 /
                                  //   0.29 ms at 2 / 4 / 8 chunks against 0.16 ms for one transform, then one product
                                  //   (profiles/r05_csm_chunks_two_streams.txt)
    bool csm_f32 = false;         // DSPTOOLBOX_AMD_CSM_F32: fp32 matrix instructions instead of bf16 triples
    bool deconv_generic = false;  // DSPTOOLBOX_AMD_DECONV_GENERIC: k_deconv<8192> instead of deconv8k
    bool deconv_2percu = false;   // DSPTOOLBOX_AMD_DECONV_2PERCU: the 512-thread deconv8k kernel
    bool deconv_persist = true;   // DSPTOOLBOX_AMD_DECONV_PERSIST=0: one unit per workgroup (k_deconv3q / k_deconv3) instead of k_deconv_p
    bool deconv_4percu = true;    // DSPTOOLBOX_AMD_DECONV_4PERCU=0: k_deconv3 (three per CU) instead of k_deconv3q
    bool fir_generic = false;     // DSPTOOLBOX_AMD_FIR_GENERIC: k_fir<16384> instead of fir16k
    int fir4k_min_taps = 1025;    // DSPTOOLBOX_AMD_FIR_4K: 0 never fir4k, 1 always, n > 1 from n taps on
    bool fir_stage = false;       // DSPTOOLBOX_AMD_FIR_STAGE=1: fir4k's stores through a per-wave LDS strip (16-byte stores; measured, not faster)
    bool welch_long_3percu = false;  // DSPTOOLBOX_AMD_WELCH_LONG_3PERCU=1: the long-window cross loop (welchl::k_yc) without the next class sequence in
                                  //   flight through the transform: 152 instead of 174 registers, three workgroups per CU instead of two
    bool fir_3percu = true;       // DSPTOOLBOX_AMD_FIR_3PERCU=0: two-partition filters (2050 ... 4097 taps) on fir4k::k_fir<2> (two workgroups per CU,
                                  //   the next filter's tap spectra in flight through the transform) instead of k_fir3 (three per CU, the tap
                                  //   spectra requested at the top of each filter's pass: 1.66-1.68 against 1.74-1.76 ms on the bench shape)
    bool fir_direct = true;       // DSPTOOLBOX_AMD_FIR_DIRECT=0: no direct float64 sum for a signal shorter than the filter
    bool finish_wide = false;     // DSPTOOLBOX_AMD_FINISH_WIDE=1: k_welch_finish by 64-bit loads (the path of partial slabs >= 4 GiB)
    // ---- tuning overrides (0 = the built-in choice) ------------------------------------------------
    int stft_ct = 0, stft_fpw = 0, stft4k_chunks = 0, istft_fpw = 0;
    int welch_chunks = 0, welch1k_chunks = 0;
    int welch_long_min = 16384;   // DSPTOOLBOX_AMD_WELCH_LONG_MIN: smallest window (16384 ... 262144) on kernels_welch_long.hpp
                                  //   (32768: 16384-sample windows on kernels_welch16384.hpp, 0.45 against 0.36 ms on the headline shape)
    int fir_block = 0, fir_chunks = 0, fir_split = 0;
    size_t bluestein_cache_bytes = (size_t)256 << 20;  // DSPTOOLBOX_AMD_BLUESTEIN_CACHE_MB

    static ds_config from_env() {
        ds_config g;
        auto set = [](const char* name) { return getenv(name) != nullptr; };
        auto is = [](const char* name, char v) {
            const char* e = getenv(name);
            return e && e[0] == v;
        };
        auto num = [](const char* name) {
            const char* e = getenv(name);
            return e ? atoi(e) : 0;
        };
        g.welch_generic = set("DSPTOOLBOX_AMD_WELCH_GENERIC");
        g.no_welch4096 = is("DSPTOOLBOX_AMD_NO_WELCH4096", '1');
        g.w2048_wave = is("DSPTOOLBOX_AMD_W2048_WAVE", '1');
        g.w4_two_per_cu = is("DSPTOOLBOX_AMD_W4_TWO_PER_CU", '1');
        g.stft_generic = set("DSPTOOLBOX_AMD_STFT_GENERIC");
        g.istft_fused = !(set("DSPTOOLBOX_AMD_ISTFT_FUSED") && num("DSPTOOLBOX_AMD_ISTFT_FUSED") == 0);
        g.istft_wave = !(set("DSPTOOLBOX_AMD_ISTFT_WAVE") && num("DSPTOOLBOX_AMD_ISTFT_WAVE") == 0);
        g.istft_one_ch = set("DSPTOOLBOX_AMD_ISTFT_CT") && num("DSPTOOLBOX_AMD_ISTFT_CT") == 1;
        g.csm_generic = set("DSPTOOLBOX_AMD_CSM_GENERIC");
        g.csm_f32 = set("DSPTOOLBOX_AMD_CSM_F32");
        if (num("DSPTOOLBOX_AMD_CSM_CHUNKS") >= 1) g.csm_chunks = std::min(8, num("DSPTOOLBOX_AMD_CSM_CHUNKS"));
        g.deconv_generic = set("DSPTOOLBOX_AMD_DECONV_GENERIC");
        g.deconv_2percu = set("DSPTOOLBOX_AMD_DECONV_2PERCU");
        g.deconv_persist = !(set("DSPTOOLBOX_AMD_DECONV_PERSIST") && num("DSPTOOLBOX_AMD_DECONV_PERSIST") == 0);
        g.deconv_4percu = !(set("DSPTOOLBOX_AMD_DECONV_4PERCU") && num("DSPTOOLBOX_AMD_DECONV_4PERCU") == 0);
        g.finish_wide = is("DSPTOOLBOX_AMD_FINISH_WIDE", '1');
        g.fir_generic = set("DSPTOOLBOX_AMD_FIR_GENERIC");
        g.fir_stage = is("DSPTOOLBOX_AMD_FIR_STAGE", '1');
        g.fir_3percu = !(set("DSPTOOLBOX_AMD_FIR_3PERCU") && num("DSPTOOLBOX_AMD_FIR_3PERCU") == 0);
        g.welch_long_3percu = is("DSPTOOLBOX_AMD_WELCH_LONG_3PERCU", '1');
        g.fir_direct = !(set("DSPTOOLBOX_AMD_FIR_DIRECT") && num("DSPTOOLBOX_AMD_FIR_DIRECT") == 0);
        if (const char* e = getenv("DSPTOOLBOX_AMD_FIR_4K")) {
            if (e[0] == '0')
                g.fir4k_min_taps = 1 << 30;
            else if (e[0] == '1' && e[1] == 0)
                g.fir4k_min_taps = 1;
            else if (atoi(e) > 1)
                g.fir4k_min_taps = atoi(e);
        }
        g.stft_ct = num("DSPTOOLBOX_AMD_STFT_CT");
        g.stft_fpw = num("DSPTOOLBOX_AMD_STFT_FPW");
        g.stft4k_chunks = num("DSPTOOLBOX_AMD_STFT4K_CHUNKS");
        g.istft_fpw = num("DSPTOOLBOX_AMD_ISTFT_FPW");
        g.welch_chunks = num("DSPTOOLBOX_AMD_WELCH_CHUNKS");
        g.welch1k_chunks = num("DSPTOOLBOX_AMD_WELCH1K_CHUNKS");
        if (num("DSPTOOLBOX_AMD_WELCH_LONG_MIN") >= 16384) g.welch_long_min = num("DSPTOOLBOX_AMD_WELCH_LONG_MIN");
        g.fir_block = num("DSPTOOLBOX_AMD_FIR_BLOCK");
        g.fir_chunks = num("DSPTOOLBOX_AMD_FIR_CHUNKS");
        g.fir_split = num("DSPTOOLBOX_AMD_FIR_SPLIT");
        if (set("DSPTOOLBOX_AMD_BLUESTEIN_CACHE_MB"))
            g.bluestein_cache_bytes = (size_t)std::max(1, num("DSPTOOLBOX_AMD_BLUESTEIN_CACHE_MB")) << 20;
        return g;
    }
};
