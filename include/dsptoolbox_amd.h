/*
 * dsptoolbox_amd -- C-ABI of the MI355X (gfx950) spectral hot path.
 *
 * The reference (dsptoolbox 0.8) is pure Python and has no FFI; its "plugin
 * boundary" for this path is the ndarray-in / ndarray-out private backend layer
 * (SURVEY.md section 1, L2).  Each entry point below replaces one of those backend
 * functions; the Python host shim (dsptoolbox_amd/backend.py) binds them with
 * ctypes exactly as INTEGRATION.md shows.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; the message is
 *     available through ds_last_error().  Nothing throws across the ABI.
 *   - signals are PLANAR fp32: channel c, sample n at x[c*ld + n]
 *     (the reference stores (samples, channels) float64; the shim transposes).
 *   - outputs are written in the reference's axis order so the host only
 *     casts: (bins, channels), (bins, frames, channels), (bins, ch, ch).
 *   - `_dev` variants take DEVICE pointers (inputs already resident in HBM,
 *     nothing is copied, everything is enqueued on the context's stream and
 *     NOT synchronised).  The plain variants take HOST pointers, stage through
 *     the context's workspace and return after the result is in host memory.
 *   - a ds_ctx is bound to one device and one HIP stream; not re-entrant.
 */
#ifndef DSPTOOLBOX_AMD_H
#define DSPTOOLBOX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ds_ctx ds_ctx;
typedef struct { float re, im; } ds_c32;

/* error codes */
#define DS_OK            0
#define DS_ERR_ARG      -1   /* invalid argument (shape, size, null pointer)   */
#define DS_ERR_UNSUP    -2   /* valid in the reference, not built yet on GPU   */
#define DS_ERR_HIP      -3   /* HIP runtime error                              */
#define DS_ERR_NOMEM    -4
#define DS_ERR_COMM     -5   /* RCCL error                                     */

/* transfer-function modes: transfer_functions/enums.py:4-16 */
#define DS_TF_H1 1
#define DS_TF_H2 2
#define DS_TF_H3 3

/* Welch averaging over frames: _spectral_methods.py:151-162 */
#define DS_AVG_MEAN   0
#define DS_AVG_MEDIAN 1

/* filter-bank modes: standard/enums.py:279-292 */
#define DS_FB_PARALLEL   1
#define DS_FB_SEQUENTIAL 2
#define DS_FB_SUMMED     3

/* ---- context, memory, timing ------------------------------------------- */
int         ds_version(void);
int         ds_device_count(void);
int         ds_init(int device, ds_ctx** out);
void        ds_destroy(ds_ctx* ctx);
const char* ds_last_error(ds_ctx* ctx);            /* ctx may be NULL          */
int         ds_malloc(ds_ctx* ctx, void** dptr, size_t bytes);
int         ds_free(ds_ctx* ctx, void* dptr);
/* Page-locked host memory for result staging (device-resident signals): a download into
 * it runs at the link's rate and needs no driver-side bounce buffer.  Plumbing: nothing in the reference.     */
int         ds_host_alloc(ds_ctx* ctx, void** hptr, size_t bytes);
int         ds_host_free(ds_ctx* ctx, void* hptr);
int         ds_upload(ds_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int         ds_download(ds_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int         ds_memset(ds_ctx* ctx, void* dst_dev, int value, size_t bytes);
int         ds_sync(ds_ctx* ctx);
/* hipEvent pair recorded on the context's stream (bench.py's live kernel time) */
int         ds_timer_start(ds_ctx* ctx);
int         ds_timer_stop(ds_ctx* ctx, float* elapsed_ms);
/* per-kernel HIP-event timing on the context's stream: while enabled every
 * kernel launch is bracketed by an event pair; ds_profile_report synchronises
 * and returns "name total_ms launches\n" lines for the launches since the
 * previous report (string owned by the context).                             */
int         ds_profile_enable(ds_ctx* ctx, int on);
const char* ds_profile_report(ds_ctx* ctx);
/* restrict the bracketing to one kernel name (NULL or "": all kernels): every
 * event pair costs ~3 us of stream time, which matters for 10 us kernels      */
int         ds_profile_only(ds_ctx* ctx, const char* kernel_name);
/* bracket only every `every`-th matching launch (1 = all): an event pair costs ~3 us of stream time */
int         ds_profile_stride(ds_ctx* ctx, int every);
/* what an event pair adds to the kernels it brackets: n_kernels empty kernels bracketed the same
 * way, average over reps, in ms.  b1 (one kernel) and b2 (two) give 2 b1 - b2, a lower bound of the
 * bracket's fixed cost (bench.py reports kernel times with and without it) */
int         ds_profile_overhead(ds_ctx* ctx, int reps, int n_kernels, double* ms);
/* which kernel family ran: the launch names since the previous call, space separated, each "group" or
 * "group@variant" (e.g. "csm_gemm@b3", "fir@4k_p2"; string owned by the context).  The library reads its
 * DSPTOOLBOX_AMD_* switches once, in ds_init (csrc/config.hpp); tests open a context under a switch and
 * check here that the other family really computed the result.                                          */
const char* ds_routes(ds_ctx* ctx);
/* max FFT length one workgroup transforms inside LDS (complex points)       */
int         ds_max_fft_len(void);

/* ---- STFT: replaces _stft, standard/_spectral_methods.py:176-282 --------
 * frame k of channel c = samples [k*hop - pad_front, k*hop - pad_front + W)
 * (out-of-range samples read as 0, which is the reference's zero padding:
 * helpers/other.py:181-213), times window[W]; optional detrend (mean of the
 * windowed frame, :264-265); rFFT of length nfft (crop / zero-pad, :268);
 * out[b][k][c] *= scale, bins 0 and nfft/2 additionally *= edge_scale;
 * power != 0 stores |.|^2 in .re (:271-278).  B = nfft/2 + 1.               */
int ds_stft_r2c_dev(ds_ctx* ctx, const float* x_dev, int64_t n_samples, int n_ch,
                    int64_t ld, int W, int hop, int nfft, int64_t pad_front,
                    int n_frames, const float* window_dev, int detrend,
                    float scale, float edge_scale, int power, ds_c32* out_dev);
int ds_stft_r2c(ds_ctx* ctx, const float* x, int64_t n_samples, int n_ch,
                int W, int hop, int nfft, int64_t pad_front, int n_frames,
                const float* window, int detrend, float scale, float edge_scale,
                int power, ds_c32* out);

/* ---- Welch spectra: replaces _welch, _spectral_methods.py:10-173 ---------
 * Raw frame averages Sxx=mean|X|^2, Syy=mean|Y|^2, Sxy=mean conj(X)Y over
 * n_frames frames (frame k = samples [k*hop, k*hop+W), zero padded), then the
 * reference's finish(): S *= norm_scale (1, 1/W^2, 1/W for the three FFT
 * norms, enums.py:53-75); if halve_edges: S *= factor and bins 0, W/2 halved
 * (:165-168); if amp_sqrt: principal sqrt (:170-171).  average = DS_AVG_MEDIAN takes the
 * per-bin median over frames (real and imaginary parts separately) times the
 * reference's bias n (:153-162) instead of the mean.
 *
 * ds_welch_tf: replaces compute_transfer_function,
 * transfer_functions/transfer_functions.py:419-539.  n_cx is 1 (one input for
 * every output channel) or n_cy (pairwise).  tf[b][c] (B x n_cy), coh[b][c].
 *
 * ds_welch_psd: auto spectra of every channel of x (psd[b][c], B x n_cx) --
 * Signal.get_spectrum with the Welch method, classes/signal.py:881-897.      */
int ds_welch_tf_dev(ds_ctx* ctx, const float* x_dev, int n_cx, int64_t ldx,
                    const float* y_dev, int n_cy, int64_t ldy, int64_t n_samples,
                    int W, int hop, int n_frames, const float* window_dev,
                    int detrend, int average, int mode, int amp_sqrt, double norm_scale,
                    double factor, int halve_edges, ds_c32* tf_dev, float* coh_dev);
int ds_welch_tf(ds_ctx* ctx, const float* x, int n_cx, const float* y, int n_cy,
                int64_t n_samples, int W, int hop, int n_frames, const float* window,
                int detrend, int average, int mode, int amp_sqrt, double norm_scale,
                double factor, int halve_edges, ds_c32* tf, float* coh);
/* the same with the reference's array layout at the boundary: x (n_samples, n_cx), y (n_samples,
 * n_cy) float64 C-order (classes/signal.py:222-301).  The cast + transpose to planar float32 runs
 * on host threads straight into pinned chunks whose DMA overlaps the next chunk's cast.        */
/* float64 on both sides (the copy back runs through the same pinned chunks, widened / interleaved by
 * host threads while the next chunk's DMA is in flight): STFT (bins, frames, channels) complex128
 * as interleaved doubles; FIR y (bands or 1, n_samples, n_ch) float64.                            */
int ds_stft_r2c_f64(ds_ctx* ctx, const double* x, int64_t n_samples, int n_ch, int W, int hop,
                    int nfft, int64_t pad_front, int n_frames, const float* window, int detrend,
                    float scale, float edge_scale, int power, double* out_c128);
/* the same for whole-signal spectra, one-item spectral division and the inverse STFT (round 4): float64 /
 * complex128 arrays in the reference's own layouts on both sides, cast in host threads through pinned chunks
 * while the previous chunk's copy is in flight.  spec (bins, channels) complex128; ir (n_out, channels) float64;
 * stft (bins, frames, channels) complex128 in, out (total_length, channels) float64.                          */
int ds_rfft_f64(ds_ctx* ctx, const double* x, int n_ch, int64_t n_samples, int n_fft, float scale,
                double* spec_c128);
int ds_deconv_f64(ds_ctx* ctx, const double* y, int n_ch, int64_t n_samples, int n_fft, const ds_c32* r,
                  int r_per_channel, int64_t n_out, double* ir);
int ds_istft_f64(ds_ctx* ctx, const double* stft_c128, int n_bins, int n_frames, int n_ch, int nfft, int W,
                 int step, int frame_offset, int n_frames_total, const float* window, float scale,
                 int64_t total_length, double* out);
int ds_fir_ola_f64(ds_ctx* ctx, const double* x, int n_ch, int64_t n_samples, const float* taps,
                   int n_filt, int n_taps, int mode, double* y);
int ds_welch_psd_f64(ds_ctx* ctx, const double* x, int n_cx, int64_t n_samples, int W, int hop,
                     int n_frames, const float* window, int detrend, int average, int amp_sqrt,
                     double norm_scale, double factor, int halve_edges, float* psd);
int ds_csm_f64(ds_ctx* ctx, const double* x, int n_ch, int64_t n_samples, int W, int hop,
               int n_frames, const float* window, int detrend, int average, int amp_sqrt,
               double norm_scale, double factor, int halve_edges, ds_c32* csm);
int ds_welch_tf_f64(ds_ctx* ctx, const double* x, int n_cx, const double* y, int n_cy,
                    int64_t n_samples, int W, int hop, int n_frames, const float* window,
                    int detrend, int average, int mode, int amp_sqrt, double norm_scale,
                    double factor, int halve_edges, ds_c32* tf, float* coh);
/* The same estimate in float64 END TO END (transforms, sums, finish) for small or ill-conditioned
 * problems: x (n_samples, n_cx), y (n_samples, n_cy) float64 C-order exactly as the reference
 * holds them, float64 window, mean or median averaging (median: at most 4096 frames), W a power
 * of two <= 262144 (16384: as the 8192-point complex transform of the even / odd samples; 2^15 ... 2^18: one
 * decimation-in-frequency stage in front of that transform); tf[b][c] complex128
 * (interleaved re, im), coh[b][c] float64.  With fp32 transforms every frame's rounding floor
 * (1e-7 of its peak) lands on all bins, so bins 80 dB down -- the top of a fast pink sweep,
 * BASELINE config 1 -- are only good to 1e-4; this route keeps the reference's 1e-12.        */
int ds_welch_tf_x64(ds_ctx* ctx, const double* x, int n_cx, const double* y, int n_cy,
                    int64_t n_samples, int W, int hop, int n_frames, const double* window,
                    int detrend, int average, int mode, int amp_sqrt, double norm_scale,
                    double factor, int halve_edges, double* tf, double* coh);
/* _welch itself in float64 end to end (_spectral_methods.py:10-173): auto spectra (y NULL) or the cross
 * spectra conj(X_i) Y_i of channel pairs, mean or median averaging; x, y (n_samples, n_ch) float64 C order,
 * out [nb][n_ch] complex128 (auto spectra: imaginary part 0).  What the host mirror takes for SHORT
 * estimates (fewer than 128 frames: no averaging-down of the fp32 transform rounding).                     */
int ds_welch_spec_x64(ds_ctx* ctx, const double* x, const double* y, int n_ch, int64_t n_samples, int W,
                      int hop, int n_frames, const double* window, int detrend, int average, int amp_sqrt,
                      double norm_scale, double factor, int halve_edges, double* out);
/* _csm_welch in float64 end to end (_spectral_methods.py:285-371; up to 1024 channels; average = median, the
 * per-pair median of _welch :153-162, up to 128 frames): csm [nb][n_ch][n_ch] complex128, same element order as ds_csm. */
int ds_csm_x64(ds_ctx* ctx, const double* x, int n_ch, int64_t n_samples, int W, int hop, int n_frames,
               const double* window, int detrend, int average, int amp_sqrt, double norm_scale, double factor,
               int halve_edges, double* csm);
int ds_welch_psd_dev(ds_ctx* ctx, const float* x_dev, int n_cx, int64_t ldx,
                     int64_t n_samples, int W, int hop, int n_frames,
                     const float* window_dev, int detrend, int average, int amp_sqrt,
                     double norm_scale, double factor, int halve_edges, float* psd_dev);
int ds_welch_psd(ds_ctx* ctx, const float* x, int n_cx, int64_t n_samples, int W,
                 int hop, int n_frames, const float* window, int detrend, int average,
                 int amp_sqrt, double norm_scale, double factor, int halve_edges,
                 float* psd);
/* cross spectrum of channel pairs (x[c], y[c]) -- _welch(x, y): csd[b][c]     */
int ds_welch_csd(ds_ctx* ctx, const float* x, const float* y, int n_ch,
                 int64_t n_samples, int W, int hop, int n_frames, const float* window,
                 int detrend, int average, int amp_sqrt, double norm_scale, double factor,
                 int halve_edges, ds_c32* csd);
/* ... taking the reference's (n_samples, n_ch) float64 C-order arrays as they are (cast + transpose in host threads
 * straight into page-locked upload chunks, like ds_welch_psd_f64 / ds_welch_tf_f64)                                  */
int ds_welch_csd_f64(ds_ctx* ctx, const double* x, const double* y, int n_ch,
                     int64_t n_samples, int W, int hop, int n_frames, const float* window,
                     int detrend, int average, int amp_sqrt, double norm_scale, double factor,
                     int halve_edges, ds_c32* csd);

/* ---- band powers of a spectrogram: replaces np.tensordot(mel_filters, |stft|^2) + to_db in
 * log_mel_spectrogram / mfcc, transforms/transforms.py:181-184, 421-429.
 * stft[b][fc] (n_bins x n_fc, n_fc = frames * channels, the layout ds_stft_r2c writes),
 * weights[band][b]; only bins band_start[band] <= b < band_stop[band] are read (the
 * filters are banded).  out[band][fc] = sum_b w |X|^2, then 10 log10(max(., DBL_MIN)) if
 * to_db, then |DCT-II along the band axis| with NaN -> 0 if dct_abs (MFCC).          */
int ds_band_power_dev(ds_ctx* ctx, const ds_c32* stft_dev, int n_bins, int64_t n_fc,
                      const float* weights_dev, const int* band_start_dev,
                      const int* band_stop_dev, int n_bands, int to_db, int dct_abs,
                      float* out_dev);
int ds_band_power(ds_ctx* ctx, const ds_c32* stft, int n_bins, int64_t n_fc, const float* weights,
                  const int* band_start, const int* band_stop, int n_bands, int to_db,
                  int dct_abs, float* out);

/* ---- delay-and-sum beamformer map on the CSM: replaces the grid x bin loop of
 * BeamformerDASFrequency.get_beamformer_map, beamforming/beamforming.py:853-858:
 * map[g][f] = Re( h_f[:, g]^H  CSM_f  h_f[:, g] ); csm[f][i][j] (n_bins x n_ch x n_ch, the
 * selected bins, diagonal already treated by the caller), h[f][c][g] (the steering
 * vectors, n_bins x n_ch x n_grid), map[g][f].  fp32 MFMA.                        */
int ds_das_map_dev(ds_ctx* ctx, const ds_c32* csm_dev, const ds_c32* h_dev, int n_bins, int n_ch,
                   int n_grid, float* map_dev);
int ds_das_map(ds_ctx* ctx, const ds_c32* csm, const ds_c32* h, int n_bins, int n_ch, int n_grid,
               float* map);
/* The diagonal treatment in front of the map, on the device (beamforming.py:840-845:
 * `csm *= C / (C - 1)` then `np.fill_diagonal(csm[i], 0)` for every bin): out = scale * csm, the
 * diagonal zeroed when zero_diagonal != 0.  csm_dev / out_dev [n_bins][n_ch][n_ch]; out_dev may be
 * csm_dev.  Lets a CSM that ds_csm_dev left in HBM feed ds_das_map_dev without a host round trip.  */
int ds_csm_das_prepare_dev(ds_ctx* ctx, const ds_c32* csm_dev, int n_bins, int n_ch, double scale,
                           int zero_diagonal, ds_c32* out_dev);

/* ---- inverse STFT: replaces transforms.istft, transforms/transforms.py:444-586
 * (np.fft.irfft of every frame + _reconstruct_framed_signal,
 * standard/_framed_signal_representation.py:70-137: windowed overlap-add divided by
 * the squared-window envelope clipped at 1e-4).
 * stft[b][f][c] (n_bins x n_frames x n_ch, the layout ds_stft_r2c writes); frames are
 * inverse transformed with length nfft (bins beyond nfft/2 dropped, missing ones zero),
 * scaled by `scale` (the irfft normalisation divided by the physical-unit factor),
 * cropped to W samples and windowed; frame f sits at sample (f + frame_offset)*step of
 * an output of total_length samples whose envelope counts n_frames_total window
 * positions (the reference adds an empty frame before and after unpadded data:
 * frame_offset 1, n_frames_total n_frames + 2).  out[c][n], planar.             */
int ds_istft_dev(ds_ctx* ctx, const ds_c32* stft_dev, int n_bins, int n_frames, int n_ch,
                 int nfft, int W, int step, int frame_offset, int n_frames_total,
                 const float* window_dev, float scale, int64_t total_length,
                 float* out_dev, int64_t ld_out);
int ds_istft(ds_ctx* ctx, const ds_c32* stft, int n_bins, int n_frames, int n_ch, int nfft,
             int W, int step, int frame_offset, int n_frames_total, const float* window,
             float scale, int64_t total_length, float* out);

/* ---- cross-spectral matrix: replaces _csm_welch, _spectral_methods.py:285-371
 * csm[b][i][j], B x C x C; lower triangle csm[b][i2][i1] (i2>=i1) =
 * finish(mean or median over frames of conj(X_i1) X_i2), upper = its conjugate
 * (:351-369).  average: DS_AVG_MEAN (MFMA rank-F update) / DS_AVG_MEDIAN.      */
int ds_csm_dev(ds_ctx* ctx, const float* x_dev, int n_ch, int64_t ld, int64_t n_samples,
               int W, int hop, int n_frames, const float* window_dev, int detrend,
               int average, int amp_sqrt, double norm_scale, double factor,
               int halve_edges, ds_c32* csm_dev);
int ds_csm(ds_ctx* ctx, const float* x, int n_ch, int64_t n_samples, int W, int hop,
           int n_frames, const float* window, int detrend, int average, int amp_sqrt,
           double norm_scale, double factor, int halve_edges, ds_c32* csm);
/* bins [bin_start, bin_start + bin_count) only (csm_dev[0] is the matrix of bin_start, mean
 * averaging): the multi-GPU split of the CSM -- every rank transforms all channels and keeps
 * its own bin range, no reduction between ranks.                                         */
int ds_csm_bins_dev(ds_ctx* ctx, const float* x_dev, int n_ch, int64_t ld, int64_t n_samples,
                    int W, int hop, int n_frames, const float* window_dev, int detrend,
                    int amp_sqrt, double norm_scale, double factor, int halve_edges,
                    int bin_start, int bin_count, ds_c32* csm_dev);

/* CSM from spectra that are already on hand: X[b][f][c] (n_bins x n_frames x n_ch, the
 * STFT layout) -> csm[b][i][j] = finish(norm_scale/n_frames * sum_f X_i conj X_j).
 * n_frames = 1 replaces _csm_fft, _spectral_methods.py:374-443 (outer product of one
 * whole-signal spectrum; FFTBackward: no finish, others: edges halved, factor, sqrt).     */
int ds_csm_spec_dev(ds_ctx* ctx, const ds_c32* X_dev, int n_bins, int n_frames, int n_ch,
                    int amp_sqrt, double norm_scale, double factor, int halve_edges,
                    ds_c32* csm_dev);
int ds_csm_spec(ds_ctx* ctx, const ds_c32* X, int n_bins, int n_frames, int n_ch, int amp_sqrt,
                double norm_scale, double factor, int halve_edges, ds_c32* csm);

/* ---- whole-signal rFFT / regularised spectral division -------------------
 * ds_rfft: Signal.get_spectrum with SpectrumMethod.FFT, classes/signal.py:899-911
 * (any n_fft >= 2: one-workgroup LDS FFT up to ds_max_fft_len(), four-step FFT for powers
 * of two up to 2^24, Bluestein for every other length up to 2^23; input zero padded).
 * spec[b][c], (n_fft/2+1) x n_ch, multiplied by scale.
 *
 * ds_deconv: replaces _spectral_deconvolve,
 * transfer_functions/_transfer_functions.py:19-42, for a batch of n_items
 * signals of n_ch channels each (planar: y[(item*n_ch + c)*ld + n]).
 * r[(B) or (n_ch x B)] is the regularised inverse conj(X)/(|X|^2+eps) (or
 * 1/X) built by ds_deconv_inverse; ir = irfft(rfft(y, n_fft) * r, n_fft),
 * first n_out samples stored.                                                */
int ds_rfft_dev(ds_ctx* ctx, const float* x_dev, int n_ch, int64_t ld, int64_t n_samples,
                int n_fft, float scale, ds_c32* spec_dev);
int ds_rfft(ds_ctx* ctx, const float* x, int n_ch, int64_t n_samples, int n_fft,
            float scale, ds_c32* spec);
/* r[c][b] = eps ? conj(X)/(|X|^2+eps[b]) : 1/X   (eps_dev may be NULL)        */
int ds_deconv_inverse_dev(ds_ctx* ctx, const ds_c32* xspec_dev /*B x n_ch*/, int n_ch,
                          int n_bins, const float* eps_dev, ds_c32* r_dev /*n_ch x B*/);
int ds_deconv_dev(ds_ctx* ctx, const float* y_dev, int n_items, int n_ch, int64_t ld,
                  int64_t n_samples, int n_fft, const ds_c32* r_dev, int r_per_channel,
                  int64_t n_out, int64_t ld_out, float* ir_dev);
int ds_deconv(ds_ctx* ctx, const float* y, int n_items, int n_ch, int64_t n_samples,
              int n_fft, const ds_c32* r, int r_per_channel, int64_t n_out, float* ir);

/* ---- FIR filtering by FFT block convolution: replaces _lfilter_fir,
 * classes/filter_helpers.py:454-503 (scipy.signal.oaconvolve(...)[:N]) and the
 * filter loop of _filterbank_on_signal, :385-451.
 * y = (x * taps_k)[0:N] for each of n_filt filters of n_taps taps
 * (taps[k*n_taps + t]); DS_FB_PARALLEL: y[(k*n_ch + c)*ld_y + n];
 * DS_FB_SUMMED / DS_FB_SEQUENTIAL: y[c*ld_y + n].
 * Up to 8193 taps: LDS-resident blocks of <= 16384 points; longer filters (to
 * 2^23 + 1 taps): overlap-save on the four-step FFT.  Filter state (zi) and
 * zero-phase filtering are built on this call by the host shim: the full
 * convolution is this call over the signal followed by n_taps - 1 zeros.      */
int ds_fir_ola_dev(ds_ctx* ctx, const float* x_dev, int n_ch, int64_t ldx,
                   int64_t n_samples, const float* taps_dev, int n_filt, int n_taps,
                   int mode, float* y_dev, int64_t ld_y);
int ds_fir_ola(ds_ctx* ctx, const float* x, int n_ch, int64_t n_samples,
               const float* taps, int n_filt, int n_taps, int mode, float* y);

/* ---- block-streaming FIR classes with device-resident state ------------------------------
 * One process_block of the reference's real-time classes (classes/fir_filter_realtime.py:75-335),
 * executed literally on buffers that stay on the device between calls; per call only the block
 * goes up and the filtered block comes back.
 * ds_fir_part_step_dev: FIRUniformPartitioned (:206-240) / FIRUniformPartitionedMultichannel
 *   (:296-335) for the channels [ch0, ch0 + n_call): shift the 2*bs input buffers, transform,
 *   store the spectra at slot `ind` of the delay line S[b][P][C], accumulate
 *   sum_p H[b][p][ch or 0] * S[b][(ind - p) mod P][ch], inverse transform, return the last bs
 *   samples.  inbuf [C][2 bs], block [n_call][bs], H [bs + 1][P][Cf] (Cf = 1: one impulse
 *   response for all channels, else Cf = C), out [n_call][bs].  The caller advances `ind`.
 * ds_fir_ols_step_dev: FIRFilterOverlapSave.process_block (:120-142) for ONE channel: buffer row
 *   [L] (L = next_fast_len(T + bs), any length), H [L / 2 + 1]; the inverse transform has the
 *   length numpy's irfft picks without an argument, 2 (L / 2) -- so for odd L the block is NOT
 *   the convolution, exactly as in the reference; the buffer is rolled by bs afterwards.     */
int ds_fir_part_step_dev(ds_ctx* ctx, float* inbuf_dev, const float* block_dev, int bs, int n_ch,
                         int ch0, int n_call, const ds_c32* h_dev, int n_part, int n_fir_ch,
                         ds_c32* delay_dev, int ind, float* out_dev);
int ds_fir_ols_step_dev(ds_ctx* ctx, float* buffer_row_dev, const float* block_dev, int bs,
                        int64_t total_length, const ds_c32* h_dev, float* out_dev);

/* ---- host marshalling (no device work): the reference hands (samples, channels) float64
 * C-order arrays (classes/signal.py:222-301) and expects the same back; the kernels take planar
 * float32.  Multi-threaded cast + transpose on the host (numpy's strided cast takes 0.2 s for the
 * 537 MB of the headline shape; this takes a fraction of it).  threads <= 0: min(16, cores).  */
int ds_host_planar_f32(const double* src, int64_t n_samples, int n_ch, float* dst, int64_t ld,
                       int threads);   /* dst[c*ld + n] = (float)src[n*n_ch + c] */
int ds_host_interleave_f64(const float* src, int64_t n_samples, int n_ch, int64_t ld, double* dst,
                           int threads); /* dst[n*n_ch + c] = (double)src[c*ld + n] */
int ds_host_widen_f64(const float* src, int64_t n, double* dst, int threads); /* dst[i] = (double)src[i] */

/* ---- FIR transfer functions: replaces scipy.signal.freqz(b, 1, worN=f, fs) in Filter.get_transfer_function
 * (classes/filter.py:862-900) and its per-filter loop in FilterBank.get_transfer_function
 * (classes/filterbank.py:615-655): out[k][i] = sum_n taps[k][n] exp(-2 pi i freqs_hz[i] n / fs_hz), float64.
 * taps [n_filt][n_taps] and out [n_filt][n_freq] are complex128 (interleaved doubles), host pointers.   */
int ds_fir_freqz(ds_ctx* ctx, const double* taps, int n_filt, int n_taps, const double* freqs_hz,
                 int n_freq, double fs_hz, double* out);

/* ---- multi-GPU: RCCL over xGMI, one process per GPU -----------------------
 * The hot path has no exchange step: the only collectives are the
 * broadcast of a shared input (sweep / taps / inverse spectrum) and the gather of sharded
 * results.  Rank 0 creates the id, the host exchange (dsptoolbox_amd/rendezvous.py: a TCP
 * star; or a file, MPI, a torch.distributed store ...) hands the 128 bytes to every rank.
 * ds_allgather: rank r's bytes_per_rank bytes at send_dev land at recv_dev + r * bytes_per_rank
 * on every rank (ncclAllGather on the context's stream; send may alias its own slot).  */
int ds_comm_unique_id(char id_out[128]);
int ds_comm_init(ds_ctx* ctx, int n_ranks, int rank, const char id[128]);
/* ranks of that communicator as RCCL counts them (ncclCommCount): what a multi-GPU bench line reports */
int ds_comm_count(ds_ctx* ctx, int* n_ranks);
int ds_bcast(ds_ctx* ctx, void* buf_dev, size_t bytes, int root);
int ds_allgather(ds_ctx* ctx, const void* send_dev, void* recv_dev, size_t bytes_per_rank);
int ds_comm_destroy(ds_ctx* ctx);

/* ---- measurement: device-to-device copy bandwidth of THIS GPU (a 16-byte-per-lane streaming
 * copy kernel over `bytes` bytes, `reps` timed launches after one warm-up; GB/s counts read +
 * written bytes) -- the measured denominator bench.py reports next to the nominal 8 TB/s
 * (SURVEY.md section 8(d)).                                                                  */
int ds_measure_copy(ds_ctx* ctx, size_t bytes, int reps, double* gb_per_s);
/* free / total device memory of the context's GPU (hipMemGetInfo), for leak checks */
int ds_mem_info(ds_ctx* ctx, size_t* free_bytes, size_t* total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DSPTOOLBOX_AMD_H */
