"""Host shim: the reference's private numeric backends, same
names and argument meaning, executed by the HIP library through ctypes.

    _welch        <- dsptoolbox/standard/_spectral_methods.py:10-173
    _stft         <- dsptoolbox/standard/_spectral_methods.py:176-282
    _csm_welch    <- dsptoolbox/standard/_spectral_methods.py:285-371
    _lfilter_fir  <- dsptoolbox/classes/filter_helpers.py:454-503
    welch_transfer_function  <- the per-channel _welch loop of
                     transfer_functions/transfer_functions.py:476-534, fused
    rfft_spectrum <- scipy.fft.rfft call of classes/signal.py:899-911
    spectral_division <- transfer_functions/_transfer_functions.py:19-42

Arrays cross the boundary as (samples, channels) float64 like in the
reference; the shim transposes to planar fp32, the device computes in
fp32/complex64 (finish() in fp64) and results are cast back to
float64/complex128.  Anything the device path does not implement raises
NotImplementedError -- nothing is silently computed on the CPU.
"""

from __future__ import annotations

import ctypes as C
import os
from warnings import warn

import numpy as np
from scipy.signal import check_COLA
from scipy.signal.windows import get_window

from ._lib import DeviceBuffer, DevicePlanar, get_context, load_library
from .standard.enums import SpectrumScaling, Window

DS_TF = {"H1": 1, "H2": 2, "H3": 3}
DS_AVG = {"mean": 0, "median": 1}
DS_FB_PARALLEL, DS_FB_SEQUENTIAL, DS_FB_SUMMED = 1, 2, 3


def _planar_f32(x: np.ndarray) -> np.ndarray:
    """(N, C) float64 -> (C, N) float32 C-contiguous.  Large C-order float64 arrays go through the
    library's threaded cast + transpose (numpy's strided cast takes 0.2 s for the 537 MB of the
    headline shape); everything else through numpy."""
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    if x.ndim == 2 and x.dtype == np.float64 and x.flags.c_contiguous and x.size >= (1 << 20):
        lib = load_library()
        out = np.empty((x.shape[1], x.shape[0]), dtype=np.float32)
        if lib.ds_host_planar_f32(_ptr(x), x.shape[0], x.shape[1], _ptr(out), x.shape[0], 0) == 0:
            return out
    return np.ascontiguousarray(x.T, dtype=np.float32)


def _interleaved_f64(planar: np.ndarray, dst: np.ndarray | None = None) -> np.ndarray:
    """(C, N) float32 C-contiguous -> (N, C) float64 (the reverse of _planar_f32), into dst if given."""
    n_ch, n = planar.shape
    if dst is None:
        dst = np.empty((n, n_ch), dtype=np.float64)
    if planar.dtype == np.float32 and planar.flags.c_contiguous and dst.flags.c_contiguous \
            and planar.size >= (1 << 20):
        if load_library().ds_host_interleave_f64(_ptr(planar), n, n_ch, n, _ptr(dst), 0) == 0:
            return dst
    dst[...] = planar.T
    return dst


def _widen(a: np.ndarray) -> np.ndarray:
    """float32 -> float64 / complex64 -> complex128 of a C-contiguous array (threaded for large ones)."""
    wide = np.complex128 if a.dtype == np.complex64 else np.float64
    if a.dtype in (np.float32, np.complex64) and a.flags.c_contiguous and a.size >= (1 << 14):
        out = np.empty(a.shape, dtype=wide)
        n = a.size * (2 if a.dtype == np.complex64 else 1)
        # (mid-size arrays -- the spectra of a device-resident estimate -- in the calling thread: numpy's astype
        # takes 80 us for 2049 x 64 complex values, this loop 25)
        if load_library().ds_host_widen_f64(_ptr(a), n, _ptr(out), 0 if a.size >= (1 << 20) else 1) == 0:
            return out
    return a.astype(wide)


def _fusable(a) -> bool:
    """Large (N, C) float64 C-order array: crosses the boundary as it is (the *_f64 entry points cast
    + transpose it in host threads straight into pinned upload chunks)."""
    return isinstance(a, np.ndarray) and a.ndim == 2 and a.dtype == np.float64 and a.flags.c_contiguous \
        and a.size >= (1 << 20)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


_WINDOWS: dict = {}


def _window_array(window_type, length: int) -> np.ndarray:
    """scipy.signal.get_window(spec, length, fftbins=True), computed once per (spec, length): the array is handed out
    read-only (55 us for 4096 Hann samples is a quarter of a device-resident transfer-function call)."""
    spec = window_type.to_scipy_format() if isinstance(window_type, Window) else window_type
    try:
        key = (spec if not isinstance(spec, list) else tuple(spec), int(length))
        hash(key)
    except TypeError:
        return get_window(spec, length, fftbins=True)
    w = _WINDOWS.get(key)
    if w is None:
        if len(_WINDOWS) >= 64:
            _WINDOWS.pop(next(iter(_WINDOWS)))
        w = _WINDOWS[key] = np.array(get_window(spec, length, fftbins=True))  # (scipy hands out a view: own the data)
        w.setflags(write=False)
    return w


def _finish_params(scaling: SpectrumScaling, W: int, fs_hz: int, window: np.ndarray):
    """(amp_sqrt, norm_scale, factor, halve_edges) of _welch's tail (:141-171)."""
    norm = scaling.fft_norm()
    norm_scale = 1.0 if norm == "backward" else (1.0 / W**2 if norm == "forward" else 1.0 / W)
    phys = scaling.has_physical_units()
    factor = float(np.asarray(scaling.get_scaling_factor(W, fs_hz, window)).ravel()[0]) if phys else 1.0
    return int(scaling.is_amplitude_scaling()), float(norm_scale), factor, int(phys)


def _welch_checks(window_length_samples, overlap_percent, average):
    assert window_length_samples in [2**k for k in range(3, 19)], (
        "Window length should be a power of 2 between [8, 262_144] or [2**3, 2**18]")
    assert overlap_percent >= 0 and overlap_percent < 100, \
        "overlap_percent should be between 0 and 100"
    assert average in ("mean", "median"), f"{average} is not valid. Use either mean or median"


_COLA: dict = {}


def _window_key(window: np.ndarray):
    """Identity of one of _window_array's read-only arrays (they live as long as the cache), content hash otherwise."""
    if not window.flags.writeable and window.base is None:
        return ("id", id(window), window.size)
    return ("bytes", window.size, hash(np.ascontiguousarray(window).tobytes()))


def _cola_ok(window: np.ndarray, overlap: int) -> bool:
    """scipy.signal.check_COLA, remembered per (window, overlap)."""
    key = (int(overlap), _window_key(window))
    ok = _COLA.get(key)
    if ok is None:
        if len(_COLA) >= 256:
            _COLA.clear()
        ok = _COLA[key] = bool(check_COLA(window, nperseg=len(window), noverlap=overlap))
    return ok


def _welch_framing(n_samples: int, W: int, overlap_percent: float, window: np.ndarray):
    overlap = int(overlap_percent / 100 * W)  # truncation, _spectral_methods.py:106
    hop = W - overlap
    if not _cola_ok(window, overlap):
        warn("Selected window type and overlap do not meet the constant "
             "overlap and add constraint! Results might be distorted")
    n_frames = int(np.ceil(n_samples / hop))  # helpers/other.py:206
    return hop, n_frames


def _welch(x, y, fs_hz: int, window_type, window_length_samples: int, overlap_percent: float,
           detrend: bool, average: str, scaling: SpectrumScaling):
    """Welch auto (y None) or cross spectrum; shapes as in the reference."""
    auto = y is None
    x = np.asarray(x).squeeze()
    if not auto:
        y = np.asarray(y).squeeze()
        assert x.shape == y.shape, "Shapes of data do not match"
    assert x.ndim <= 2, f"{x.shape} are too many dimensions. Use flat arrays or 2D-Arrays instead"
    multi = x.ndim == 2
    _welch_checks(window_length_samples, overlap_percent, average)
    W = int(window_length_samples)
    window = _window_array(window_type, W)
    hop, n_frames = _welch_framing(x.shape[0], W, overlap_percent, window)
    amp, norm_scale, factor, phys = _finish_params(scaling, W, fs_hz, window)
    avg = DS_AVG[average]
    ctx = get_context()
    B = W // 2 + 1
    n_chan = x.shape[1] if multi else 1
    if _x64_short(SPEC_PRECISION, (1 if auto else 2) * n_chan, n_frames, W, average):
        # short estimate: the reference's own float64 arithmetic on the device (ds_welch_spec_x64)
        x64 = np.ascontiguousarray(x.reshape(x.shape[0], n_chan), dtype=np.float64)
        y64 = None if auto else np.ascontiguousarray(y.reshape(x.shape[0], n_chan), dtype=np.float64)
        w64 = np.ascontiguousarray(window, dtype=np.float64)
        out = np.empty((B, n_chan), dtype=np.complex128)
        ctx.check(ctx.lib.ds_welch_spec_x64(ctx.handle, _ptr(x64), None if auto else _ptr(y64), n_chan, x.shape[0], W, hop,
                                            n_frames, _ptr(w64), int(bool(detrend)), avg, amp, norm_scale, factor, phys,
                                            _ptr(out)), "ds_welch_spec_x64")
        res = out if (not auto or avg) else np.ascontiguousarray(out.real)  # (median auto spectra are complex in the reference)
        return res if multi else res[:, 0]
    w32 = window.astype(np.float32)
    if auto and _fusable(x):
        n, n_ch = x.shape
        out = np.empty((B, n_ch), dtype=np.float32)
        ctx.check(ctx.lib.ds_welch_psd_f64(ctx.handle, _ptr(x), n_ch, n, W, hop, n_frames, _ptr(w32),
                                           int(bool(detrend)), avg, amp, norm_scale, factor, phys,
                                           _ptr(out)), "ds_welch_psd_f64")
        res = out.astype(np.complex128 if avg else np.float64)
        return res if multi else res[:, 0]
    if not auto and _fusable(x) and _fusable(y):
        n, n_ch = x.shape
        out = np.empty((B, n_ch), dtype=np.complex64)
        ctx.check(ctx.lib.ds_welch_csd_f64(ctx.handle, _ptr(x), _ptr(y), n_ch, n, W, hop, n_frames, _ptr(w32),
                                           int(bool(detrend)), avg, amp, norm_scale, factor, phys,
                                           _ptr(out)), "ds_welch_csd_f64")
        return _widen(out)
    xp = _planar_f32(x)
    n_ch, n = xp.shape
    if auto:
        out = np.empty((B, n_ch), dtype=np.float32)
        ctx.check(ctx.lib.ds_welch_psd(ctx.handle, _ptr(xp), n_ch, n, W, hop, n_frames, _ptr(w32),
                                       int(bool(detrend)), avg, amp, norm_scale, factor, phys,
                                       _ptr(out)), "ds_welch_psd")
        # median averaging makes the reference's autospectrum complex128 (median_re + 1j*median_im)
        res = out.astype(np.complex128 if avg else np.float64)
    else:
        yp = _planar_f32(y)
        out = np.empty((B, n_ch), dtype=np.complex64)
        ctx.check(ctx.lib.ds_welch_csd(ctx.handle, _ptr(xp), _ptr(yp), n_ch, n, W, hop, n_frames,
                                       _ptr(w32), int(bool(detrend)), avg, amp, norm_scale, factor,
                                       phys, _ptr(out)), "ds_welch_csd")
        res = out.astype(np.complex128)
    return res if multi else res[:, 0]


# Arithmetic of the transfer-function estimate behind the reference-shaped API
# (transfer_functions.compute_transfer_function): "auto" takes the float64 route
# (ds_welch_tf_x64: float64 transforms, sums and finish, the reference's own precision) when the
# problem is small -- frame spectra of all channels <= 64 MB, window <= 262144 (median averaging: at
# most 4096 frames) -- or when it is SHORT: fewer than 128 frames (and <= 1.25 GB of frame spectra), where
# an fp32 estimate has too few frames to average its transform rounding down (the two sweep cases of
# round 2 that reached 1.1e-6 / 1.8e-6 in the coherence had 98 and 110 frames of 8192 samples: they are
# tests now) --
# and the fp32 kernels otherwise; "f32" / "f64" force one.  Environment:
# DSPTOOLBOX_AMD_TF_PRECISION.  backend.welch_transfer_function itself defaults to "f32".
TF_PRECISION = os.environ.get("DSPTOOLBOX_AMD_TF_PRECISION", "auto")
_X64_AUTO_BYTES = 64 << 20
_X64_MAX_WINDOW = 262144  # (16384 until round 4: k_frames_cls / k_split of kernels_welch_f64.hpp carry it to the reference's limit)
_X64_SHORT_BYTES = 1280 << 20  # frame spectra of a short estimate (transfer functions, auto / cross spectra): see _short_bytes
_X64_MATRIX_BYTES = 256 << 20  # ... of a short cross-spectral matrix (float64 pair sums grow with channels^2), and of the matrix itself


# The same for the Welch spectra themselves (get_spectrum with the Welch method: ds_welch_psd / ds_welch_csd) and
# the cross-spectral matrix (get_csm: ds_csm): "auto" sends SHORT estimates -- fewer than 128 frames, frame
# spectra <= 1.25 GB, window <= 262144; the matrix: up to 1024 channels, <= 256 MB of frame spectra and of matrices -- through
# ds_welch_spec_x64 / ds_csm_x64; "f32" keeps the fp32 kernels for every shape.  Environment:
# DSPTOOLBOX_AMD_SPEC_PRECISION.  (tests/sweeps/edge_welch.py, round 4: fp32 cross spectra and matrices of
# one to five frames reach 2-3e-6 of the largest element under the amplitude scalings.)
SPEC_PRECISION = os.environ.get("DSPTOOLBOX_AMD_SPEC_PRECISION", "auto")


def _short_bytes(W: int) -> int:
    """Byte cap of a short estimate's frame spectra on the float64 route: 1.25 GB for every window (at 50 % overlap the frame
    spectra of a signal are 16 bytes per sample whatever the window, so this is 64 + 1 channels x 2^20 samples).  It was 256 MB up
    to 16384-sample windows until round 5 (the float64 KERNELS are 5-8 x slower than the fp32 ones: 3.5 against 0.4 ms at
    1 GB); measured end to end through this host API with the reference's float64 arrays (tools/x64_cap_time.py,
    profiles/r05_sweeps.txt) the float64 route costs the same or less -- 13.4 against 12.6 ms at 1 GB of 8192-sample frames, 8.8
    against 22.6 ms and 3.7 against 6.2 ms at 0.5 / 0.3 GB of 16384-sample frames: the arrays cross PCIe as they are instead of
    through a host cast -- and the estimates of 45 ... 61 frames of 8192 / 16384 samples that the randomized sweeps found at
    1.0-1.7e-6 in the coherence on the fp32 kernels (20 + 20 and 33 + 33 channels: 320-530 MB) now take it."""
    return _X64_SHORT_BYTES


def _x64_short(precision, n_spectra: int, n_frames: int, W: int, average: str, cap: int | None = None) -> bool:
    """Does a SHORT estimate of `n_spectra` channel spectra take the float64 route?"""
    assert precision in ("auto", "f32"), "DSPTOOLBOX_AMD_SPEC_PRECISION: 'auto' or 'f32'"
    if precision != "auto" or W > _X64_MAX_WINDOW or n_frames >= 128:
        return False
    if average != "mean" and n_frames > 4096:
        return False
    return n_spectra * n_frames * (W // 2 + 1) * 16 <= (_short_bytes(W) if cap is None else cap)


def _tf_x64_applies(precision, n_cx: int, n_cy: int, n_frames: int, W: int, average: str) -> bool:
    ok = W <= _X64_MAX_WINDOW and (average == "mean" or n_frames <= 4096)
    if precision == "f64":
        if not ok:
            raise NotImplementedError("the float64 route covers windows up to 262144 (median: up to 4096 frames)")
        return True
    if precision == "auto":
        nbytes = (n_cx + n_cy) * n_frames * (W // 2 + 1) * 16
        return ok and (nbytes <= _X64_AUTO_BYTES or (n_frames < 128 and nbytes <= _short_bytes(W)))
    assert precision in (None, "f32"), "precision: 'f32', 'f64' or 'auto'"
    return False


def welch_transfer_function(output_td, input_td, fs_hz: int, window_length_samples: int, mode: str,
                            window_type=Window.Hann, overlap_percent: float = 50.0,
                            detrend: bool = True, average: str = "mean",
                            scaling: SpectrumScaling = SpectrumScaling.FFTBackward,
                            precision: str | None = None):
    """H1/H2/H3 + coherence for every output channel in one device call.
    output_td (N, Cy); input_td (N, 1) or (N, Cy).  -> (tf complex128 (B, Cy),
    coherence float64 (B, Cy)).  precision: "f32" (default), "f64" or "auto" (see TF_PRECISION)."""
    _welch_checks(window_length_samples, overlap_percent, average)
    if mode not in DS_TF:
        raise ValueError("Unsupported transfer function type")
    W = int(window_length_samples)
    window = _window_array(window_type, W)
    yo, xi = np.asarray(output_td), np.asarray(input_td)
    if yo.ndim == 1:
        yo = yo[:, None]
    if xi.ndim == 1:
        xi = xi[:, None]
    if precision not in (None, "f32"):
        n = yo.shape[0]
        assert xi.shape[0] == n, "Signal lengths do not match"
        hop, n_frames = _welch_framing(n, W, overlap_percent, window)
        if _tf_x64_applies(precision, xi.shape[1], yo.shape[1], n_frames, W, average):
            amp, norm_scale, factor, phys = _finish_params(scaling, W, fs_hz, window)
            y64 = np.ascontiguousarray(yo, dtype=np.float64)
            x64 = np.ascontiguousarray(xi, dtype=np.float64)
            w64 = np.ascontiguousarray(window, dtype=np.float64)
            B = W // 2 + 1
            tf = np.empty((B, yo.shape[1]), dtype=np.complex128)
            coh = np.empty((B, yo.shape[1]), dtype=np.float64)
            ctx = get_context()
            ctx.check(ctx.lib.ds_welch_tf_x64(ctx.handle, _ptr(x64), x64.shape[1], _ptr(y64), y64.shape[1], n, W,
                                              hop, n_frames, _ptr(w64), int(bool(detrend)), DS_AVG[average], DS_TF[mode], amp,
                                              norm_scale, factor, phys, _ptr(tf), _ptr(coh)), "ds_welch_tf_x64")
            return tf, coh
    # large float64 C-order arrays (the reference's own layout) cross the boundary as they are: the
    # library casts + transposes them in threads straight into pinned upload chunks
    fused = _fusable(yo) and xi.ndim == 2 and xi.dtype == np.float64 and xi.flags.c_contiguous
    if fused:
        n, n_cy = yo.shape
        n_cx = xi.shape[1]
        assert xi.shape[0] == n, "Signal lengths do not match"
    else:
        yp, xp = _planar_f32(yo), _planar_f32(xi)
        n_cy, n = yp.shape
        n_cx = xp.shape[0]
        assert xp.shape[1] == n, "Signal lengths do not match"
    hop, n_frames = _welch_framing(n, W, overlap_percent, window)
    amp, norm_scale, factor, phys = _finish_params(scaling, W, fs_hz, window)
    w32 = window.astype(np.float32)
    B = W // 2 + 1
    tf = np.empty((B, n_cy), dtype=np.complex64)
    coh = np.empty((B, n_cy), dtype=np.float32)
    ctx = get_context()
    if fused:
        ctx.check(ctx.lib.ds_welch_tf_f64(ctx.handle, _ptr(xi), n_cx, _ptr(yo), n_cy, n, W, hop, n_frames,
                                          _ptr(w32), int(bool(detrend)), DS_AVG[average], DS_TF[mode], amp,
                                          norm_scale, factor, phys, _ptr(tf), _ptr(coh)), "ds_welch_tf_f64")
    else:
        ctx.check(ctx.lib.ds_welch_tf(ctx.handle, _ptr(xp), n_cx, _ptr(yp), n_cy, n, W, hop, n_frames,
                                      _ptr(w32), int(bool(detrend)), DS_AVG[average], DS_TF[mode], amp,
                                      norm_scale, factor, phys, _ptr(tf), _ptr(coh)), "ds_welch_tf")
    return tf.astype(np.complex128), coh.astype(np.float64)


# ---- the same calls over samples that are ALREADY in HBM (Signal.to_device / from_planar_f32) -------------------------
# No cast, no transpose, no upload of the signal; small results come down through the context's page-locked staging
# buffer, results that are signals stay on the device (DevicePlanar).  Windows and taps are a few KB: uploaded per call.
class _Borrowed:
    """A context-owned device buffer lent to one call: free() is a no-op."""

    def __init__(self, buf):  # a DeviceBuffer, or a DevicePlanar (device-resident samples)
        self.ptr, self.nbytes, self.ctx = buf.ptr, getattr(buf, "nbytes", 0), buf.ctx

    def free(self):
        pass


def _window_dev(ctx, window: np.ndarray):
    """The float32 window on the device, kept per context (a handful of KB each, keyed by content): a resident call
    neither allocates nor uploads one (hipMalloc + hipFree cost more than the 0.12 ms of kernels they surround)."""
    cache = ctx.__dict__.setdefault("_window_cache", {})
    key = _window_key(window)
    hit = cache.get(key)
    if hit is None:
        if len(cache) >= 32:
            cache.pop(next(iter(cache)))[0].free()
        # (the window object rides along: an id is only a key while its object lives)
        hit = cache[key] = (DeviceBuffer.from_array(ctx, np.ascontiguousarray(window, dtype=np.float32)), window)
    return _Borrowed(hit[0])


def _result_scratch(ctx, nbytes: int):
    """Context-owned device buffer for SMALL results that are downloaded before the call returns (transfer functions,
    spectra): reused by every call of the thread, grown when needed."""
    cur = ctx.__dict__.get("_result_scratch")
    if cur is None or cur.nbytes < nbytes:
        if cur is not None:
            cur.free()
        cur = ctx.__dict__["_result_scratch"] = DeviceBuffer(ctx, max(int(nbytes), 1 << 22))
    return _Borrowed(cur)


def welch_transfer_function_device(y_dev: DevicePlanar, x_dev: DevicePlanar, fs_hz: int, window_length_samples: int,
                                   mode: str, window_type=Window.Hann, overlap_percent: float = 50.0,
                                   detrend: bool = True, average: str = "mean",
                                   scaling: SpectrumScaling = SpectrumScaling.FFTBackward, narrow: bool = False):
    """welch_transfer_function on device-resident planar float32 samples (fp32 kernels, ds_welch_tf_dev).
    -> (tf complex128 (B, Cy), coherence float64 (B, Cy)); narrow=True: the complex64 / float32 arrays as they came
    off the device (page-locked, owned by the caller) -- what Spectrum widens on first access."""
    _welch_checks(window_length_samples, overlap_percent, average)
    if mode not in DS_TF:
        raise ValueError("Unsupported transfer function type")
    W = int(window_length_samples)
    window = _window_array(window_type, W)
    n, n_cy, n_cx = y_dev.n_samples, y_dev.n_ch, x_dev.n_ch
    assert x_dev.n_samples == n, "Signal lengths do not match"
    hop, n_frames = _welch_framing(n, W, overlap_percent, window)
    amp, norm_scale, factor, phys = _finish_params(scaling, W, fs_hz, window)
    B = W // 2 + 1
    ctx = y_dev.ctx
    d_w = _window_dev(ctx, window)
    d_res = _result_scratch(ctx, B * n_cy * 12)
    try:
        ctx.check(ctx.lib.ds_welch_tf_dev(ctx.handle, C.c_void_p(x_dev.ptr), n_cx, x_dev.ld, C.c_void_p(y_dev.ptr), n_cy,
                                          y_dev.ld, n, W, hop, n_frames, C.c_void_p(d_w.ptr), int(bool(detrend)),
                                          DS_AVG[average], DS_TF[mode], amp, norm_scale, factor, phys,
                                          C.c_void_p(d_res.ptr), C.c_void_p(d_res.ptr + B * n_cy * 8)), "ds_welch_tf_dev")
        raw = ctx.download_result(d_res.ptr, (B * n_cy * 12,), np.uint8)
        tf = raw[:B * n_cy * 8].view(np.complex64).reshape(B, n_cy)
        coh = raw[B * n_cy * 8:].view(np.float32).reshape(B, n_cy)
    finally:
        d_w.free()
        d_res.free()
    return (tf, coh) if narrow else (_widen(tf), _widen(coh))


def _welch_psd_device(x_dev: DevicePlanar, fs_hz: int, window_type, window_length_samples: int, overlap_percent: float,
                      detrend: bool, average: str, scaling: SpectrumScaling):
    """Welch auto spectra of every channel of a device-resident signal (ds_welch_psd_dev) -> (B, C) as _welch."""
    _welch_checks(window_length_samples, overlap_percent, average)
    W = int(window_length_samples)
    window = _window_array(window_type, W)
    hop, n_frames = _welch_framing(x_dev.n_samples, W, overlap_percent, window)
    amp, norm_scale, factor, phys = _finish_params(scaling, W, fs_hz, window)
    B = W // 2 + 1
    ctx = x_dev.ctx
    d_w = _window_dev(ctx, window)
    d_o = _result_scratch(ctx, B * x_dev.n_ch * 4)
    try:
        ctx.check(ctx.lib.ds_welch_psd_dev(ctx.handle, C.c_void_p(x_dev.ptr), x_dev.n_ch, x_dev.ld, x_dev.n_samples, W, hop,
                                           n_frames, C.c_void_p(d_w.ptr), int(bool(detrend)), DS_AVG[average], amp, norm_scale,
                                           factor, phys, C.c_void_p(d_o.ptr)), "ds_welch_psd_dev")
        out = ctx.download_staged(d_o.ptr, (B, x_dev.n_ch), np.float32)
        return out.astype(np.complex128 if DS_AVG[average] else np.float64)
    finally:
        d_w.free()
        d_o.free()


class DeviceSTFT:
    """A spectrogram that stays in HBM: (bins, frames, channels) complex64 in `buf` -- the layout ds_stft_r2c writes,
    ds_istft and ds_band_power read.  What `Signal.get_spectrogram(on_device=True)` returns in place of the array;
    `to_host()` gives the reference's complex128 array."""

    def __init__(self, buf: DeviceBuffer, shape, power: bool):
        self.buf, self.shape, self.power = buf, tuple(int(v) for v in shape), bool(power)

    def to_host(self) -> np.ndarray:
        n = int(np.prod(self.shape))
        if self.power or n < (1 << 20):
            out = self.buf.to_array(self.shape, np.complex64)
            return out.real.astype(np.float64) if self.power else _widen(out)
        # a large spectrogram: page-locked chunks (the link's rate), each widened by the host threads while it is hot
        ctx, lib = self.buf.ctx, load_library()
        res = np.empty(self.shape, dtype=np.complex128)
        flat = res.reshape(-1)
        chunk = 4 << 20  # complex values: 32 MB down, 64 MB out
        for i0 in range(0, n, chunk):
            m = min(chunk, n - i0)
            part = ctx.download_staged(self.buf.ptr + 8 * i0, (m,), np.complex64)
            if lib.ds_host_widen_f64(_ptr(part), 2 * m, C.c_void_p(flat.ctypes.data + 16 * i0), 0) != 0:
                flat[i0:i0 + m] = part
        return res

    def __deepcopy__(self, memo):
        return self


def _stft_device(x_dev: DevicePlanar, fs_hz: int, window_length_samples: int, window_type, overlap_percent: float,
                 fft_length_samples, detrend: bool, padding: bool, scaling: SpectrumScaling, keep_on_device: bool):
    """_stft of a device-resident signal -> (time_s, freqs_hz, stft): stft the (B', F, C) complex128 array, or a
    DeviceSTFT when keep_on_device."""
    pl = _stft_plan(_ShapeOnly(x_dev.n_samples, x_dev.n_ch), fs_hz, window_length_samples, window_type, overlap_percent,
                    fft_length_samples, padding, scaling, planar=False)
    ctx = x_dev.ctx
    d_w = _window_dev(ctx, pl["w32"])
    shape = (pl["B"], pl["n_frames"], pl["n_ch"])
    d_s = DeviceBuffer(ctx, int(np.prod(shape)) * 8)
    try:
        ctx.check(ctx.lib.ds_stft_r2c_dev(ctx.handle, C.c_void_p(x_dev.ptr), pl["n"], pl["n_ch"], x_dev.ld, pl["W"], pl["hop"],
                                          pl["nfft"], pl["pad_front"], pl["n_frames"], C.c_void_p(d_w.ptr), int(bool(detrend)),
                                          pl["scale"], pl["edge"], pl["power"], C.c_void_p(d_s.ptr)), "ds_stft_r2c_dev")
        dev = DeviceSTFT(d_s, shape, pl["power"])
        if keep_on_device:
            ctx.sync()  # (the window buffer is freed below)
            return pl["time_s"], pl["freqs_hz"], dev
        out = dev.to_host()
        d_s.free()
        return pl["time_s"], pl["freqs_hz"], out
    finally:
        d_w.free()


class _ShapeOnly:
    """(N, C) shape carrier for _stft_plan(planar=False) when the samples are on the device."""

    def __init__(self, n: int, n_ch: int):
        self.shape = (int(n), int(n_ch))


def fir_filter_bank_device(x_dev: DevicePlanar, taps_list, mode: int):
    """fir_filter_bank over device-resident samples (ds_fir_ola_dev): Parallel -> a list of K DevicePlanar (slices of ONE
    output buffer, band-major as the kernel writes it), Sequential / Summed -> one DevicePlanar.  Nothing comes down."""
    taps = np.ascontiguousarray(np.stack([np.asarray(t, dtype=np.float64) for t in taps_list]), dtype=np.float32)
    k, t = taps.shape
    ctx = x_dev.ctx
    n, n_ch = x_dev.n_samples, x_dev.n_ch
    n_out = k if mode == DS_FB_PARALLEL else 1
    d_t = DeviceBuffer.from_array(ctx, taps)
    d_y = DeviceBuffer(ctx, n_out * n_ch * n * 4)
    try:
        ctx.check(ctx.lib.ds_fir_ola_dev(ctx.handle, C.c_void_p(x_dev.ptr), n_ch, x_dev.ld, n, C.c_void_p(d_t.ptr), k, t,
                                         int(mode), C.c_void_p(d_y.ptr), n), "ds_fir_ola_dev")
        ctx.sync()  # (the taps buffer is freed below)
    except BaseException:
        d_y.free()
        raise
    finally:
        d_t.free()
    outs = [DevicePlanar(d_y, n_ch, n, n, 4 * i * n_ch * n) for i in range(n_out)]
    return outs if mode == DS_FB_PARALLEL else outs[0]


def _istft_device(stft: DeviceSTFT, nfft: int, W: int, step: int, window, scale: float, frame_offset: int,
                  n_frames_total: int) -> DevicePlanar:
    """_istft of a device-resident spectrogram -> device-resident planar samples (ds_istft_dev)."""
    if nfft < 2:
        raise ValueError("fft_length_samples must be at least 2")
    assert not stft.power, "a power spectrogram has no phase to invert"
    n_bins, n_frames, n_ch = stft.shape
    if W > nfft:
        raise ValueError(f"operands could not be broadcast together with shapes ({nfft},{n_frames},{n_ch}) ({W},1,1)")
    total_length = int(step * n_frames_total + W * (1 - step / W))
    ctx = stft.buf.ctx
    d_w = _window_dev(ctx, window)
    d_o = DeviceBuffer(ctx, n_ch * total_length * 4)
    try:
        ctx.check(ctx.lib.ds_istft_dev(ctx.handle, C.c_void_p(stft.buf.ptr), n_bins, n_frames, n_ch, nfft, W, step,
                                       frame_offset, n_frames_total, C.c_void_p(d_w.ptr), float(scale), total_length,
                                       C.c_void_p(d_o.ptr), total_length), "ds_istft_dev")
        ctx.sync()
    except BaseException:
        d_o.free()
        raise
    finally:
        d_w.free()
    return DevicePlanar(d_o, n_ch, total_length)


def spectral_division_device(y_dev: DevicePlanar, x_dev: DevicePlanar, n_fft: int, n_out: int, eps_from_spectrum=None):
    """irfft(rfft(y, n_fft) * R, n_fft)[:n_out] with R = conj(X) / (|X|^2 + eps) (or 1 / X) built from the spectrum of the
    device-resident x (one channel for every channel of y, or one per channel), all on the device: ds_rfft_dev ->
    ds_deconv_inverse_dev -> ds_deconv_dev.  eps_from_spectrum(denum_fft (B, Cx) complex128) -> eps (B,) is the host's
    band detection (the reference's find_frequencies_above_threshold on channel 0); None: plain division.
    -> DevicePlanar (Cy, n_out)."""
    ctx = y_dev.ctx
    n, n_cy, n_cx = y_dev.n_samples, y_dev.n_ch, x_dev.n_ch
    assert x_dev.n_samples == n and n <= n_fft and n_out <= n_fft
    B = n_fft // 2 + 1
    d_xs = DeviceBuffer(ctx, B * n_cx * 8)
    d_r = DeviceBuffer(ctx, B * n_cx * 8)
    d_o = DeviceBuffer(ctx, n_cy * int(n_out) * 4)
    d_e = None
    try:
        ctx.check(ctx.lib.ds_rfft_dev(ctx.handle, C.c_void_p(x_dev.ptr), n_cx, x_dev.ld, n, int(n_fft), 1.0,
                                      C.c_void_p(d_xs.ptr)), "ds_rfft_dev")
        if eps_from_spectrum is not None:
            den = ctx.download_staged(d_xs.ptr, (B, n_cx), np.complex64).astype(np.complex128)
            d_e = DeviceBuffer.from_array(ctx, np.ascontiguousarray(eps_from_spectrum(den), dtype=np.float32))
        ctx.check(ctx.lib.ds_deconv_inverse_dev(ctx.handle, C.c_void_p(d_xs.ptr), n_cx, B, C.c_void_p(d_e.ptr) if d_e else None,
                                                C.c_void_p(d_r.ptr)), "ds_deconv_inverse")
        ctx.check(ctx.lib.ds_deconv_dev(ctx.handle, C.c_void_p(y_dev.ptr), 1, n_cy, y_dev.ld, n, int(n_fft), C.c_void_p(d_r.ptr),
                                        int(n_cx > 1), int(n_out), int(n_out), C.c_void_p(d_o.ptr)), "ds_deconv_dev")
        ctx.sync()
    except BaseException:
        d_o.free()
        raise
    finally:
        for d in (d_xs, d_r, d_e):
            if d is not None:
                d.free()
    return DevicePlanar(d_o, n_cy, int(n_out))


def _stft_plan(x, fs_hz: int, window_length_samples: int, window_type, overlap_percent: float,
               fft_length_samples, padding: bool, scaling: SpectrumScaling, planar: bool = True):
    """Argument checks and launch parameters of the STFT (shared by _stft and the fused
    spectrogram consumers)."""
    assert window_length_samples in [2**k for k in range(4, 17)], (
        "Window length should be a power of 2 between [16, 65536] or [2**4, 2**16]")
    assert overlap_percent >= 0 and overlap_percent < 100, \
        "overlap_percent should be between 0 and 100"
    W = int(window_length_samples)
    nfft = W if fft_length_samples is None else int(fft_length_samples)
    # any positive length, as numpy's rfft(n=...) (_spectral_methods.py:268): frames are cropped to
    # nfft or zero-padded; powers of two >= 8 take the fused kernels, everything else the general
    # route of the library (kernels_stft_any.hpp)
    assert nfft >= 2, "fft_length_samples must be at least 2"
    window = _window_array(window_type, W)
    overlap = int(overlap_percent / 100 * W + 0.5)  # rounding, _spectral_methods.py:247
    hop = W - overlap
    if not _cola_ok(window, overlap):
        warn("Selected window type and overlap do not meet the constant "
             "overlap and add constraint! Results might be distorted")
    if planar:
        xp = _planar_f32(x)
        n_ch, n = xp.shape
    else:  # the caller hands the (N, C) float64 array to a *_f64 entry point as it is
        xp = None
        n, n_ch = x.shape
    pad_front = overlap if padding else 0
    n_padded = n + 2 * pad_front
    n_frames = int(np.ceil(n_padded / hop))
    if scaling.has_physical_units():
        scale = float(np.asarray(scaling.get_scaling_factor(nfft, fs_hz, window)).ravel()[0])
        edge = 2**-0.5
        power = int(not scaling.is_amplitude_scaling())
    else:
        norm = scaling.fft_norm()
        scale = 1.0 if norm == "backward" else (1.0 / nfft if norm == "forward" else nfft**-0.5)
        edge, power = 1.0, 0
    time_s = np.linspace(0, n_padded / fs_hz, n_frames)
    freqs_hz = np.fft.rfftfreq(W, 1 / fs_hz)
    return dict(xp=xp, n=n, n_ch=n_ch, W=W, hop=hop, nfft=nfft, pad_front=pad_front, n_frames=n_frames,
                w32=window.astype(np.float32), scale=scale, edge=edge, power=power, B=nfft // 2 + 1,
                time_s=time_s, freqs_hz=freqs_hz)


def _stft(x, fs_hz: int, window_length_samples: int, window_type, overlap_percent: float,
          fft_length_samples, detrend: bool, padding: bool, scaling: SpectrumScaling):
    """-> (time_s (F,), freqs_hz (B,), stft (B', F, C))."""
    xa = np.asarray(x)
    if _fusable(xa):
        # float64 on both sides: threaded cast into pinned upload chunks, widened back from pinned
        # download chunks (power scalings keep only the real part: numpy path below)
        pl = _stft_plan(xa, fs_hz, window_length_samples, window_type, overlap_percent, fft_length_samples,
                        padding, scaling, planar=False)
        if not pl["power"]:
            out = np.empty((pl["B"], pl["n_frames"], pl["n_ch"]), dtype=np.complex128)
            ctx = get_context()
            ctx.check(ctx.lib.ds_stft_r2c_f64(ctx.handle, _ptr(xa), pl["n"], pl["n_ch"], pl["W"], pl["hop"],
                                              pl["nfft"], pl["pad_front"], pl["n_frames"], _ptr(pl["w32"]),
                                              int(bool(detrend)), pl["scale"], pl["edge"], pl["power"],
                                              _ptr(out)), "ds_stft_r2c_f64")
            return pl["time_s"], pl["freqs_hz"], out
    pl = _stft_plan(x, fs_hz, window_length_samples, window_type, overlap_percent, fft_length_samples,
                    padding, scaling)
    out = np.empty((pl["B"], pl["n_frames"], pl["n_ch"]), dtype=np.complex64)
    ctx = get_context()
    ctx.check(ctx.lib.ds_stft_r2c(ctx.handle, _ptr(pl["xp"]), pl["n"], pl["n_ch"], pl["W"], pl["hop"],
                                  pl["nfft"], pl["pad_front"], pl["n_frames"], _ptr(pl["w32"]),
                                  int(bool(detrend)), pl["scale"], pl["edge"], pl["power"], _ptr(out)),
              "ds_stft_r2c")
    stft = out.real.astype(np.float64) if pl["power"] else _widen(out)
    return pl["time_s"], pl["freqs_hz"], stft


def _spectrogram_band_power(x, fs_hz: int, window_length_samples: int, window_type, overlap_percent: float,
                            fft_length_samples, detrend: bool, padding: bool, scaling: SpectrumScaling,
                            band_filters, to_db: bool, dct_abs: bool):
    """STFT -> sum_b filters[band, b] |stft[b]|^2 (-> dB -> |DCT-II| over bands), everything on the
    device: the spectrogram never travels to the host.  band_filters (bands, B').
    -> (time_s, freqs_hz, out (bands, F, C) float64)."""
    resident = isinstance(x, DevicePlanar)  # a device-resident signal: its samples are read in place
    if resident:
        pl = _stft_plan(_ShapeOnly(x.n_samples, x.n_ch), fs_hz, window_length_samples, window_type, overlap_percent,
                        fft_length_samples, padding, scaling, planar=False)
    else:
        pl = _stft_plan(x, fs_hz, window_length_samples, window_type, overlap_percent, fft_length_samples,
                        padding, scaling)
    filt = np.ascontiguousarray(band_filters, dtype=np.float32)
    assert filt.ndim == 2 and filt.shape[1] == pl["B"], (
        f"Shape of the mel filter matrix {filt.shape} does not match the STFT "
        f"{(pl['B'], pl['n_frames'], pl['n_ch'])}")
    n_bands = filt.shape[0]
    nz = filt != 0
    b0 = np.where(nz.any(axis=1), nz.argmax(axis=1), 0).astype(np.int32)
    b1 = np.where(nz.any(axis=1), filt.shape[1] - nz[:, ::-1].argmax(axis=1), 0).astype(np.int32)
    ctx = x.ctx if resident else get_context()
    n_fc = pl["n_frames"] * pl["n_ch"]
    d_x = _Borrowed(x) if resident else DeviceBuffer.from_array(ctx, pl["xp"])
    x_ld = x.ld if resident else pl["n"]
    d_w = DeviceBuffer.from_array(ctx, pl["w32"])
    d_s = DeviceBuffer(ctx, pl["B"] * n_fc * 8)
    d_f = DeviceBuffer.from_array(ctx, filt)
    d_b0, d_b1 = DeviceBuffer.from_array(ctx, b0), DeviceBuffer.from_array(ctx, b1)
    d_o = DeviceBuffer(ctx, n_bands * n_fc * 4)
    try:
        ctx.check(ctx.lib.ds_stft_r2c_dev(ctx.handle, C.c_void_p(d_x.ptr), pl["n"], pl["n_ch"], x_ld, pl["W"],
                                          pl["hop"], pl["nfft"], pl["pad_front"], pl["n_frames"],
                                          C.c_void_p(d_w.ptr), int(bool(detrend)), pl["scale"], pl["edge"],
                                          pl["power"], C.c_void_p(d_s.ptr)), "ds_stft_r2c_dev")
        ctx.check(ctx.lib.ds_band_power_dev(ctx.handle, C.c_void_p(d_s.ptr), pl["B"], n_fc, C.c_void_p(d_f.ptr),
                                            C.c_void_p(d_b0.ptr), C.c_void_p(d_b1.ptr), n_bands,
                                            int(bool(to_db)), int(bool(dct_abs)), C.c_void_p(d_o.ptr)),
                  "ds_band_power_dev")
        out = d_o.to_array((n_bands, pl["n_frames"], pl["n_ch"]), np.float32)
    finally:
        for d in (d_x, d_w, d_s, d_f, d_b0, d_b1, d_o):
            d.free()
    return pl["time_s"], pl["freqs_hz"], out.astype(np.float64)


def _das_map(csm, h):
    """Re(h^H csm h) per grid point and bin: csm (F, C, C), h (F, C, G) -> (G, F) float64."""
    cs = np.ascontiguousarray(csm, dtype=np.complex64)
    hs = np.ascontiguousarray(h, dtype=np.complex64)
    assert cs.ndim == 3 and hs.ndim == 3 and cs.shape[1] == cs.shape[2], "csm must be (bins, C, C)"
    assert hs.shape[0] == cs.shape[0] and hs.shape[1] == cs.shape[1], \
        "steering vector must be (bins, C, grid points)"
    n_bins, n_ch, n_grid = hs.shape
    out = np.empty((n_grid, n_bins), dtype=np.float32)
    ctx = get_context()
    ctx.check(ctx.lib.ds_das_map(ctx.handle, _ptr(cs), _ptr(hs), n_bins, n_ch, n_grid, _ptr(out)),
              "ds_das_map")
    return out.astype(np.float64)


def _istft(stft, nfft: int, W: int, step: int, window, scale: float, frame_offset: int,
           n_frames_total: int):
    """Frame-wise irfft (length nfft, cropped to W) * scale * window, overlap-added at
    (frame + frame_offset) * step and divided by the squared-window envelope clipped at 1e-4
    (standard/_framed_signal_representation.py:70-137).  stft (B, F, C) -> (total_length, C)."""
    if nfft < 2:
        raise ValueError("fft_length_samples must be at least 2")
    if np.isrealobj(stft):
        stft = stft.astype(np.complex128)
    n_bins, n_frames, n_ch = stft.shape
    if W > nfft:  # the reference's `td_framed *= window[:, None, None]` cannot broadcast either
        raise ValueError(f"operands could not be broadcast together with shapes ({nfft},{n_frames},{n_ch}) ({W},1,1)")
    # length of the reference's reconstruction buffer (same float expression, :112-115)
    total_length = int(step * n_frames_total + W * (1 - step / W))
    if stft.dtype == np.complex128 and stft.flags.c_contiguous and stft.size >= (1 << 19):
        # a large complex128 spectrogram: narrowed in host threads into pinned upload chunks, float64 (N, C) back
        res = np.empty((total_length, n_ch), dtype=np.float64)
        w32 = np.ascontiguousarray(window, dtype=np.float32)
        ctx = get_context()
        ctx.check(ctx.lib.ds_istft_f64(ctx.handle, _ptr(stft), n_bins, n_frames, n_ch, nfft, W, step, frame_offset,
                                       n_frames_total, _ptr(w32), float(scale), total_length, _ptr(res)), "ds_istft_f64")
        return res
    sp = np.ascontiguousarray(stft, dtype=np.complex64)
    out = np.empty((n_ch, total_length), dtype=np.float32)
    w32 = np.ascontiguousarray(window, dtype=np.float32)
    ctx = get_context()
    ctx.check(ctx.lib.ds_istft(ctx.handle, _ptr(sp), n_bins, n_frames, n_ch, nfft, W, step, frame_offset,
                               n_frames_total, _ptr(w32), float(scale), total_length, _ptr(out)), "ds_istft")
    return _interleaved_f64(out)


_STAGED_RESULT_BYTES = 512 << 20  # largest result that goes through the context's page-locked staging buffer (which stays allocated)


def _csm_welch(time_data, sampling_rate_hz: int, window_length_samples: int, window_type,
               overlap_percent, detrend: bool, average: str, scaling: SpectrumScaling):
    """-> (f (B,), csm (B, C, C) complex128)."""
    _welch_checks(window_length_samples, overlap_percent, average)
    W = int(window_length_samples)
    window = _window_array(window_type, W)
    td = np.asarray(time_data)
    fused = _fusable(td)
    if fused:
        n, n_ch = td.shape
    else:
        xp = _planar_f32(td)
        n_ch, n = xp.shape
    hop, n_frames = _welch_framing(n, W, overlap_percent, window)
    amp, norm_scale, factor, phys = _finish_params(scaling, W, sampling_rate_hz, window)
    B = W // 2 + 1
    if n_ch <= 1024 and _x64_short(SPEC_PRECISION, n_ch, n_frames, W, average, cap=_X64_MATRIX_BYTES) \
            and B * n_ch * n_ch * 16 <= _X64_MATRIX_BYTES:  # (the matrix itself, complex128, crosses PCIe too)
        # short estimate: float64 end to end on the device (ds_csm_x64)
        x64 = np.ascontiguousarray(td.reshape(n, n_ch) if td.ndim == 2 else td[:, None], dtype=np.float64)
        out64 = np.empty((B, n_ch, n_ch), dtype=np.complex128)
        w64 = np.ascontiguousarray(window, dtype=np.float64)
        ctx = get_context()
        ctx.check(ctx.lib.ds_csm_x64(ctx.handle, _ptr(x64), n_ch, n, W, hop, n_frames, _ptr(w64), int(bool(detrend)),
                                     DS_AVG[average], amp, norm_scale, factor, phys, _ptr(out64)), "ds_csm_x64")
        return np.fft.rfftfreq(W, 1 / sampling_rate_hz), out64
    w32 = window.astype(np.float32)
    ctx = get_context()
    # the complex64 matrices land in the context's page-locked staging buffer (the link's full rate, no first-touch faults of
    # a fresh array) and are widened out of it into the complex128 array the caller gets
    # (up to 512 MB; a larger result -- hundreds of channels -- comes back into an ordinary array)
    nbytes = B * n_ch * n_ch * 8
    out = (ctx.staging(nbytes).view(np.complex64).reshape(B, n_ch, n_ch) if nbytes <= _STAGED_RESULT_BYTES
           else np.empty((B, n_ch, n_ch), dtype=np.complex64))
    if fused:
        ctx.check(ctx.lib.ds_csm_f64(ctx.handle, _ptr(td), n_ch, n, W, hop, n_frames, _ptr(w32),
                                     int(bool(detrend)), DS_AVG[average], amp, norm_scale, factor, phys,
                                     _ptr(out)), "ds_csm_f64")
    else:
        ctx.check(ctx.lib.ds_csm(ctx.handle, _ptr(xp), n_ch, n, W, hop, n_frames, _ptr(w32),
                                 int(bool(detrend)), DS_AVG[average], amp, norm_scale, factor, phys,
                                 _ptr(out)), "ds_csm")
    return np.fft.rfftfreq(W, 1 / sampling_rate_hz), _widen(out)


class DeviceCSM:
    """A Welch cross-spectral matrix that stays in HBM: (bins, C, C) complex64 in `buf`, the frequency
    vector on the host.  What `Signal.get_csm(on_device=True)` returns and the device delay-and-sum map
    consumes (the reference's beamformers take `self.signal.get_csm()` and keep going,
    beamforming/beamforming.py:838-876)."""

    def __init__(self, ctx, buf, freqs_hz, n_ch: int):
        self.ctx, self.buf, self.freqs_hz, self.n_ch = ctx, buf, freqs_hz, int(n_ch)
        self.n_bins = len(freqs_hz)

    def to_host(self) -> np.ndarray:
        shape = (self.n_bins, self.n_ch, self.n_ch)
        if self.n_bins * self.n_ch * self.n_ch * 8 > _STAGED_RESULT_BYTES:
            return _widen(self.buf.to_array(shape, np.complex64))
        return _widen(self.ctx.download_staged(self.buf.ptr, shape, np.complex64))

    def free(self):
        self.buf.free()


def _csm_welch_device(time_data, sampling_rate_hz: int, window_length_samples: int, window_type,
                      overlap_percent, detrend: bool, average: str, scaling: SpectrumScaling) -> DeviceCSM:
    """_csm_welch whose result stays on the device (ds_csm_dev) -> DeviceCSM.  time_data: the (N, C) array, or the
    DevicePlanar of a device-resident signal (read in place)."""
    _welch_checks(window_length_samples, overlap_percent, average)
    W = int(window_length_samples)
    window = _window_array(window_type, W)
    resident = isinstance(time_data, DevicePlanar)
    if resident:
        ctx, n_ch, n, ld = time_data.ctx, time_data.n_ch, time_data.n_samples, time_data.ld
        d_x = None
        x_ptr = time_data.ptr
    else:
        xp = _planar_f32(np.asarray(time_data))
        n_ch, n = xp.shape
        ld = n
        ctx = get_context()
        d_x = DeviceBuffer.from_array(ctx, xp)
        x_ptr = d_x.ptr
    hop, n_frames = _welch_framing(n, W, overlap_percent, window)
    amp, norm_scale, factor, phys = _finish_params(scaling, W, sampling_rate_hz, window)
    B = W // 2 + 1
    d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
    d_c = DeviceBuffer(ctx, B * n_ch * n_ch * 8)
    try:
        ctx.check(ctx.lib.ds_csm_dev(ctx.handle, C.c_void_p(x_ptr), n_ch, ld, n, W, hop, n_frames,
                                     C.c_void_p(d_w.ptr), int(bool(detrend)), DS_AVG[average], amp, norm_scale, factor,
                                     phys, C.c_void_p(d_c.ptr)), "ds_csm_dev")
        ctx.sync()
    except BaseException:
        d_c.free()  # (the matrix only leaves this function inside a DeviceCSM)
        raise
    finally:
        if d_x is not None:
            d_x.free()
        d_w.free()
    return DeviceCSM(ctx, d_c, np.fft.rfftfreq(W, 1 / sampling_rate_hz), n_ch)


def _das_map_device(csm: DeviceCSM, id1: int, id2: int, h, remove_csm_diagonal: bool):
    """Delay-and-sum map of the bins [id1, id2) of a device-resident CSM: the diagonal treatment
    (beamforming.py:840-845) and Re(h^H csm h) (:853-858) on the device; only the steering vectors go up and
    the (grid points, bins) map comes down.  h (bins, C, G) complex -> (G, bins) float64."""
    hs = np.ascontiguousarray(h, dtype=np.complex64)
    nb = id2 - id1
    assert hs.ndim == 3 and hs.shape[0] == nb and hs.shape[1] == csm.n_ch, "steering vector must be (bins, C, grid points)"
    n_grid = hs.shape[2]
    ctx = csm.ctx
    d_h = DeviceBuffer.from_array(ctx, hs)
    d_s = DeviceBuffer(ctx, nb * csm.n_ch * csm.n_ch * 8)
    d_m = DeviceBuffer(ctx, n_grid * nb * 4)
    try:
        src = csm.buf.ptr + id1 * csm.n_ch * csm.n_ch * 8
        scale = csm.n_ch / (csm.n_ch - 1) if remove_csm_diagonal else 1.0
        ctx.check(ctx.lib.ds_csm_das_prepare_dev(ctx.handle, C.c_void_p(src), nb, csm.n_ch, float(scale),
                                                 int(bool(remove_csm_diagonal)), C.c_void_p(d_s.ptr)), "ds_csm_das_prepare_dev")
        ctx.check(ctx.lib.ds_das_map_dev(ctx.handle, C.c_void_p(d_s.ptr), C.c_void_p(d_h.ptr), nb, csm.n_ch, n_grid,
                                         C.c_void_p(d_m.ptr)), "ds_das_map_dev")
        out = d_m.to_array((n_grid, nb), np.float32)
    finally:
        for d in (d_h, d_s, d_m):
            d.free()
    return out.astype(np.float64)


def _csm_welch_bins(time_data, sampling_rate_hz: int, window_length_samples: int, window_type,
                    overlap_percent, detrend: bool, scaling: SpectrumScaling, bin_start: int,
                    bin_stop: int):
    """Rows [bin_start, bin_stop) of the Welch CSM (mean averaging): one rank's share when the
    matrix is split by frequency bins.  -> (bin_stop - bin_start, C, C) complex128."""
    _welch_checks(window_length_samples, overlap_percent, "mean")
    W = int(window_length_samples)
    window = _window_array(window_type, W)
    xp = _planar_f32(time_data)
    n_ch, n = xp.shape
    hop, n_frames = _welch_framing(n, W, overlap_percent, window)
    amp, norm_scale, factor, phys = _finish_params(scaling, W, sampling_rate_hz, window)
    count = int(bin_stop) - int(bin_start)
    if count <= 0:
        return np.zeros((0, n_ch, n_ch), dtype=np.complex128)
    ctx = get_context()
    d_x = DeviceBuffer.from_array(ctx, xp)
    d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
    d_c = DeviceBuffer(ctx, count * n_ch * n_ch * 8)
    try:
        ctx.check(ctx.lib.ds_csm_bins_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, W, hop, n_frames,
                                          C.c_void_p(d_w.ptr), int(bool(detrend)), amp, norm_scale, factor,
                                          phys, int(bin_start), count, C.c_void_p(d_c.ptr)), "ds_csm_bins_dev")
        out = d_c.to_array((count, n_ch, n_ch), np.complex64)
    finally:
        for d in (d_x, d_w, d_c):
            d.free()
    return out.astype(np.complex128)


def _csm_fft(spectrum, scaling: SpectrumScaling, window, sampling_rate_hz: int):
    """Cross-spectral matrix of ONE whole-signal spectrum (B, C) (FFTBackward-normalised),
    dsptoolbox/standard/_spectral_methods.py:374-443.  -> (B, C, C) complex128."""
    if window is not None:
        raise NotImplementedError("time-windowed signals are outside the GPU hot path")
    xs = np.ascontiguousarray(spectrum, dtype=np.complex64)
    nb, n_ch = xs.shape
    if scaling == SpectrumScaling.FFTBackward:
        amp, factor, halve = 0, 1.0, 0
    else:
        # the reference passes `spectrum.shape[0] // 2 + 1` as the length (:436)
        factor = float(np.asarray(SpectrumScaling.FFTBackward.conversion_factor(
            scaling, nb // 2 + 1, sampling_rate_hz, None)).ravel()[0])
        amp, halve = int(scaling.is_amplitude_scaling()), 1
    out = np.empty((nb, n_ch, n_ch), dtype=np.complex64)
    ctx = get_context()
    ctx.check(ctx.lib.ds_csm_spec(ctx.handle, _ptr(xs), nb, 1, n_ch, amp, 1.0, factor, halve,
                                  _ptr(out)), "ds_csm_spec")
    return out.astype(np.complex128)


def rfft_spectrum(time_data, n_fft: int, scale: float = 1.0):
    """rfft(time_data, n=n_fft, axis=0) * scale -> (n_fft/2+1, C) complex128."""
    if _fusable(time_data):  # float64 in, complex128 out through the pinned chunk pipelines
        n, n_ch = time_data.shape
        out = np.empty((n_fft // 2 + 1, n_ch), dtype=np.complex128)
        ctx = get_context()
        ctx.check(ctx.lib.ds_rfft_f64(ctx.handle, _ptr(time_data), n_ch, n, int(n_fft), float(scale), _ptr(out)), "ds_rfft_f64")
        return out
    xp = _planar_f32(time_data)
    n_ch, n = xp.shape
    out = np.empty((n_fft // 2 + 1, n_ch), dtype=np.complex64)
    ctx = get_context()
    ctx.check(ctx.lib.ds_rfft(ctx.handle, _ptr(xp), n_ch, n, int(n_fft), float(scale), _ptr(out)),
              "ds_rfft")
    return out.astype(np.complex128)


def spectral_division(num_td, n_fft: int, inverse_spectrum, n_out: int):
    """irfft(rfft(num_td, n_fft) * inverse_spectrum, n_fft)[:n_out] per channel.
    num_td (N, C) or (M, N, C) for a batch of M items; inverse_spectrum (B,)
    shared or (B, C) per channel.  -> same leading shape, float64."""
    num_td = np.asarray(num_td)
    batched = num_td.ndim == 3
    if not batched and _fusable(num_td):  # one large item: float64 on both sides through the pinned chunk pipelines
        n, n_ch = num_td.shape
        r = np.asarray(inverse_spectrum)
        per_channel = r.ndim == 2
        rp = np.ascontiguousarray(r.T if per_channel else r, dtype=np.complex64)
        assert rp.shape[-1] == n_fft // 2 + 1, "Frequency vector does not match"
        res = np.empty((int(n_out), n_ch), dtype=np.float64)
        ctx = get_context()
        ctx.check(ctx.lib.ds_deconv_f64(ctx.handle, _ptr(num_td), n_ch, n, int(n_fft), _ptr(rp), int(per_channel), int(n_out),
                                        _ptr(res)), "ds_deconv_f64")
        return res
    items = num_td if batched else num_td[None]
    m, n, n_ch = items.shape
    yp = np.empty((m, n_ch, n), dtype=np.float32)  # (M, C, N)
    if n * n_ch >= (1 << 20):
        for i in range(m):
            yp[i] = _planar_f32(items[i])
    else:
        yp[...] = np.transpose(items, (0, 2, 1))
    r = np.asarray(inverse_spectrum)
    per_channel = r.ndim == 2
    rp = np.ascontiguousarray(r.T if per_channel else r, dtype=np.complex64)  # (C, B) or (B,)
    assert rp.shape[-1] == n_fft // 2 + 1, "Frequency vector does not match"
    out = np.empty((m, n_ch, n_out), dtype=np.float32)
    ctx = get_context()
    ctx.check(ctx.lib.ds_deconv(ctx.handle, _ptr(yp), m, n_ch, n, int(n_fft), _ptr(rp),
                                int(per_channel), int(n_out), _ptr(out)), "ds_deconv")
    res = np.empty((m, n_out, n_ch), dtype=np.float64)
    if n_out * n_ch >= (1 << 20):
        for i in range(m):
            _interleaved_f64(out[i], res[i])
    else:
        res[...] = np.transpose(out, (0, 2, 1))
    return res if batched else res[0]


def regularized_inverse(denum_spectrum, eps=None):
    """conj(X)/(|X|^2+eps) (regularised) or 1/X, on the device.  denum_spectrum
    (B, C) complex; eps (B,) or None.  -> (B, C) complex128."""
    xs = np.ascontiguousarray(denum_spectrum, dtype=np.complex64)
    nb, n_ch = xs.shape
    ctx = get_context()
    d_x = DeviceBuffer.from_array(ctx, xs)
    d_e = DeviceBuffer.from_array(ctx, np.ascontiguousarray(eps, dtype=np.float32)) if eps is not None else None
    d_r = DeviceBuffer(ctx, xs.nbytes)
    ctx.check(ctx.lib.ds_deconv_inverse_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, nb,
                                            C.c_void_p(d_e.ptr) if d_e else None,
                                            C.c_void_p(d_r.ptr)), "ds_deconv_inverse")
    r = d_r.to_array((n_ch, nb), np.complex64)
    for d in (d_x, d_e, d_r):
        if d is not None:
            d.free()
    return r.T.astype(np.complex128)


def fir_filter_bank(x, taps_list, mode: int):
    """x (N, C); taps_list K arrays of equal length T.  Parallel -> (K, N, C);
    Sequential / Summed -> (N, C).  float64."""
    taps = np.ascontiguousarray(np.stack([np.asarray(t, dtype=np.float64) for t in taps_list]),
                                dtype=np.float32)
    k, t = taps.shape
    xa = np.asarray(x)
    if _fusable(xa):  # float64 on both sides through the pinned chunk pipelines
        n, n_ch = xa.shape
        res = np.empty(((k if mode == DS_FB_PARALLEL else 1), n, n_ch), dtype=np.float64)
        ctx = get_context()
        ctx.check(ctx.lib.ds_fir_ola_f64(ctx.handle, _ptr(xa), n_ch, n, _ptr(taps), k, t, int(mode), _ptr(res)),
                  "ds_fir_ola_f64")
        return res if mode == DS_FB_PARALLEL else res[0]
    xp = _planar_f32(x)
    n_ch, n = xp.shape
    out = np.empty(((k if mode == DS_FB_PARALLEL else 1), n_ch, n), dtype=np.float32)
    ctx = get_context()
    ctx.check(ctx.lib.ds_fir_ola(ctx.handle, _ptr(xp), n_ch, n, _ptr(taps), k, t, int(mode),
                                 _ptr(out)), "ds_fir_ola")
    res = np.empty((out.shape[0], n, n_ch), dtype=np.float64)
    for i in range(out.shape[0]):
        _interleaved_f64(out[i], res[i])
    return res if mode == DS_FB_PARALLEL else res[0]


def _lfilter_fir(b, a, x, zi=None, axis: int = 0):
    """Causal FIR filtering y = (x * b)[:N] along axis 0 (dsptoolbox/classes/filter_helpers.py:
    454-503).  With zi (T-1, C): the full convolution (N + T - 1 samples, device) gets zi added
    to its head, the new state is its tail; returns (y, zf)."""
    a = np.atleast_1d(a)
    assert len(a) == 1, f"{a} is not valid. It has to be 1 in order to be a valid FIR filter"
    b = np.asarray(b)
    if b.ndim != 1:
        b = np.squeeze(b)
        assert b.ndim == 1, "FIR Filters for audio must be 1D-arrays"
    x = np.asarray(x)
    if np.iscomplexobj(x):
        raise NotImplementedError("complex input signals are outside the GPU hot path (Signal.time_data is real)")
    if np.iscomplexobj(b):
        # complex taps on a real signal (filter_helpers.py:364-371 stores the imaginary part of the output in
        # Signal.time_data_imaginary): two real convolutions, x * Re b + i x * Im b, one device call
        return _lfilter_fir_complex(b, x, zi)
    if zi is not None:
        zi = np.asarray(zi)
        assert zi.ndim == x.ndim, \
            "Vector to filter and initial values should have the same number of dimensions!"
    if x.ndim < 2:
        x = x[..., None]
        if zi is not None:
            zi = zi[..., None]
    assert x.ndim == 2, "Filtering only works on 2D-arrays"
    if zi is None:
        return fir_filter_bank(x, [b], DS_FB_PARALLEL)[0]
    # mode="full": run the device convolution over the signal followed by T - 1 zeros
    xfull = np.concatenate([x, np.zeros((len(b) - 1, x.shape[1]))], axis=0)
    y = fir_filter_bank(xfull, [b], DS_FB_PARALLEL)[0]
    y[: zi.shape[0], :] += zi
    zf = y[-zi.shape[0]:, :]
    return y[: x.shape[0], :], zf


def fir_transfer_function(taps_list, frequency_vector_hz, sampling_rate_hz: int) -> np.ndarray:
    """H_k(f) = sum_n b_k[n] exp(-2 pi i f n / fs) for every filter of the list at every frequency, float64 on
    the device (ds_fir_freqz) -- scipy.signal.freqz(b, 1, worN=f, fs=fs)[1] of the reference's
    Filter.get_transfer_function (classes/filter.py:893-900).  -> (frequencies, filters) complex128."""
    f = np.ascontiguousarray(frequency_vector_hz, dtype=np.float64)
    assert f.ndim == 1, "Frequency vector can only have one dimension"
    n_taps = max(len(t) for t in taps_list)
    taps = np.zeros((len(taps_list), n_taps), dtype=np.complex128)
    for k, t in enumerate(taps_list):
        taps[k, :len(t)] = t
    out = np.empty((len(taps_list), len(f)), dtype=np.complex128)
    ctx = get_context()
    ctx.check(ctx.lib.ds_fir_freqz(ctx.handle, _ptr(taps), taps.shape[0], n_taps, _ptr(f), len(f),
                                   float(sampling_rate_hz), _ptr(out)), "ds_fir_freqz")
    return np.ascontiguousarray(out.T)


def _pad_trim(vector: np.ndarray, desired_length: int) -> np.ndarray:
    """Zero-pad or trim the END of axis 0 (helpers/other.py:216-259 with its defaults)."""
    v = np.asarray(vector)
    n = v.shape[0]
    if n == desired_length:
        return v.copy()
    if n > desired_length:
        return v[:desired_length].copy()
    return np.concatenate([v, np.zeros((desired_length - n,) + v.shape[1:], dtype=v.dtype)], axis=0)


def _lfilter_fir_complex(b, x, zi=None):
    """_lfilter_fir for complex taps b on a real x: the real and the imaginary part of b are two band
    filters of one parallel bank; state (complex) as in the real case."""
    if zi is not None:
        zi = np.asarray(zi)
    if x.ndim < 2:
        x = x[..., None]
        if zi is not None:
            zi = zi[..., None]
    assert x.ndim == 2, "Filtering only works on 2D-arrays"
    br, bi = np.ascontiguousarray(b.real), np.ascontiguousarray(b.imag)
    if zi is None:
        out = fir_filter_bank(x, [br, bi], DS_FB_PARALLEL)
        return out[0] + 1j * out[1]
    xfull = np.concatenate([x, np.zeros((len(b) - 1, x.shape[1]))], axis=0)
    out = fir_filter_bank(xfull, [br, bi], DS_FB_PARALLEL)
    y = out[0] + 1j * out[1]
    y[: zi.shape[0], :] += zi
    zf = y[-zi.shape[0]:, :]
    return y[: x.shape[0], :], zf


def _lfilter_zi_fir(b):
    """scipy.signal.lfilter_zi(b, [1.0]) in closed form (the reference's Filter.initialize_zi,
    classes/filter.py:331-353): steady-state step-response state of a transposed direct-form
    FIR filter, zi[i] = sum_{j > i} b[j].  Host-side parameter preparation."""
    b = np.asarray(b)
    b = b.astype(np.complex128 if np.iscomplexobj(b) else np.float64)
    return np.cumsum(b[::-1])[::-1][1:].copy()


def _filtfilt_fir(b, x):
    """Zero-phase FIR filtering = scipy.signal.filtfilt(b, [1.0], x, axis=0) with its defaults
    (odd extension by 3*len(b) samples, steady-state initial conditions), the zero_phase branch
    of the reference (filter_helpers.py:362-363).  Both convolutions run on the device; the
    extension, the time reversals and the state terms are host-side array plumbing."""
    b = np.asarray(b, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    if x.ndim < 2:
        x = x[..., None]
    edge = 3 * len(b)
    if x.shape[0] <= edge:
        raise ValueError(
            "The length of the input vector x must be greater than padlen, which is %d." % edge)
    ext = np.concatenate([2 * x[:1] - x[edge:0:-1], x, 2 * x[-1:] - x[-2:-(edge + 2):-1]], axis=0)
    zi = _lfilter_zi_fir(b)[:, None]
    y, _ = _lfilter_fir(b, [1.0], ext, zi * ext[:1])
    y, _ = _lfilter_fir(b, [1.0], y[::-1], zi * y[-1:])
    return y[::-1][edge:-edge]
