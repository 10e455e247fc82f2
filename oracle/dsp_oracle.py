"""CPU oracle for the batched spectral hot path -- TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy float64 restatement of the reference algorithms
(dsptoolbox 0.8, read-only at /root/reference).  It is the *checker* for the
HIP product path and the `cpu_baseline` leg of bench.py.  Nothing under
`dsptoolbox_amd/` may import it; only `tests/`, `__graft_entry__.smoke()` and
`bench.py --cpu-baseline` do.

Parity status: PINNED.  Every function below is compared against outputs of
the real reference (imported in the build container by `oracle/gen_golden.py`)
stored in `tests/golden/*.npz`; see `tests/test_oracle_golden.py`.

Each function cites the reference file:line it follows.  Plain ndarrays in,
plain ndarrays out; scalings are passed by *name* ("FFTBackward", ...) so the
oracle does not depend on the product's enums.
"""

from __future__ import annotations

import numpy as np
from scipy.fft import next_fast_len, rfft as sp_rfft
from scipy.signal import oaconvolve
from scipy.signal.windows import get_window

AMPLITUDE_SCALINGS = (
    "AmplitudeSpectrum",
    "AmplitudeSpectralDensity",
    "FFTBackward",
    "FFTForward",
    "FFTOrthogonal",
)
PHYSICAL_SCALINGS = (
    "AmplitudeSpectrum",
    "AmplitudeSpectralDensity",
    "PowerSpectrum",
    "PowerSpectralDensity",
)
ALL_SCALINGS = AMPLITUDE_SCALINGS + ("PowerSpectrum", "PowerSpectralDensity")


# --------------------------------------------------------------------------
# scaling algebra  (dsptoolbox/standard/enums.py:21-229)
# --------------------------------------------------------------------------
def fft_norm(scaling: str) -> str:
    """enums.py:53-75"""
    if scaling == "FFTForward":
        return "forward"
    if scaling == "FFTOrthogonal":
        return "ortho"
    return "backward"


def is_amplitude_scaling(scaling: str) -> bool:
    """enums.py:77-92"""
    return scaling in AMPLITUDE_SCALINGS


def has_physical_units(scaling: str) -> bool:
    """enums.py:109-123"""
    return scaling in PHYSICAL_SCALINGS


def get_scaling_factor(scaling: str, length: int, fs_hz: int, window):
    """enums.py:183-229 (returns an array, like the reference)."""
    if scaling == "FFTBackward":
        return np.atleast_1d(1.0)
    if scaling == "FFTForward":
        return np.atleast_1d(1.0 / length)
    if scaling == "FFTOrthogonal":
        return np.atleast_1d((1.0 / length) ** 0.5)
    if scaling in ("AmplitudeSpectralDensity", "PowerSpectralDensity"):
        if window is None:
            factor = (2 / length / fs_hz) ** 0.5
        else:
            factor = (2 / np.sum(window**2, axis=0, keepdims=True) / fs_hz) ** 0.5
    else:
        if window is None:
            factor = 2**0.5 / length
        else:
            factor = 2**0.5 / np.sum(window, axis=0, keepdims=True)
    if is_amplitude_scaling(scaling):
        return np.atleast_1d(factor)
    return np.atleast_1d(factor**2.0)


def conversion_factor(src: str, dst: str, length: int, fs_hz: int, window):
    """enums.py:141-181"""
    fin = get_scaling_factor(src, length, fs_hz, window).astype(np.float64)
    fout = get_scaling_factor(dst, length, fs_hz, window).astype(np.float64)
    if not (is_amplitude_scaling(src) ^ is_amplitude_scaling(dst)):
        return fout / fin
    if is_amplitude_scaling(src):
        fin = fin**2.0
    else:
        fout = fout**2.0
    return fout / fin


# --------------------------------------------------------------------------
# framing  (helpers/other.py:181-259, standard/_framed_signal_representation.py:9-67)
# --------------------------------------------------------------------------
def compute_number_frames(window_length: int, step: int, signal_length: int,
                          zero_padding: bool):
    """helpers/other.py:181-213"""
    if zero_padding:
        n_frames = int(np.ceil(signal_length / step))
        padding = window_length - int(signal_length % step)
    else:
        padding = 0
        n_frames = int(np.ceil((signal_length - window_length) / step))
    return n_frames, padding


def get_framed_signal(td: np.ndarray, window_length: int, step: int,
                      keep_last_frames: bool = True) -> np.ndarray:
    """_framed_signal_representation.py:9-67 -> (W, F, C) float64 copy."""
    assert td.ndim == 2
    n_frames, pad = compute_number_frames(window_length, step, td.shape[0],
                                          keep_last_frames)
    tdp = np.concatenate([td, np.zeros((pad, td.shape[1]))], axis=0)
    idx = np.arange(window_length)[:, None] + step * np.arange(n_frames)[None, :]
    return tdp[idx, :].astype(np.float64, copy=True)


# --------------------------------------------------------------------------
# Welch  (standard/_spectral_methods.py:10-173)
# --------------------------------------------------------------------------
def welch(x, y, fs_hz: int, window_spec, window_length_samples: int,
          overlap_percent: float, detrend: bool, average: str, scaling: str):
    """Auto (y is None) or cross spectrum.  x,y: (N,) or (N,C)."""
    auto = y is None
    x = np.asarray(x, dtype=np.float64).squeeze()
    if not auto:
        y = np.asarray(y, dtype=np.float64).squeeze()
        assert x.shape == y.shape
    multi = x.ndim == 2
    assert window_length_samples in [2**k for k in range(3, 19)]
    assert 0 <= overlap_percent < 100
    assert average in ("mean", "median")
    window = get_window(window_spec, window_length_samples, fftbins=True)
    overlap = int(overlap_percent / 100 * window_length_samples)  # :106 truncation
    step = window_length_samples - overlap
    if not multi:
        x = x[:, None]
        if not auto:
            y = y[:, None]
    xf = get_framed_signal(x, window_length_samples, step) * window[:, None, None]
    if detrend:  # :136-139, AFTER windowing
        xf = xf - np.mean(xf, axis=0)
    X = np.fft.rfft(xf, axis=0, norm=fft_norm(scaling))
    if auto:
        sp = np.abs(X) ** 2.0
    else:
        yf = get_framed_signal(y, window_length_samples, step) * window[:, None, None]
        if detrend:
            yf = yf - np.mean(yf, axis=0)
        sp = X.conjugate() * np.fft.rfft(yf, axis=0, norm=fft_norm(scaling))
    if average == "mean":
        csd = np.mean(sp, axis=1)
    else:  # :153-162
        csd = np.median(sp.real, axis=1) + 1j * np.median(sp.imag, axis=1)
        n = sp.shape[1] if sp.shape[1] % 2 == 1 else sp.shape[1] - 1
        csd = csd / np.sum((-1) ** (n + 1) / n)
    if has_physical_units(scaling):  # :165-168
        csd = csd * get_scaling_factor(scaling, window_length_samples, fs_hz, window)
        csd[np.array([0, -1]), ...] /= 2
    if is_amplitude_scaling(scaling):  # :170-171
        csd = np.sqrt(csd)
    if not multi:
        csd = csd[:, 0]
    return csd


def compute_transfer_function(y, x, fs_hz: int, window_length_samples: int,
                              mode: str, window_spec="hann",
                              overlap_percent: float = 50.0, detrend: bool = True,
                              average: str = "mean", scaling: str = "FFTBackward"):
    """transfer_functions/transfer_functions.py:419-539.
    y: (N,Cy) output, x: (N,1) or (N,Cy) input.  Returns (tf complex128 (B,Cy),
    coherence float64 (B,Cy)).  Keeps the reference's per-channel loop."""
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    assert x.shape[0] == y.shape[0]
    multichannel = x.shape[1] == 1
    if not multichannel:
        assert x.shape[1] == y.shape[1]
    kw = dict(fs_hz=fs_hz, window_spec=window_spec,
              window_length_samples=window_length_samples,
              overlap_percent=overlap_percent, detrend=detrend, average=average,
              scaling=scaling)
    B = window_length_samples // 2 + 1
    tf = np.zeros((B, y.shape[1]), dtype=np.complex128)
    coh = np.zeros((B, y.shape[1]))
    if multichannel:
        G_xx = welch(x[:, 0], None, **kw)
    with np.errstate(divide="ignore", invalid="ignore"):
        for n in range(y.shape[1]):
            G_yy = welch(y[:, n], None, **kw)
            ni = 0 if multichannel else n
            if not multichannel:
                G_xx = welch(x[:, ni], None, **kw)
            if mode == "H2":
                G_yx = welch(y[:, n], x[:, ni], **kw)
            G_xy = welch(x[:, ni], y[:, n], **kw)
            if mode == "H1":
                tf[:, n] = G_xy / G_xx
            elif mode == "H2":
                tf[:, n] = G_yy / G_yx
            elif mode == "H3":
                tf[:, n] = G_xy / np.abs(G_xy) * (G_yy / G_xx) ** 0.5
            else:
                raise ValueError("Unsupported transfer function type")
            coh[:, n] = np.abs(G_xy) ** 2 / G_xx / G_yy
    return tf, coh


def compute_transfer_function_batched(y, x, fs_hz, window_length_samples, mode,
                                      window_spec="hann", overlap_percent=50.0,
                                      detrend=True, scaling="FFTBackward",
                                      workers=1):
    """Same numbers as compute_transfer_function (mean average only) but with
    one framing + one rFFT batch per signal; used as the multi-core CPU
    baseline.  Checked against the per-channel form in tests."""
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    W = window_length_samples
    window = get_window(window_spec, W, fftbins=True)
    step = W - int(overlap_percent / 100 * W)

    def spec(sig):
        f = get_framed_signal(sig, W, step) * window[:, None, None]
        if detrend:
            f -= np.mean(f, axis=0)
        return sp_rfft(f, axis=0, norm=fft_norm(scaling), workers=workers)

    X, Y = spec(x), spec(y)

    def finish(S):
        if has_physical_units(scaling):
            S = S * get_scaling_factor(scaling, W, fs_hz, window)
            S[np.array([0, -1]), ...] /= 2
        return np.sqrt(S) if is_amplitude_scaling(scaling) else S

    G_xx = finish(np.mean(np.abs(X) ** 2, axis=1))
    G_yy = finish(np.mean(np.abs(Y) ** 2, axis=1))
    G_xy = finish(np.mean(X.conj() * Y, axis=1).astype(np.complex128))
    with np.errstate(divide="ignore", invalid="ignore"):
        if mode == "H1":
            tf = G_xy / G_xx
        elif mode == "H2":
            tf = G_yy / finish(np.mean(Y.conj() * X, axis=1).astype(np.complex128))
        elif mode == "H3":
            tf = G_xy / np.abs(G_xy) * (G_yy / G_xx) ** 0.5
        else:
            raise ValueError("Unsupported transfer function type")
        coh = np.abs(G_xy) ** 2 / G_xx / G_yy
    return tf, coh


# --------------------------------------------------------------------------
# STFT  (standard/_spectral_methods.py:176-282)
# --------------------------------------------------------------------------
def stft(x, fs_hz: int, window_length_samples: int, window_spec,
         overlap_percent: float, fft_length_samples, detrend: bool,
         padding: bool, scaling: str):
    x = np.asarray(x, dtype=np.float64)
    assert window_length_samples in [2**k for k in range(4, 17)]
    assert 0 <= overlap_percent < 100
    if fft_length_samples is None:
        fft_length_samples = window_length_samples
    window = get_window(window_spec, window_length_samples, fftbins=True)
    overlap = int(overlap_percent / 100 * window_length_samples + 0.5)  # :247 rounding
    step = window_length_samples - overlap
    if padding:
        x = np.pad(x, ((overlap, overlap), (0, 0)))
    tx = get_framed_signal(x, window_length_samples, step, True)
    tx *= window[:, None, None]
    if detrend:
        tx -= np.mean(tx, axis=0)
    out = np.fft.rfft(tx, axis=0, n=fft_length_samples, norm=fft_norm(scaling))
    if has_physical_units(scaling):  # :271-278
        out[0, ...] /= 2**0.5
        if fft_length_samples % 2 == 0:
            out[-1, ...] /= 2**0.5
        factor = get_scaling_factor(scaling, fft_length_samples, fs_hz, window)
        if not is_amplitude_scaling(scaling):
            out = np.abs(out) ** 2.0
        out = out * factor
    time_s = np.linspace(0, len(x) / fs_hz, out.shape[1])
    freqs_hz = np.fft.rfftfreq(len(window), 1 / fs_hz)
    return time_s, freqs_hz, out


# --------------------------------------------------------------------------
# inverse STFT  (transforms/transforms.py:444-586, standard/_framed_signal_representation.py:
# 70-137, standard/_standard_backend.py:408-427)
# --------------------------------------------------------------------------
def pad_trim(td, desired_length: int):
    """helpers/other.py:216-259 for a (N, C) array, padding / trimming at the end."""
    n = td.shape[0]
    if n >= desired_length:
        return td[:desired_length].copy()
    return np.concatenate([td, np.zeros((desired_length - n, td.shape[1]), dtype=td.dtype)])


def reconstruct_framed_signal(td_framed, step_size: int, window, safety_threshold=1e-4):
    """Overlap-add of windowed frames divided by the squared-window envelope (clipped)."""
    td_framed = td_framed * window[:, None, None]
    W, F = td_framed.shape[0], td_framed.shape[1]
    total_length = int(step_size * F + W * (1 - step_size / W))
    td = np.zeros((total_length, td_framed.shape[-1]))
    envelope = np.zeros(total_length)
    start = 0
    for i in range(F):
        td[start:start + W, :] += td_framed[:, i, :]
        envelope[start:start + W] += window**2
        start += step_size
    envelope = np.clip(envelope, a_min=safety_threshold, a_max=None)
    return td / envelope[:, None]


def istft(stft_data, fs_hz, window_length_samples: int, window_spec, overlap_percent: float,
          fft_length_samples, padding: bool, scaling: str, original_length=None):
    """transforms.istft; original_length: the `original_signal` branch (_pad_trim to it)."""
    window = get_window(window_spec, window_length_samples)
    td_framed = np.fft.irfft(stft_data, axis=0, n=fft_length_samples, norm=fft_norm(scaling))
    td_framed = td_framed[:window_length_samples, ...]
    if has_physical_units(scaling):
        td_framed = td_framed / get_scaling_factor(scaling, fft_length_samples, fs_hz, window)
    step = int((1 - overlap_percent / 100) * len(window))
    if padding:
        td = reconstruct_framed_signal(td_framed, step, window)
        overlap = int(overlap_percent / 100 * len(window))
        td = td[overlap:-overlap, :]
    else:
        extra = np.zeros_like(td_framed[:, 0, :])[:, None, :]
        td_framed = np.append(np.append(extra, td_framed, axis=1), extra, axis=1)
        td = reconstruct_framed_signal(td_framed, step, window)
        td = td[step:-step, :]
    if original_length is not None:
        td = pad_trim(td, original_length)
    return td


# --------------------------------------------------------------------------
# CSM  (standard/_spectral_methods.py:285-443)
# --------------------------------------------------------------------------
def csm_welch(td, fs_hz: int, window_length_samples: int, window_spec,
              overlap_percent: float, detrend: bool, average: str, scaling: str):
    """Reference-faithful pair loop (:351-369)."""
    td = np.asarray(td, dtype=np.float64)
    C = td.shape[1]
    csm = np.zeros((window_length_samples // 2 + 1, C, C), dtype=np.complex128)
    for i1 in range(C):
        for i2 in range(i1, C):
            csm[:, i2, i1] = welch(td[:, i1], td[:, i2] if i1 != i2 else None,
                                   fs_hz, window_spec, window_length_samples,
                                   overlap_percent, detrend, average, scaling)
            if i1 == i2:
                csm[:, i1, i2] *= 0.5
    csm += np.swapaxes(csm, 1, 2).conjugate()
    f = np.fft.rfftfreq(window_length_samples, 1 / fs_hz)
    return f, csm


def csm_welch_batched(td, fs_hz, window_length_samples, window_spec,
                      overlap_percent, detrend, scaling, workers=1):
    """(1/F) sum_f x x^H per bin + the same finish as welch(); equals csm_welch
    (mean average) -- checked in tests.  Multi-core CPU baseline."""
    td = np.asarray(td, dtype=np.float64)
    W = window_length_samples
    window = get_window(window_spec, W, fftbins=True)
    step = W - int(overlap_percent / 100 * W)
    fr = get_framed_signal(td, W, step) * window[:, None, None]
    if detrend:
        fr -= np.mean(fr, axis=0)
    X = sp_rfft(fr, axis=0, norm=fft_norm(scaling), workers=workers)  # (B,F,C)
    # csm[b, i2, i1] = mean_f conj(X[b,f,i1]) X[b,f,i2]
    S = np.einsum("bfi,bfj->bji", X.conj(), X) / X.shape[1]
    if has_physical_units(scaling):
        S = S * get_scaling_factor(scaling, W, fs_hz, window)
        S[np.array([0, -1]), ...] /= 2
    if is_amplitude_scaling(scaling):
        # the reference takes sqrt of the lower triangle (i2 >= i1) and mirrors
        # its conjugate; for a Hermitian S this equals the element-wise
        # principal root except on the negative real axis.
        low = np.tril(np.ones(S.shape[1:], dtype=bool))
        R = np.where(low[None], np.sqrt(S), 0)
        d = np.einsum("bii->bi", R).copy()
        R = R + np.swapaxes(R, 1, 2).conjugate()
        np.einsum("bii->bi", R)[...] = d
        S = R
    return np.fft.rfftfreq(W, 1 / fs_hz), S


def csm_fft(spectrum, scaling: str, window, fs_hz: int):
    """_spectral_methods.py:374-443; `spectrum` is the FFTBackward rFFT (B,C)."""
    spectrum = np.asarray(spectrum, dtype=np.complex128)
    C = spectrum.shape[1]
    csm = np.zeros((spectrum.shape[0], C, C), dtype=np.complex128)
    for i1 in range(C):
        for i2 in range(i1, C):
            csm[:, i2, i1] = spectrum[:, i1].conjugate() * spectrum[:, i2]
            if i1 == i2:
                csm[:, i1, i2] *= 0.5
    csm += np.swapaxes(csm, 1, 2).conjugate()
    if scaling == "FFTBackward":
        return csm
    csm[np.array([0, -1]), ...] /= 2.0
    # NOTE reference passes `spectrum.shape[0] // 2 + 1` as the length (:436)
    factor = conversion_factor("FFTBackward", scaling, spectrum.shape[0] // 2 + 1,
                               fs_hz, window)[:, None]
    factor = np.repeat(factor, C, axis=-1)
    csm *= factor[None, ...]
    if is_amplitude_scaling(scaling):
        csm = np.sqrt(csm)
    return csm


# --------------------------------------------------------------------------
# whole-signal spectrum (classes/signal.py:899-938, helpers/spectrum_utilities.py:268-328)
# --------------------------------------------------------------------------
def spectrum_fft(td, fs_hz: int, scaling: str = "FFTBackward",
                 pad_to_fast_length: bool = True):
    td = np.asarray(td, dtype=np.float64)
    n = next_fast_len(td.shape[0], True) if pad_to_fast_length else td.shape[0]
    sp = sp_rfft(td, axis=0, norm=fft_norm(scaling), n=n)
    if has_physical_units(scaling):
        factor = get_scaling_factor(scaling, n, fs_hz, None)
        sp[0] /= 2**0.5
        if n % 2 == 0:
            sp[-1] /= 2**0.5
        if not is_amplitude_scaling(scaling):
            sp = np.abs(sp) ** 2
        sp = sp * factor
    return np.fft.rfftfreq(n, 1 / fs_hz), sp


# --------------------------------------------------------------------------
# spectral deconvolution
# --------------------------------------------------------------------------
def to_db(x, amplitude_input: bool):
    """helpers/gain_and_level.py:158-200 (default clipping)."""
    factor = 20.0 if amplitude_input else 10.0
    tiny = float(np.finfo(np.float64).smallest_normal)
    return factor * np.log10(np.clip(np.abs(x), a_min=tiny, a_max=None))


def find_frequencies_above_threshold(spec, f, threshold_db):
    """helpers/other.py:34-41"""
    d = to_db(spec, True)
    d = d - np.max(d)
    fr = f[d > threshold_db]
    return [fr[0], fr[-1]]


def find_nearest_points_index_in_vector(points, vector):
    """helpers/other.py:9-31"""
    points = np.atleast_1d(np.array(points))
    return np.array([np.argmin(np.abs(p - vector)) for p in points], dtype=np.int_)


def tukey_like_window_hann(points, window_length: int, inverse: bool):
    """helpers/windows.py:8-76 with Window.Hann, at_start=True."""
    i0, i1, i2, i3 = [int(i) for i in points]
    nl = i1 - i0
    low = get_window("hann", nl * 2, fftbins=True)[:nl] if nl > 0 else np.ones(nl)
    nh = i3 - i2
    high = get_window("hann", nh * 2, fftbins=True)[nh:] if nh > 1 else np.ones(nh)
    w = np.concatenate((np.zeros(i0), low, np.ones(i2 - i1), high,
                        np.zeros(window_length - i3)))
    return 1 - w if inverse else w


def regularization_eps(denum_fft_ch0, freqs_hz, fs_hz, start_stop_hz, threshold_db):
    """transfer_functions.py:152-167 + _transfer_functions.py:31-35.
    Returns (eps (B,), start_stop_hz (4,))."""
    if start_stop_hz is None:
        start_stop_hz = find_frequencies_above_threshold(denum_fft_ch0, freqs_hz,
                                                         threshold_db)
    if len(start_stop_hz) == 2:
        start_stop_hz = np.array([
            start_stop_hz[0] / np.sqrt(2), start_stop_hz[0], start_stop_hz[1],
            np.min([start_stop_hz[1] * np.sqrt(2), fs_hz / 2])])
    elif len(start_stop_hz) != 4:
        raise ValueError("start_stop_hz vector should have 2 or 4 values")
    ids = find_nearest_points_index_in_vector(start_stop_hz, freqs_hz)
    eps = tukey_like_window_hann(ids, len(freqs_hz), True) * 10 ** (30 / 20)
    return eps, np.asarray(start_stop_hz)


def spectral_deconvolve(y, x, fs_hz: int, apply_regularization: bool = True,
                        start_stop_hz=None, threshold_db: float = -30.0,
                        padding: bool = False, keep_original_length: bool = False,
                        scaling_y: str = "FFTBackward", scaling_x: str = "FFTBackward"):
    """transfer_functions.py:61-184 (pad_to_fast_length=True).  y (N,C); x (N,1|C) -> (N|2N, C).
    scaling_y / scaling_x: the spectrum scaling each Signal carries -- only the METHOD is forced to FFT
    (:142-143), get_spectrum still applies the scaling (classes/signal.py:899-938)."""
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    assert y.shape[0] == x.shape[0]
    multichannel = x.shape[1] == 1
    if not multichannel:
        assert x.shape[1] == y.shape[1]
    if not apply_regularization:
        assert start_stop_hz is None
    N0 = y.shape[0]
    if padding:
        y = np.concatenate([y, np.zeros_like(y)], axis=0)
        x = np.concatenate([x, np.zeros_like(x)], axis=0)
    freqs, den = spectrum_fft(x, fs_hz, scaling_x)
    _, num = spectrum_fft(y, fs_hz, scaling_y)
    nt = y.shape[0]
    out = np.zeros_like(y)
    eps = None
    for n in range(y.shape[1]):
        nd = 0 if multichannel else n
        if apply_regularization:
            if eps is None:  # band derived ONCE, from the first denominator used
                eps, start_stop_hz = regularization_eps(den[:, nd], freqs, fs_hz,
                                                        start_stop_hz, threshold_db)
            reg = den[:, nd].conj() / (np.abs(den[:, nd]) ** 2 + eps)
            out[:, n] = np.fft.irfft(num[:, n] * reg, n=nt)
        else:
            out[:, n] = np.fft.irfft(np.divide(num[:, n], den[:, nd]), n=nt)
    if padding and keep_original_length:
        out = out[:N0]
    return out


# --------------------------------------------------------------------------
# FIR filtering (classes/filter_helpers.py:288-503)
# --------------------------------------------------------------------------
def lfilter_fir(b, x, zi=None):
    """filter_helpers.py:454-503: oaconvolve(...)[:N]; with zi (T-1, C) the state is added to
    the head of the full convolution and the new state is its tail (:493-500)."""
    b = np.asarray(b)
    b = b.astype(np.complex128 if np.iscomplexobj(b) else np.float64)  # complex taps: complex output (:364-371)
    if b.ndim != 1:  # :475-477 (a one-tap filter is already 1-D and stays so)
        b = np.squeeze(b)
        assert b.ndim == 1, "FIR Filters for audio must be 1D-arrays"
    x = np.asarray(x, dtype=np.float64)
    if x.ndim < 2:
        x = x[:, None]
        if zi is not None:
            zi = np.asarray(zi)[:, None]
    y = oaconvolve(x, b[:, None], mode="full", axes=0)
    if zi is None:
        return y[: x.shape[0], :]
    y[: zi.shape[0], :] += zi
    zf = y[-zi.shape[0]:, :]
    return y[: x.shape[0], :], zf


def lfilter_zi_fir(b):
    """scipy.signal.lfilter_zi(b, [1.0]) (Filter.initialize_zi, filter.py:331-353) in closed
    form: the step-response steady state of a transposed direct-form FIR filter is the tail
    sum of the taps, zi[i] = sum_{j > i} b[j]."""
    b = np.asarray(b)
    b = b.astype(np.complex128 if np.iscomplexobj(b) else np.float64)
    return np.cumsum(b[::-1])[::-1][1:].copy()


def filtfilt_fir(b, x):
    """scipy.signal.filtfilt(b, [1.0], x, axis=0) with its defaults (padtype="odd",
    padlen = 3 * len(b), method="pad") -- the zero_phase branch of _filter_on_signal_ba
    (filter_helpers.py:362-363) -- restated on top of lfilter_fir."""
    b = np.asarray(b, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    edge = 3 * len(b)
    if x.shape[0] <= edge:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
    ext = np.concatenate([2 * x[:1] - x[edge:0:-1], x, 2 * x[-1:] - x[-2:-(edge + 2):-1]], axis=0)
    zi = lfilter_zi_fir(b)[:, None]
    y, _ = lfilter_fir(b, ext, zi * ext[:1])
    y, _ = lfilter_fir(b, y[::-1], zi * y[-1:])
    return y[::-1][edge:-edge]


def filter_fir_on_channels(b, td, channels=None):
    """filter_helpers.py:288-382 FIR branch: selected channels filtered, the
    rest bypassed."""
    td = np.asarray(td, dtype=np.float64)
    out = td.copy()
    if channels is None:
        channels = np.arange(td.shape[1])
    channels = np.atleast_1d(channels)
    out[:, channels] = lfilter_fir(b, td[:, channels])
    return out


def filterbank_fir(taps_list, td, mode: str):
    """filter_helpers.py:385-451.  Parallel -> (N, C, K) stack of band outputs;
    Sequential / Summed -> (N, C)."""
    td = np.asarray(td, dtype=np.float64)
    if mode == "Parallel":
        return np.stack([lfilter_fir(b, td) for b in taps_list], axis=-1)
    if mode == "Sequential":
        out = td.copy()
        for b in taps_list:
            out = lfilter_fir(b, out)
        return out
    if mode == "Summed":
        acc = np.zeros((td.shape[0], td.shape[1], len(taps_list)))
        for n, b in enumerate(taps_list):
            acc[:, :, n] = lfilter_fir(b, td)
        return np.sum(acc, axis=-1)
    raise ValueError("Invalid filter bank apply mode")


def fir_transfer_function(b, frequency_vector_hz, fs_hz: int):
    """Filter.get_transfer_function for a FIR filter (classes/filter.py:893-900):
    scipy.signal.freqz(b, 1, worN=f, fs=fs)[1] = polyval of the taps in z^-1 = exp(-2 pi i f / fs)."""
    from scipy.signal import freqz
    return freqz(np.asarray(b), [1.0], np.asarray(frequency_vector_hz, dtype=np.float64), fs=fs_hz)[1]


def filter_get_ir(b, length_samples: int, zero_phase: bool = False):
    """Filter.get_ir (classes/filter.py:818-860) for a FIR filter: the padded taps (a length below the tap count is
    raised to it), or a unit impulse through the zero-phase filtering.  -> (L, 1)"""
    b = np.asarray(b, dtype=np.float64)
    if not zero_phase:
        return pad_trim(b[:, None], max(length_samples, len(b)))
    d = np.zeros((length_samples, 1))
    d[0] = 1.0
    return filtfilt_fir(b, d)


def _constrained(td):
    """What the time_data setter of a Signal with constrain_amplitude=True keeps (classes/signal.py:273-292): data above
    0 dBFS is divided by its peak."""
    peak = np.max(np.abs(td))
    return td / peak if peak > 1.0 else td


def filterbank_get_ir(taps_list, length_samples: int, mode: str, zero_phase: bool = False):
    """FilterBank.get_ir (classes/filterbank.py:534-613): a one-channel unit impulse through the bank; a length below the
    highest order becomes that order + 100.  The impulse is an ImpulseResponse with constrain_amplitude=True
    (generators.dirac, generators.py:310-313), and so is every Signal made from it on the way
    (copy_with_new_time_data): an output above 0 dBFS comes back divided by its peak -- per band, and after every
    stage of the sequential mode.  Parallel -> (L, 1, K), else (L, 1)."""
    max_order = max(len(b) - 1 for b in taps_list)
    if max_order > length_samples:
        length_samples = max_order + 100
    d = np.zeros((length_samples, 1))
    d[0] = 1.0
    f = (lambda b, x: filtfilt_fir(b, x)) if zero_phase else (lambda b, x: lfilter_fir(b, x))
    if mode == "Parallel":
        return np.stack([_constrained(f(b, d)) for b in taps_list], axis=-1)
    if mode == "Sequential":
        out = d
        for b in taps_list:
            out = _constrained(f(b, out))
        return out
    return _constrained(np.sum(np.stack([_constrained(f(b, d)) for b in taps_list], axis=-1), axis=-1))


def filterbank_transfer_function(taps_list, frequency_vector_hz, fs_hz: int, mode: str):
    """FilterBank.get_transfer_function (classes/filterbank.py:615-655): Parallel (frequency, filter); Sequential the
    product; Summed ONE plus the sum (the reference initialises its sum with ones, :649)."""
    h = np.stack([fir_transfer_function(b, frequency_vector_hz, fs_hz) for b in taps_list], axis=1)
    if mode == "Parallel":
        return h
    if mode == "Sequential":
        return np.prod(h, axis=1)
    if mode == "Summed":
        return 1.0 + np.sum(h, axis=1)
    raise ValueError("No valid mode")


def filter_multiband(taps_list, bands, zero_phase: bool = False):
    """FilterBank.filter_multiband_signal (classes/filterbank.py:479-532): band n through filter n.  -> (N, K, C)"""
    f = filtfilt_fir if zero_phase else lfilter_fir
    return np.stack([f(b, np.asarray(td, dtype=np.float64)) for b, td in zip(taps_list, bands)], axis=1)


def add_channel(td, new_td):
    """Signal.add_channel (classes/signal.py:812-852): the new data as (samples, channels) -- transposed when it has more
    columns than rows --, zero-padded or trimmed at its end to the signal's length, appended as new channels."""
    new_td = np.array(new_td, dtype=np.float64)
    if new_td.ndim < 2:
        new_td = new_td[:, None]
    if new_td.shape[1] > new_td.shape[0]:
        new_td = new_td.T
    return np.concatenate([np.asarray(td, dtype=np.float64), pad_trim(new_td, td.shape[0])], axis=1)


def mel_filterbank(f_hz, range_hz=None, n_bands=40, normalize=True):
    """transforms/transforms.py:206-277"""
    f_hz = np.squeeze(f_hz)
    if range_hz is None:
        range_hz = f_hz[[0, -1]]
    range_hz = np.sort(np.atleast_1d(np.asarray(range_hz, dtype=np.float64).squeeze()))
    range_mel = 2595 * np.log10(1 + range_hz / 700)
    centers = np.linspace(range_mel[0], range_mel[1], n_bands + 2, endpoint=True)
    bands_hz = 700 * (10 ** (centers / 2595) - 1)
    inds = np.array([np.argmin(np.abs(b - f_hz)) for b in bands_hz], dtype=int)
    m = np.zeros((n_bands, len(f_hz)))
    for n in range(n_bands):
        ni = n + 1
        m[n, inds[ni - 1]:inds[ni]] = np.linspace(0, 1, inds[ni] - inds[ni - 1], endpoint=False)
        m[n, inds[ni]:inds[ni + 1]] = np.linspace(1, 0, inds[ni + 1] - inds[ni], endpoint=False)
        if normalize:
            m[n, :] /= np.sum(m[n, :])
    return m, centers[1:-1]


def to_db_power(x):
    """helpers/gain_and_level.py:158-200 with amplitude_input=False and the default floor."""
    return 10.0 * np.log10(np.clip(np.abs(x), a_min=np.finfo(np.float64).smallest_normal, a_max=None))


def log_mel_spectrogram(stft_data, f_hz, range_hz=None, n_bands=40):
    """transforms/transforms.py:177-184 on a given spectrogram (B, F, C)."""
    mfilt, f_mel = mel_filterbank(f_hz, range_hz, n_bands, normalize=True)
    return f_mel, to_db_power(np.tensordot(mfilt, np.abs(stft_data) ** 2.0, axes=(-1, 0)))


def mfcc(stft_data, f_hz, mel_filters=None):
    """transforms/transforms.py:413-429"""
    from scipy.fft import dct
    if mel_filters is None:
        mel_filters, f_mel = mel_filterbank(f_hz, None, n_bands=40)
    else:
        f_mel = np.array([0, mel_filters.shape[0]])
    sp = np.tensordot(mel_filters, np.abs(stft_data) ** 2.0, axes=(-1, 0))
    out = np.abs(dct(to_db_power(sp), type=2, axis=0))
    np.nan_to_num(out, copy=False, nan=0)
    return f_mel, out


def chroma_stft(stft_data, f_hz, tuning_a_hz=440, compression=0.5):
    """transforms/transforms.py:640-665 on a given spectrogram (B, F, C): pitch bands of a quarter
    tone around the 128 MIDI pitches (transforms/_transforms.py:10-26), summed over octaves,
    log(1 + compression * .).  -> (chroma (12, F, C), pitch (128, F, C))."""
    en = np.abs(stft_data) ** 2
    pitch_f = tuning_a_hz * 2 ** ((np.arange(128) - 69) / 12)
    pt = np.zeros((128, len(f_hz)))
    for i, fn in enumerate(pitch_f):
        pt[i, (f_hz >= fn * 2 ** (-1 / 24)) & (f_hz < fn * 2 ** (1 / 24))] = 1
    ct = np.zeros((12, 128))
    for i in range(12):
        ct[i, i::12] = 1
    pitch = np.tensordot(pt, en, (1, 0))
    chroma = np.tensordot(ct, pitch, (1, 0))
    return np.log(1 + compression * chroma), np.log(1 + compression * pitch)


class FIRFilterOverlapSave:
    """classes/fir_filter_realtime.py:75-157: one FFT block of next_fast_len(T + blocksize).
    Literal, including the defect that irfft is called without its length: when
    next_fast_len returns an ODD length L the inverse transform has L - 1 points and the block
    output is not the convolution (the reference's test, tests/test_classes.py:1527-1554, uses a
    48 000-tap response + 512, whose fast length is even)."""

    def __init__(self, b):
        self.fir = np.asarray(b, dtype=np.float64)

    def prepare(self, blocksize_samples, n_channels):
        import scipy.fft as sfft
        self.blocksize = blocksize_samples
        self.total_length = sfft.next_fast_len(len(self.fir) + blocksize_samples, True)
        self.fir_spectrum = np.fft.rfft(self.fir, n=self.total_length)
        self.buffer = np.zeros((self.total_length, n_channels))

    def process_block(self, block, channel):
        self.buffer[-self.blocksize:, channel] = block
        out = np.fft.irfft(np.fft.rfft(self.buffer[:, channel]) * self.fir_spectrum)
        self.buffer[:-self.blocksize, channel] = self.buffer[self.blocksize:, channel].copy()
        return out[-self.blocksize:]


class FIRUniformPartitioned:
    """classes/fir_filter_realtime.py:160-240: uniform partitions of blocksize taps and a
    frequency-domain delay line of the last n_partitions input spectra.  Literal, including the
    ONE delay-line index shared by all channels (:227-229): driven channel after channel it
    advances n_channels slots per block, so with several channels the result is the convolution
    only when n_channels = 1 (mod n_partitions)."""

    def __init__(self, fir):
        self.fir = np.asarray(fir, dtype=np.float64)

    def prepare(self, blocksize_samples, n_channels):
        B = self.blocksize = blocksize_samples
        self.n_partitions = len(self.fir) // B + 1
        part = np.zeros((B, self.n_partitions))
        for n in range(self.n_partitions):
            seg = self.fir[n * B:(n + 1) * B]
            part[:len(seg), n] = seg
        self.partitioned_spectrum = np.fft.rfft(part, axis=0, n=2 * B)
        self.buffer_ind = 0
        self.buffer_spectra = np.zeros((B + 1, self.n_partitions, n_channels), dtype=np.complex128)
        self.input_buffer = np.zeros((2 * B, n_channels))

    def process_block(self, block, channel):
        B = self.blocksize
        self.input_buffer[:B, channel] = self.input_buffer[-B:, channel].copy()
        self.input_buffer[-B:, channel] = block
        self.buffer_spectra[:, self.buffer_ind, channel] = np.fft.rfft(self.input_buffer[:, channel])
        out = np.sum(self.partitioned_spectrum
                     * self.buffer_spectra[:, self.buffer_ind - np.arange(self.n_partitions), channel], axis=1)
        self.buffer_ind = (self.buffer_ind + 1) % self.n_partitions
        return np.fft.irfft(out)[-B:]


class FIRUniformPartitionedMultichannel:
    """classes/fir_filter_realtime.py:243-335: one impulse response per channel, all channels per
    call.  The constructor passes the coefficients through Signal.from_time_data (:262), whose
    default constrain_amplitude scales them to a peak of 1 when the peak exceeds 1
    (classes/signal.py:222-301)."""

    def __init__(self, fir):
        fir = np.asarray(fir, dtype=np.float64)
        fir = fir.reshape(-1, 1) if fir.ndim == 1 else fir
        peak = np.max(np.abs(fir))
        self.fir = fir / peak if peak > 1 else fir.copy()

    def prepare(self, blocksize_samples):
        B = self.blocksize = blocksize_samples
        self.n_partitions = self.fir.shape[0] // B + 1
        C = self.fir.shape[1]
        part = np.zeros((B, self.n_partitions, C))
        for n in range(self.n_partitions):
            seg = self.fir[n * B:(n + 1) * B]
            part[:len(seg), n, :] = seg
        self.partitioned_spectrum = np.fft.rfft(part, axis=0, n=2 * B)
        self.buffer_ind = 0
        self.buffer_spectra = np.zeros((B + 1, self.n_partitions, C), dtype=np.complex128)
        self.input_buffer = np.zeros((2 * B, C))

    def process_block(self, block):
        B = self.blocksize
        self.input_buffer[:B] = self.input_buffer[-B:].copy()
        self.input_buffer[-B:] = block
        self.buffer_spectra[:, self.buffer_ind] = np.fft.rfft(self.input_buffer, axis=0)
        out = np.sum(self.partitioned_spectrum
                     * self.buffer_spectra[:, self.buffer_ind - np.arange(self.n_partitions), ...], axis=1)
        self.buffer_ind = (self.buffer_ind + 1) % self.n_partitions
        return np.fft.irfft(out, axis=0)[-B:]


def das_map(f, csm, h, remove_csm_diagonal=True):
    """beamforming/beamforming.py:838-876: Re(h^H CSM h) per grid point and bin, optional
    diagonal removal (energy-compensated), clipping of negative values, Simpson integration."""
    from scipy.integrate import simpson
    csm = np.array(csm, dtype=np.complex128)
    h = np.asarray(h, dtype=np.complex128)
    C = csm.shape[1]
    if remove_csm_diagonal:
        csm *= C / (C - 1)
        for i in range(len(f)):
            np.fill_diagonal(csm[i], 0)
    m = np.einsum("fcg,fcd,fdg->gf", h.conj(), csm, h).real
    if remove_csm_diagonal:
        m[m < 0] = 0
    return simpson(m, dx=f[1] - f[0], axis=1) if len(f) > 1 else m.squeeze()


def convolve_rir_on_signal(td, rir, keep_peak_level=True, keep_length=True):
    """room_acoustics/room_acoustics.py:216-266 (oaconvolve and convolve agree to rounding)."""
    td = np.asarray(td, dtype=np.float64)
    out = oaconvolve(td, np.asarray(rir, dtype=np.float64).reshape(-1, 1), axes=0, mode="full")
    if keep_length:
        out = out[: td.shape[0], ...]
    if keep_peak_level:
        out = out * (np.max(np.abs(td), axis=0) / np.max(np.abs(out), axis=0))[None, ...]
    return out


# --------------------------------------------------------------------------
# parity metrics (max-norm relative and relative L2)
# --------------------------------------------------------------------------
def rel_max(a, b) -> float:
    a = np.asarray(a)
    b = np.asarray(b)
    den = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b))) / (den if den > 0 else 1.0)


def rel_l2(a, b) -> float:
    a = np.asarray(a)
    b = np.asarray(b)
    den = float(np.linalg.norm(b.ravel()))
    return float(np.linalg.norm((a - b).ravel())) / (den if den > 0 else 1.0)
