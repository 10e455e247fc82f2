"""compute_transfer_function and spectral_deconvolve
(API mirror of dsptoolbox/transfer_functions/transfer_functions.py:419-539 and
:61-184; regularisation window helpers/windows.py:8-76, band detection
helpers/other.py:9-41)."""

import numpy as np
from scipy.fft import next_fast_len
from scipy.signal.windows import get_window

from .. import backend
from ..classes import ImpulseResponse, Signal, Spectrum
from ..standard.enums import SpectrumMethod
from .enums import TransferFunctionType

__all__ = ["compute_transfer_function", "spectral_deconvolve", "TransferFunctionType"]


def compute_transfer_function(output: Signal, input: Signal, window_length_samples: int,
                              mode: TransferFunctionType = TransferFunctionType.H2) -> Spectrum:
    """Welch H1 / H2 / H3 transfer functions with coherence.  A one-channel
    input is the input of every output channel.  Window, overlap, detrend,
    average and scaling are taken from the INPUT signal's spectrum parameters."""
    assert input.sampling_rate_hz == output.sampling_rate_hz, "Sampling rates do not match"
    assert len(input) == len(output), "Signal lengths do not match"
    if input.number_of_channels != 1:
        assert input.number_of_channels == output.number_of_channels, \
            "Channel number does not match between signals"
    par = input._spectrum_parameters.copy()
    assert type(par) is dict, "Spectrum parameters should be passed as a dictionary"
    for k in ("window_length_samples", "method", "smoothing", "pad_to_fast_length"):
        par.pop(k)
    if not isinstance(mode, TransferFunctionType):
        raise ValueError("Unsupported transfer function type")
    W = int(window_length_samples)
    if output.on_device and not output.is_complex_signal and not input.is_complex_signal:
        # device-resident samples (Signal.to_device / from_planar_f32): the fp32 kernels read them in place -- unless
        # the precision rule sends this shape through the float64 kernels, which take the host arrays
        window = backend._window_array(par["window_type"], W) if W in [2**k for k in range(3, 19)] else None
        if window is not None:
            _, n_frames = backend._welch_framing(output.length_samples, W, par["overlap_percent"], window)
            if not backend._tf_x64_applies(backend.TF_PRECISION, input.number_of_channels, output.number_of_channels,
                                           n_frames, W, par["average"]):
                tf, coherence = backend.welch_transfer_function_device(
                    output.device_samples, input.to_device().device_samples, input.sampling_rate_hz, W, mode.name,
                    narrow=True, **par)
                return Spectrum._from_device_result(np.fft.rfftfreq(W, 1 / input.sampling_rate_hz), tf, coherence)
    # small problems run in float64 end to end like the reference (backend.TF_PRECISION)
    tf, coherence = backend.welch_transfer_function(
        output.time_data, input.time_data, input.sampling_rate_hz, window_length_samples,
        mode.name, precision=backend.TF_PRECISION, **par)
    spec = Spectrum(np.fft.rfftfreq(window_length_samples, 1 / input.sampling_rate_hz), tf)
    spec.set_coherence(coherence)
    return spec


# ---- spectral deconvolution --------------------------------------------------
def _to_db_amplitude(x):
    tiny = float(np.finfo(np.float64).smallest_normal)
    return 20.0 * np.log10(np.clip(np.abs(x), a_min=tiny, a_max=None))


def find_frequencies_above_threshold(spec, f, threshold_db, normalize=True):
    d = _to_db_amplitude(spec)
    if normalize:
        d = d - np.max(d)
    fr = f[d > threshold_db]
    return [fr[0], fr[-1]]


def find_nearest_points_index_in_vector(points, vector):
    points = np.atleast_1d(np.array(points))
    return np.array([np.argmin(np.abs(p - vector)) for p in points], dtype=np.int_)


def _inverse_hann_band(ids, length: int):
    """1 - (0 .. Hann rise .. 1 .. Hann fall .. 0) over the four bin indices."""
    i0, i1, i2, i3 = [int(i) for i in ids]
    nl, nh = i1 - i0, i3 - i2
    low = get_window("hann", nl * 2, fftbins=True)[:nl] if nl > 0 else np.ones(nl)
    high = get_window("hann", nh * 2, fftbins=True)[nh:] if nh > 1 else np.ones(nh)
    w = np.concatenate((np.zeros(i0), low, np.ones(i2 - i1), high, np.zeros(length - i3)))
    return 1 - w


def _spectral_deconvolve_scaled(output, input, apply_regularization, start_stop_hz, threshold_db, padding,
                                keep_original_length, multichannel):
    """spectral_deconvolve for signals whose spectrum parameters ask for a scaling other than the plain
    transform (FFTForward / FFTOrthogonal norms, amplitude or power spectra in physical units): the
    reference divides whatever `get_spectrum` returns (transfer_functions.py:142-177, classes/signal.py:
    899-938 -- power scalings make both spectra real, |X|^2 k).  Transforms on the device
    (Signal.get_spectrum), the B x C division in float64 on the host, the inverse transform of any length
    on the device."""
    fs_hz = output.sampling_rate_hz
    original_length = output.time_data.shape[0]
    n_time = original_length * 2 if padding else original_length

    def spectrum_of(sig):
        td = sig.time_data
        if padding:
            td = np.concatenate((td, np.zeros_like(td)), axis=0)
        tmp = Signal(None, td, fs_hz)
        tmp._spectrum_parameters = dict(sig._spectrum_parameters)
        tmp.spectrum_method = SpectrumMethod.FFT
        return tmp.get_spectrum()

    _, denum_fft = spectrum_of(input)
    freqs_hz, num_fft = spectrum_of(output)
    if apply_regularization:
        if start_stop_hz is None:
            start_stop_hz = find_frequencies_above_threshold(denum_fft[:, 0], freqs_hz, threshold_db)
        if len(start_stop_hz) == 2:
            start_stop_hz = np.array([start_stop_hz[0] / np.sqrt(2), start_stop_hz[0], start_stop_hz[1],
                                      np.min([start_stop_hz[1] * np.sqrt(2), fs_hz / 2])])
        elif len(start_stop_hz) != 4:
            raise ValueError("start_stop_hz vector should have 2 or 4 values")
        ids = find_nearest_points_index_in_vector(start_stop_hz, freqs_hz)
        eps = _inverse_hann_band(ids, len(freqs_hz)) * 10 ** (30 / 20)
        den = denum_fft[:, :1] if multichannel else denum_fft
        prod = num_fft * (np.conj(den) / (np.abs(den) ** 2 + eps[:, None]))
    else:
        prod = num_fft / (denum_fft[:, :1] if multichannel else denum_fft)
    # np.fft.irfft(prod, n=n_time): the spectrum is cropped or zero-padded to n_time // 2 + 1 bins
    nb = n_time // 2 + 1
    spec = np.zeros((nb, prod.shape[1]), dtype=np.complex128)
    spec[: min(nb, prod.shape[0])] = prod[:nb]
    delta = np.zeros((n_time, prod.shape[1]))
    delta[0, :] = 1.0
    new_time_data = backend.spectral_division(delta, n_time, spec, n_time)
    new_sig = ImpulseResponse(None, new_time_data, fs_hz, constrain_amplitude=False)
    if padding and keep_original_length:
        new_sig.time_data = new_sig.time_data[:original_length].copy()
    return new_sig


def spectral_deconvolve(output: Signal, input: Signal, apply_regularization: bool = True,
                        start_stop_hz=None, threshold_db: float = -30.0, padding: bool = False,
                        keep_original_length: bool = False) -> ImpulseResponse:
    """Impulse response by (regularised) spectral division output / input."""
    assert len(output) == len(input), "Lengths do not match for spectral deconvolution"
    multichannel = input.number_of_channels == 1
    if not multichannel:
        assert output.number_of_channels == input.number_of_channels, \
            "The number of channels do not match."
    assert output.sampling_rate_hz == input.sampling_rate_hz, "Sampling rates do not match"
    if not apply_regularization:
        assert start_stop_hz is None, \
            "No start_stop_hz vector can be passed when using standard mode"
    for s in (output, input):
        if s._spectrum_parameters["smoothing"] != 0:
            raise NotImplementedError("fractional-octave spectrum smoothing is outside the GPU hot path")
    # Only the spectrum METHOD is forced (transfer_functions.py:142-143): each signal's own scaling still
    # applies inside get_spectrum (classes/signal.py:899-938).  The defaults (FFTBackward) give the plain
    # transform and take the fused device path below; any other scaling goes through the general one.
    plain = all(s._spectrum_parameters["scaling"].fft_norm() == "backward"
                and not s._spectrum_parameters["scaling"].has_physical_units() for s in (output, input))
    if not plain:
        return _spectral_deconvolve_scaled(output, input, apply_regularization, start_stop_hz, threshold_db,
                                           padding, keep_original_length, multichannel)
    fs_hz = output.sampling_rate_hz
    original_length = len(output)
    n_time = original_length * 2 if padding else original_length
    n_fft = (next_fast_len(n_time, True)
             if input._spectrum_parameters["pad_to_fast_length"] else n_time)
    if output.on_device and n_fft == n_time and not output.is_complex_signal and not input.is_complex_signal:
        # device-resident samples: spectrum of the input, regularised inverse, division and inverse transform on the
        # device; only the input's spectrum comes down for the band detection (and a (bins,) eps goes up)
        freqs_dev = np.fft.rfftfreq(n_fft, 1 / fs_hz)

        def eps_of(den, band=start_stop_hz):
            if band is None:  # band from the FIRST denominator channel only
                band = find_frequencies_above_threshold(den[:, 0], freqs_dev, threshold_db)
            if len(band) == 2:
                band = np.array([band[0] / np.sqrt(2), band[0], band[1], np.min([band[1] * np.sqrt(2), fs_hz / 2])])
            elif len(band) != 4:
                raise ValueError("start_stop_hz vector should have 2 or 4 values")
            return _inverse_hann_band(find_nearest_points_index_in_vector(band, freqs_dev), len(freqs_dev)) * 10 ** (30 / 20)

        n_out = original_length if (padding and keep_original_length) else n_time
        dev = backend.spectral_division_device(output.device_samples, input.to_device().device_samples, n_fft, n_out,
                                               eps_of if apply_regularization else None)
        return ImpulseResponse.from_planar_f32(dev, fs_hz)
    # rfft(n=n_fft) zero pads, so the optional x2 padding needs no copy
    denum_fft = backend.rfft_spectrum(input.time_data, n_fft)
    freqs_hz = np.fft.rfftfreq(n_fft, 1 / fs_hz)
    eps = None
    if apply_regularization:
        if start_stop_hz is None:  # band from the FIRST denominator channel only
            start_stop_hz = find_frequencies_above_threshold(denum_fft[:, 0], freqs_hz,
                                                             threshold_db)
        if len(start_stop_hz) == 2:
            start_stop_hz = np.array([start_stop_hz[0] / np.sqrt(2), start_stop_hz[0],
                                      start_stop_hz[1],
                                      np.min([start_stop_hz[1] * np.sqrt(2), fs_hz / 2])])
        elif len(start_stop_hz) != 4:
            raise ValueError("start_stop_hz vector should have 2 or 4 values")
        ids = find_nearest_points_index_in_vector(start_stop_hz, freqs_hz)
        eps = _inverse_hann_band(ids, len(freqs_hz)) * 10 ** (30 / 20)
    inverse = backend.regularized_inverse(denum_fft, eps)  # (B, Cx)
    if n_fft == n_time:
        new_time_data = backend.spectral_division(
            output.time_data, n_fft, inverse[:, 0] if multichannel else inverse, n_time)
    else:
        # The signal length is not a fast length: the reference transforms with n_fft =
        # next_fast_len(n_time) points but inverts with np.fft.irfft(..., n=n_time)
        # (_transfer_functions.py:37-41), and numpy then CROPS the spectrum to n_time//2 + 1 bins
        # and runs an n_time-point inverse (the bins keep their values but not their spacing).
        # Reproduced literally: forward spectrum on the device, the cropped product as a
        # per-channel "inverse spectrum" against a unit impulse (whose rfft is 1): the device
        # computes irfft_{n_time}(1 * product).
        num_fft = backend.rfft_spectrum(output.time_data, n_fft)               # (n_fft//2 + 1, C)
        prod = (num_fft * (inverse[:, :1] if multichannel else inverse))[: n_time // 2 + 1]
        delta = np.zeros((n_time, prod.shape[1]))
        delta[0, :] = 1.0
        new_time_data = backend.spectral_division(delta, n_time, prod, n_time)
    new_sig = ImpulseResponse(None, new_time_data, fs_hz, constrain_amplitude=False)
    if padding and keep_original_length:
        new_sig.time_data = new_sig.time_data[:original_length].copy()
    return new_sig
