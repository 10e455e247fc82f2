"""transforms: the inverse STFT (dsptoolbox/transforms/transforms.py:444-586, SURVEY.md section 8(f)
row 1) and the STFT consumers log_mel_spectrogram / mfcc / chroma_stft (:113-203, :335-441, :589-684,
row 4) on the device.  Same signature, parameter handling and quirks as the reference;
the frame-wise inverse FFTs and the windowed overlap-add with the squared-window envelope
(standard/_framed_signal_representation.py:70-137) run in the HIP library (ds_istft)."""

from __future__ import annotations

import numpy as np
from scipy.signal import get_window

from .. import backend
from ..classes.signal import Signal

__all__ = ["istft", "mel_filterbank", "log_mel_spectrogram", "mfcc", "chroma_stft"]


def _pad_trim(td: np.ndarray, desired_length: int) -> np.ndarray:
    """helpers/other.py:216-259 for (N, C) data, at the end."""
    n = td.shape[0]
    if n >= desired_length:
        return td[:desired_length].copy()
    return np.concatenate([td, np.zeros((desired_length - n, td.shape[1]), dtype=td.dtype)])


def istft(stft, original_signal: Signal | None = None, parameters: dict | None = None,
          sampling_rate_hz: int | None = None, window_length_samples: int | None = None,
          window_type=None, overlap_percent: int | None = None,
          fft_length_samples: int | None = None, padding: bool | None = None,
          scaling=None) -> Signal:
    """Complex STFT (frequency, time frame, channel) -> time signal (Griffin & Lim)."""
    resident = isinstance(stft, backend.DeviceSTFT)  # Signal.get_spectrogram(on_device=True): stays in HBM
    if not resident:
        stft = np.asarray(stft)
    assert len(stft.shape) == 3, f"{len(stft.shape)} is not a valid number of dimensions. It must be 3"
    if original_signal is not None:
        assert parameters is None, "A signal was passed. No parameters dictionary should be passed"
        parameters = original_signal._spectrogram_parameters.copy()
    elif parameters is not None:
        pass
    else:
        assert ((window_length_samples is not None) and (window_type is not None)
                and (overlap_percent is not None) and (padding is not None)
                and (scaling is not None)), \
            "At least one of the needed parameters needed was passed as None"
        parameters = {"window_length_samples": window_length_samples, "window_type": window_type,
                      "overlap_percent": overlap_percent, "fft_length_samples": fft_length_samples,
                      "padding": padding, "scaling": scaling}
    W = parameters["window_length_samples"]
    window = get_window(parameters["window_type"].to_scipy_format(), W)
    sc = parameters["scaling"]
    # irfft(..., n=None) takes n = 2 (bins - 1)
    nfft = parameters["fft_length_samples"]
    nfft_eff = 2 * (stft.shape[0] - 1) if nfft is None else int(nfft)
    norm = sc.fft_norm()
    scale = {"backward": 1.0 / nfft_eff, "forward": 1.0, "ortho": nfft_eff ** -0.5}[norm]
    if sc.has_physical_units():
        scale = scale / float(np.asarray(sc.get_scaling_factor(nfft, sampling_rate_hz, window)).ravel()[0])
    step = int((1 - parameters["overlap_percent"] / 100) * len(window))
    n_frames = stft.shape[1]
    pad = bool(parameters["padding"])
    if resident:
        from .._lib import DevicePlanar
        dev = backend._istft_device(stft, nfft_eff, W, step, window, scale, frame_offset=0 if pad else 1,
                                    n_frames_total=n_frames if pad else n_frames + 2)
        cut = int(parameters["overlap_percent"] / 100 * len(window)) if pad else step
        length = dev.n_samples - 2 * cut
        want = len(original_signal) if original_signal is not None else length
        if cut > 0 and 0 < want <= length:
            # trimming both ends (and to the original length) is a view of the same device samples
            view = DevicePlanar(dev.owner, dev.n_ch, want, dev.ld, 4 * cut)
            if original_signal is not None:
                return original_signal._device_result(view)
            return Signal.from_planar_f32(view, sampling_rate_hz)
        td = backend._interleaved_f64(dev.to_planar())  # (padding would need zeros behind the samples: host)
    else:
        td = backend._istft(stft, nfft_eff, W, step, window, scale, frame_offset=0 if pad else 1,
                            n_frames_total=n_frames if pad else n_frames + 2)
    if pad:
        overlap = int(parameters["overlap_percent"] / 100 * len(window))
        td = td[overlap:-overlap, :]
    else:
        td = td[step:-step, :]
    if original_signal is not None:
        td = _pad_trim(td, len(original_signal))
        return original_signal.copy_with_new_time_data(td)
    return Signal(None, time_data=td, sampling_rate_hz=sampling_rate_hz)


def _hz2mel(f):
    """helpers/frequency_conversion.py:7-25"""
    return 2595 * np.log10(1 + f / 700)


def _mel2hz(mel):
    """helpers/frequency_conversion.py:28-46"""
    return 700 * (10 ** (mel / 2595) - 1)


def mel_filterbank(f_hz, range_hz=None, n_bands: int = 40, normalize: bool = True):
    """Equidistant mel triangle filters (bands, frequency) and their centre frequencies in mel
    (transforms/transforms.py:206-277).  Host-side parameter preparation."""
    f_hz = np.squeeze(f_hz)
    assert f_hz.ndim == 1, "f_hz should be a 1D-array"
    n_bands = int(n_bands)
    if range_hz is None:
        range_hz = f_hz[[0, -1]]
    else:
        range_hz = np.atleast_1d(np.asarray(range_hz).squeeze())
        assert len(range_hz) == 2, "range_hz should be an array with exactly two values!"
        range_hz = np.sort(range_hz)
        assert range_hz[-1] <= f_hz[-1], (
            f"Upper frequency in range {range_hz[-1]} is bigger than nyquist frequency {f_hz[-1]}")
        assert range_hz[0] >= 0, "Lower frequency in range must be positive"
    range_mel = _hz2mel(range_hz)
    mel_center_freqs = np.linspace(range_mel[0], range_mel[1], n_bands + 2, endpoint=True)
    bands_hz = _mel2hz(mel_center_freqs)
    inds = np.array([np.argmin(np.abs(b - f_hz)) for b in bands_hz], dtype=int)
    mel_filters = np.zeros((n_bands, len(f_hz)))
    for n in range(n_bands):
        ni = n + 1
        mel_filters[n, inds[ni - 1]:inds[ni]] = np.linspace(0, 1, inds[ni] - inds[ni - 1], endpoint=False)
        mel_filters[n, inds[ni]:inds[ni + 1]] = np.linspace(1, 0, inds[ni + 1] - inds[ni], endpoint=False)
        if normalize:
            mel_filters[n, :] /= np.sum(mel_filters[n, :])
    return mel_filters, mel_center_freqs[1:-1]


def _band_power(signal: Signal, filters, f_hz_check, to_db: bool, dct_abs: bool):
    par = signal._spectrogram_parameters
    return backend._spectrogram_band_power(
        signal.device_samples if signal.on_device else signal.time_data, signal.sampling_rate_hz,
        par["window_length_samples"], par["window_type"],
        par["overlap_percent"], par["fft_length_samples"], par["detrend"], par["padding"], par["scaling"],
        filters, to_db, dct_abs)


def _spectrogram_axes(signal: Signal):
    par = signal._spectrogram_parameters
    pl = backend._stft_plan(backend._ShapeOnly(len(signal), signal.number_of_channels), signal.sampling_rate_hz,
                            par["window_length_samples"],
                            par["window_type"], par["overlap_percent"], par["fft_length_samples"],
                            par["padding"], par["scaling"], planar=False)  # axes only: no cast of the data
    return pl["time_s"], pl["freqs_hz"], pl["B"]


def log_mel_spectrogram(s: Signal, channel: int = 0, range_hz=None, n_bands: int = 40,
                        generate_plot: bool = True, stft_parameters: dict | None = None):
    """-> (time_s, f_mel, log_mel_sp (bands, time frame, channel)); STFT, |.|^2, the mel filter
    contraction and the dB conversion run on the device."""
    if generate_plot:
        raise NotImplementedError("plotting is outside the GPU hot path: pass generate_plot=False")
    if stft_parameters is not None:
        s.set_spectrogram_parameters(**stft_parameters)
    time_s, f_hz, n_bins = _spectrogram_axes(s)
    mfilt, f_mel = mel_filterbank(f_hz, range_hz, n_bands, normalize=True)
    assert mfilt.shape[1] == n_bins, \
        "the frequency vector (window length) and the STFT (fft length) have different bin counts"
    _, _, log_mel_sp = _band_power(s, mfilt, f_hz, True, False)
    return time_s, f_mel, log_mel_sp


def mfcc(signal: Signal, channel: int = 0, mel_filters=None, generate_plot: bool = True,
         stft_parameters: dict | None = None):
    """-> (time_s, f_mel, mfcc (cepstral coefficients, time frame, channel))."""
    if generate_plot:
        raise NotImplementedError("plotting is outside the GPU hot path: pass generate_plot=False")
    if stft_parameters is not None:
        signal.set_spectrogram_parameters(**stft_parameters)
    time_s, f, n_bins = _spectrogram_axes(signal)
    if mel_filters is None:
        mel_filters, f_mel = mel_filterbank(f, None, n_bands=40)
    else:
        mel_filters = np.asarray(mel_filters)
        f_mel = np.array([0, mel_filters.shape[0]])
    assert mel_filters.shape[1] == n_bins, (
        f"Shape of the mel filter matrix {mel_filters.shape} does not match the STFT")
    _, _, out = _band_power(signal, mel_filters, f, True, True)
    return time_s, f_mel, out


def _pitch2frequency(tuning_a_hz: float = 440):
    """transforms/_transforms.py:10-26: frequencies of the MIDI pitches 0..127 (0 is C0)."""
    return tuning_a_hz * 2 ** ((np.arange(128) - 69) / 12)


def chroma_stft(signal: Signal, tuning_a_hz: float = 440, compression: float = 0.5, plot_channel: int = -1):
    """Chroma features and pitch log-STFT (transforms/transforms.py:589-684).
    -> (time_s, chroma_stft (12 notes C..B, time frame, channel), pitch_stft (128, time frame,
    channel)).  The STFT, |.|^2 and the contraction onto the quarter-tone pitch bands run on the
    device (the spectrogram never leaves it); the octave sums over the 128 pitch rows and the
    logarithmic compression are done on the (128, F, C) result."""
    assert tuning_a_hz > 0, "Tuning A4 must be greater than zero"
    assert compression > 0, "Compression factor must be greater than zero"
    if plot_channel != -1:
        raise NotImplementedError("plotting is outside the GPU hot path: pass plot_channel=-1")
    time_s, f, n_bins = _spectrogram_axes(signal)
    pitch_frequencies = _pitch2frequency(tuning_a_hz)
    pitch_transformation = np.zeros((len(pitch_frequencies), len(f)))
    for ind, fn in enumerate(pitch_frequencies):
        pitch_transformation[ind, (f >= fn * 2 ** (-1 / 24)) & (f < fn * 2 ** (1 / 24))] = 1
    assert pitch_transformation.shape[1] == n_bins, \
        "the frequency vector (window length) and the STFT (fft length) have different bin counts"
    _, _, pitch_stft = _band_power(signal, pitch_transformation, f, False, False)
    n_notes = 12
    chroma_transformation = np.zeros((n_notes, len(pitch_frequencies)))
    for i in range(n_notes):
        chroma_transformation[i, i::n_notes] = 1
    chroma = np.tensordot(chroma_transformation, pitch_stft, (1, 0))
    return time_s, np.log(1 + compression * chroma), np.log(1 + compression * pitch_stft)
