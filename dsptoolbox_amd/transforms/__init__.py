"""transforms: the inverse STFT of dsptoolbox/transforms/transforms.py:444-586 on the device
(SURVEY.md section 8(f), row 1).  Same signature, parameter handling and quirks as the reference;
the frame-wise inverse FFTs and the windowed overlap-add with the squared-window envelope
(standard/_framed_signal_representation.py:70-137) run in the HIP library (ds_istft)."""

from __future__ import annotations

import numpy as np
from scipy.signal import get_window

from .. import backend
from ..classes.signal import Signal

__all__ = ["istft"]


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
    stft = np.asarray(stft)
    assert stft.ndim == 3, f"{stft.ndim} is not a valid number of dimensions. It must be 3"
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
    td = backend._istft(stft, nfft_eff, W, step, window, scale, frame_offset=0 if pad else 1,
                        n_frames_total=n_frames if pad else n_frames + 2)
    if pad:
        overlap = int(parameters["overlap_percent"] / 100 * len(window))
        td = td[overlap:-overlap, :]
    else:
        td = td[step:-step, :]
    if original_signal is not None:
        td = _pad_trim(td, original_signal.time_data.shape[0])
        return original_signal.copy_with_new_time_data(td)
    return Signal(None, time_data=td, sampling_rate_hz=sampling_rate_hz)
