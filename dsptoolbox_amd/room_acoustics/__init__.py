"""room_acoustics: applying a room impulse response to a signal
(dsptoolbox/room_acoustics/room_acoustics.py:216-266) -- the full
linear convolution runs on the device through the FIR block-convolution path (ds_fir_ola; impulse
responses longer than 8193 taps take its four-step-FFT overlap-save branch)."""

from __future__ import annotations

import numpy as np

from .. import backend
from ..classes.signal import Signal

__all__ = ["convolve_rir_on_signal"]


def convolve_rir_on_signal(signal: Signal, rir: Signal, keep_peak_level: bool = True,
                           keep_length: bool = True) -> Signal:
    """Convolve every channel of `signal` with the single-channel `rir`."""
    assert rir.number_of_channels == 1, "RIR should not contain more than one channel."
    assert rir.sampling_rate_hz == signal.sampling_rate_hz, "The sampling rates do not match"
    x = signal.time_data
    h = rir.time_data[:, 0]
    # mode="full": the causal convolution of the signal followed by len(h) - 1 zeros
    xfull = np.concatenate([x, np.zeros((len(h) - 1, x.shape[1]))], axis=0)
    new_time_data = backend.fir_filter_bank(xfull, [h], backend.DS_FB_PARALLEL)[0]
    if keep_length:
        new_time_data = new_time_data[: len(signal), ...]
    if keep_peak_level:
        old_peak_levels = np.max(np.abs(signal.time_data), axis=0)
        new_peak_levels = np.max(np.abs(new_time_data), axis=0)
        new_time_data *= (old_peak_levels / new_peak_levels)[None, ...]
    return signal.copy_with_new_time_data(new_time_data)
