"""ImpulseResponse (API mirror of dsptoolbox/classes/impulse_response.py:21-66, copy_with_new_time_data :355-371):
a Signal that defaults to constrain_amplitude=True and SpectrumMethod.FFT."""

from copy import deepcopy

import numpy as np

from ..standard.enums import SpectrumMethod
from .signal import Signal


class ImpulseResponse(Signal):
    def __init__(self, path=None, time_data=None, sampling_rate_hz=None,
                 constrain_amplitude: bool = True, activate_cache: bool = False):
        super().__init__(path, time_data, sampling_rate_hz,
                         constrain_amplitude=constrain_amplitude, activate_cache=activate_cache)
        self.spectrum_method = SpectrumMethod.FFT

    def _after_init(self) -> None:
        if self.spectrum_method != SpectrumMethod.FFT:
            self.spectrum_method = SpectrumMethod.FFT

    @staticmethod
    def from_signal(signal: Signal):
        ir = ImpulseResponse(None, signal.time_data.copy(), signal.sampling_rate_hz,
                             signal.constrain_amplitude, signal.activate_cache)
        return ir

    @staticmethod
    def from_time_data(time_data, sampling_rate_hz: int, constrain_amplitude: bool = True):
        return ImpulseResponse(None, time_data, sampling_rate_hz, constrain_amplitude)

    def set_window(self, window):
        """Keep the window that was applied to the impulse response, one column per channel
        (dsptoolbox/classes/impulse_response.py:139-152; the reference only plots it)."""
        assert window.shape == self.time_data.shape, f"{window.shape} does not match shape {self.time_data.shape}"
        self.window = window
        return self

    def copy_with_new_time_data(self, new_time_data) -> "ImpulseResponse":
        """An impulse response stays one when it is filtered (classes/impulse_response.py:355-371)."""
        if isinstance(new_time_data, np.ndarray) and new_time_data.base is not None:
            new_time_data = new_time_data.copy()  # the new object owns its samples
        new_ir = ImpulseResponse.from_time_data(new_time_data, self.sampling_rate_hz, self.constrain_amplitude)
        new_ir.calibrated_signal = self.calibrated_signal
        new_ir.activate_cache = self.activate_cache
        new_ir._spectrum_parameters = deepcopy(self._spectrum_parameters)
        new_ir._spectrogram_parameters = deepcopy(self._spectrogram_parameters)
        if new_ir.spectrum_method != SpectrumMethod.FFT:
            new_ir.spectrum_method = SpectrumMethod.FFT
        return new_ir
