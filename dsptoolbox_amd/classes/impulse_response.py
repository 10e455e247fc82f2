"""ImpulseResponse (API mirror of dsptoolbox/classes/impulse_response.py:21-66):
a Signal that defaults to constrain_amplitude=True and SpectrumMethod.FFT."""

from ..standard.enums import SpectrumMethod
from .signal import Signal


class ImpulseResponse(Signal):
    def __init__(self, path=None, time_data=None, sampling_rate_hz=None,
                 constrain_amplitude: bool = True, activate_cache: bool = False):
        super().__init__(path, time_data, sampling_rate_hz,
                         constrain_amplitude=constrain_amplitude, activate_cache=activate_cache)
        self.spectrum_method = SpectrumMethod.FFT

    @staticmethod
    def from_signal(signal: Signal):
        ir = ImpulseResponse(None, signal.time_data.copy(), signal.sampling_rate_hz,
                             signal.constrain_amplitude, signal.activate_cache)
        return ir

    @staticmethod
    def from_time_data(time_data, sampling_rate_hz: int, constrain_amplitude: bool = True):
        return ImpulseResponse(None, time_data, sampling_rate_hz, constrain_amplitude)
