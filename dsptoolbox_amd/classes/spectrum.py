"""Spectrum container returned by compute_transfer_function
(API mirror of dsptoolbox/classes/spectrum.py: ctor :31-53, setters :138-269,
set_coherence :871-885).  Interpolation, smoothing and plotting of the
reference class are outside the hot path."""

from copy import deepcopy

import numpy as np

from ..standard.enums import SpectrumType
from ._multichannel_data import MultichannelData


class Spectrum(MultichannelData):
    def __init__(self, frequency_vector_hz, spectral_data):
        self.frequency_vector_hz = frequency_vector_hz
        self.spectral_data = spectral_data

    @classmethod
    def _from_device_result(cls, frequency_vector_hz, spectral_data, coherence=None) -> "Spectrum":
        """The same object as Spectrum(f, data) + set_coherence for arrays the library has just produced: float64 /
        complex128, (bins, channels), owned by nobody else -- no second copy, no re-validation (the checks and the
        astype of the setters are half of a device-resident call's 0.2 ms)."""
        obj = cls.__new__(cls)
        obj._Spectrum__frequency_vector_hz = frequency_vector_hz
        obj._Spectrum__spectral_data = None
        obj._narrow = (spectral_data, coherence)  # complex64 / float32 as downloaded: widened on first access
        return obj

    def _widen_result(self):
        """First look at a device-produced spectrum: complex64 -> complex128, float32 -> float64 (what the reference's
        setters hold), once."""
        narrow = self.__dict__.pop("_narrow", None)
        if narrow is not None:
            from .. import backend
            self.__spectral_data = backend._widen(narrow[0])
            if narrow[1] is not None:
                self.__dict__["coherence"] = backend._widen(narrow[1])

    def __getattr__(self, name):
        # `coherence` is a plain attribute in the reference (set by set_coherence, looked for with hasattr)
        if name == "coherence" and "_narrow" in self.__dict__ and self.__dict__["_narrow"][1] is not None:
            self._widen_result()
            return self.__dict__["coherence"]
        raise AttributeError(name)

    @staticmethod
    def from_signal(sig, complex: bool = False) -> "Spectrum":
        if complex:
            assert sig.spectrum_scaling.outputs_complex_spectrum(sig.spectrum_method), \
                "Method or scaling do not deliver a complex spectrum"
        f, sp = sig.get_spectrum()
        if complex:
            assert np.iscomplexobj(sp), "Spectrum of signal is not complex"
            return Spectrum(f, sp)
        return Spectrum(f, np.abs(sp) if sig.spectrum_scaling.is_amplitude_scaling()
                        else np.abs(sp) ** 0.5)

    @property
    def frequency_vector_hz(self):
        return self.__frequency_vector_hz

    @frequency_vector_hz.setter
    def frequency_vector_hz(self, new_freqs):
        assert not np.iscomplexobj(new_freqs), "Complex frequencies are invalid"
        f = np.atleast_1d(new_freqs).astype(np.float64)
        assert f.ndim == 1, "Frequency vector can only have a single dimension"
        assert np.all(f >= 0.0), "Negative frequencies are not supported"
        assert np.all(np.ediff1d(f) > 0.0), "Frequency vector is not strictly ascending"
        self.__frequency_vector_hz = f

    @property
    def number_frequency_bins(self) -> int:
        return len(self.frequency_vector_hz)

    @property
    def length_frequency_bins(self) -> int:
        return self.number_frequency_bins

    @property
    def spectral_data(self):
        if self.__spectral_data is None:
            self._widen_result()
        return self.__spectral_data

    @spectral_data.setter
    def spectral_data(self, new_data):
        data = np.atleast_2d(new_data)
        assert data.ndim == 2, "Spectral data must have two dimensions"
        if data.shape[0] < data.shape[1]:
            data = data.T
        assert data.shape[0] == self.number_frequency_bins, \
            "Spectral data and frequency vector lengths do not match"
        is_magnitude = np.isrealobj(data)
        self.__spectral_data = data.astype(np.float64 if is_magnitude else np.complex128)
        if self.is_magnitude:
            assert np.all(self.__spectral_data >= 0.0), \
                "No negative values are allowed for the magnitude spectrum"

    @property
    def is_magnitude(self) -> bool:
        return np.isrealobj(self.spectral_data)

    @property
    def is_complex(self) -> bool:
        return not self.is_magnitude

    @property
    def spectrum_type(self) -> SpectrumType:
        return SpectrumType.Magnitude if self.is_magnitude else SpectrumType.Complex

    @property
    def has_coherence(self) -> bool:
        return hasattr(self, "coherence")

    def set_coherence(self, coherence):
        assert coherence.shape == self.spectral_data.shape, \
            "Length of signals and given coherence do not match"
        assert not np.iscomplexobj(coherence), "Coherence cannot be complex"
        self.coherence = coherence

    def sum_channels(self, power_sum: bool = True) -> "Spectrum":
        """One-channel spectrum of all channels: the root of the summed powers (default), or the plain sum of the
        magnitude / complex data (dsptoolbox/classes/spectrum.py:435-459)."""
        if power_sum:
            return self._create_copy_with_new_data(
                np.sum(np.abs(self.spectral_data) ** 2.0, axis=1, keepdims=True) ** 0.5)
        return super().sum_channels()

    def copy(self) -> "Spectrum":
        return deepcopy(self)

    def _get_data(self):
        return self.spectral_data

    def _set_data(self, data) -> None:
        self.spectral_data = data

    def _create_copy_with_new_data(self, data):
        new = Spectrum(self.frequency_vector_hz.copy(), data)
        return new

    def _update_state(self) -> None:
        pass
