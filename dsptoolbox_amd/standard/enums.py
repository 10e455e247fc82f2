"""Enums of the spectral hot path (API mirror of dsptoolbox/standard/enums.py).

Only the members and methods the hot path touches are provided:
SpectrumMethod (:7-18), SpectrumScaling (:21-229), FilterCoefficientsType
(:232-243), FilterBankMode (:279-292), FilterPassType (:295-305), Window
(:341-437), SpectrumType.
"""

from enum import Enum, auto

import numpy as np
from scipy.signal.windows import get_window as _get_window


class SpectrumMethod(Enum):
    """How `Signal.get_spectrum` obtains a spectrum: Welch averaging or one FFT
    over the whole signal."""

    WelchPeriodogram = auto()
    FFT = auto()


class SpectrumScaling(Enum):
    """Spectrum scalings.  Amplitude-type: AmplitudeSpectrum,
    AmplitudeSpectralDensity, FFTBackward, FFTForward, FFTOrthogonal.
    Power-type: PowerSpectrum, PowerSpectralDensity."""

    AmplitudeSpectrum = auto()
    AmplitudeSpectralDensity = auto()
    PowerSpectrum = auto()
    PowerSpectralDensity = auto()
    FFTBackward = auto()
    FFTForward = auto()
    FFTOrthogonal = auto()

    def fft_norm(self) -> str:
        if self == SpectrumScaling.FFTForward:
            return "forward"
        if self == SpectrumScaling.FFTOrthogonal:
            return "ortho"
        return "backward"

    def is_amplitude_scaling(self) -> bool:
        return self not in (SpectrumScaling.PowerSpectrum,
                            SpectrumScaling.PowerSpectralDensity)

    def outputs_complex_spectrum(self, method: SpectrumMethod) -> bool:
        if method == SpectrumMethod.WelchPeriodogram:
            return False
        return self.is_amplitude_scaling()

    def has_physical_units(self) -> bool:
        return self in (SpectrumScaling.AmplitudeSpectrum,
                        SpectrumScaling.AmplitudeSpectralDensity,
                        SpectrumScaling.PowerSpectrum,
                        SpectrumScaling.PowerSpectralDensity)

    def is_spectral_density(self) -> bool:
        return self in (SpectrumScaling.AmplitudeSpectralDensity,
                        SpectrumScaling.PowerSpectralDensity)

    def get_scaling_factor(self, length_time_data_samples: int, sampling_rate_hz: int,
                           window):
        """Factor for the one-sided forward transform (DC/Nyquist corrected by
        the caller); amplitude form for amplitude scalings, squared otherwise."""
        n, fs = length_time_data_samples, sampling_rate_hz
        if self == SpectrumScaling.FFTBackward:
            return np.atleast_1d(1.0)
        if self == SpectrumScaling.FFTForward:
            return np.atleast_1d(1.0 / n)
        if self == SpectrumScaling.FFTOrthogonal:
            return np.atleast_1d((1.0 / n) ** 0.5)
        if self.is_spectral_density():
            energy = n if window is None else np.sum(window**2, axis=0, keepdims=True)
            factor = (2 / energy / fs) ** 0.5
        else:
            gain = n if window is None else np.sum(window, axis=0, keepdims=True)
            factor = 2**0.5 / gain
        factor = np.atleast_1d(factor)
        return factor if self.is_amplitude_scaling() else factor**2.0

    def conversion_factor(self, output: "SpectrumScaling", length_time_data_samples: int,
                          sampling_rate_hz: int, window):
        fin = np.asarray(self.get_scaling_factor(length_time_data_samples,
                                                 sampling_rate_hz, window), dtype=np.float64)
        fout = np.asarray(output.get_scaling_factor(length_time_data_samples,
                                                    sampling_rate_hz, window), dtype=np.float64)
        if self.is_amplitude_scaling() == output.is_amplitude_scaling():
            return fout / fin
        if self.is_amplitude_scaling():
            return fout / fin**2.0
        return fout**2.0 / fin


class SpectrumType(Enum):
    Power = auto()
    Magnitude = auto()
    Complex = auto()
    Db = auto()


class FilterCoefficientsType(Enum):
    Zpk = auto()
    Sos = auto()
    Ba = auto()


class FilterBankMode(Enum):
    """Parallel -> MultiBandSignal of band outputs; Sequential -> cascade;
    Summed -> sum of the band outputs."""

    Parallel = auto()
    Sequential = auto()
    Summed = auto()


class FilterPassType(Enum):
    Lowpass = auto()
    Highpass = auto()
    Bandpass = auto()
    Bandstop = auto()

    def __str__(self):
        return self.name.lower()

    def to_str(self):
        return str(self)


class Window(Enum):
    """Window types, produced by `scipy.signal.windows.get_window`."""

    Boxcar = auto()
    Triang = auto()
    Blackman = auto()
    Hamming = auto()
    Hann = auto()
    Bartlett = auto()
    Flattop = auto()
    Parzen = auto()
    Bohman = auto()
    Blackmanharris = auto()
    Nuttall = auto()
    Barthann = auto()
    Cosine = auto()
    Exponential = auto()
    Tukey = auto()
    Taylor = auto()
    Lanczos = auto()
    Kaiser = auto()
    KaiserBesselDerived = auto()
    Gaussian = auto()
    GeneralCosine = auto()
    GeneralGaussian = auto()
    GeneralHamming = auto()
    Dpss = auto()
    Chebwin = auto()

    @property
    def extra_parameter(self):
        return self.__extra_parameter

    def with_extra_parameter(self, extra_parameter):
        self.__extra_parameter = extra_parameter
        return self

    def needs_extra_parameter(self) -> bool:
        return self.name in ("Kaiser", "KaiserBesselDerived", "Gaussian", "GeneralCosine",
                             "GeneralGaussian", "GeneralHamming", "Dpss", "Chebwin")

    def _scipy_name(self) -> str:
        special = {"KaiserBesselDerived": "kaiser_bessel_derived",
                   "GeneralCosine": "general_cosine",
                   "GeneralGaussian": "general_gaussian",
                   "GeneralHamming": "general_hamming"}
        return special.get(self.name, self.name.lower())

    def to_scipy_format(self):
        if self.needs_extra_parameter():
            if self == Window.GeneralGaussian:
                return (self._scipy_name(), self.extra_parameter[0], self.extra_parameter[1])
            return (self._scipy_name(), self.extra_parameter)
        return self._scipy_name()

    def __call__(self, n_values: int, symmetric: bool):
        return _get_window(self.to_scipy_format(), n_values, not symmetric)
