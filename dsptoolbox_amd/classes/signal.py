"""Signal container with the hot-path getters
(API mirror of dsptoolbox/classes/signal.py: ctor :57-115, time_data setter
:222-301, set_spectrum_parameters :497-588, set_spectrogram_parameters
:706-773, get_spectrum :861-946, get_csm :948-1007, get_spectrogram
:1009-1056, copy_with_new_time_data :1647-1680).

Spectra, cross-spectral matrices and spectrograms are computed by the HIP
library (dsptoolbox_amd.backend).  File IO, plotting, smoothing and time
windows of the reference class are outside the hot path and not provided.
"""

from copy import deepcopy
from warnings import warn

import numpy as np
from scipy.fft import next_fast_len

from .. import backend
from ..standard.enums import SpectrumMethod, SpectrumScaling, Window
from ._multichannel_data import MultichannelData


class Signal(MultichannelData):
    def __init__(self, path=None, time_data=None, sampling_rate_hz=None,
                 constrain_amplitude: bool = False, activate_cache: bool = False):
        self.constrain_amplitude = constrain_amplitude
        self.calibrated_signal = False
        self.activate_cache = activate_cache
        self.__update_state()
        if path is not None:
            raise NotImplementedError(
                "audio file reading is outside the GPU hot path; pass time_data and sampling_rate_hz")
        assert time_data is not None, \
            "Either a path to an audio file or a time vector has to be passed"
        assert sampling_rate_hz is not None, "A sampling rate should be passed!"
        self.sampling_rate_hz = sampling_rate_hz
        self.time_data = time_data
        self.set_spectrum_parameters()
        self.set_spectrogram_parameters()

    @staticmethod
    def from_time_data(time_data, sampling_rate_hz: int, constrain_amplitude: bool = True):
        return Signal(None, time_data, sampling_rate_hz, constrain_amplitude)

    # ---- device residency (additive API: SURVEY section 7, hard parts 4 and 6) ----------------------------
    # The reference keeps (samples, channels) float64 on the host; the kernels read planar float32 in HBM.  Through
    # the plain API every hot-path call therefore casts, transposes and uploads its input and downloads and widens
    # its output (13.5 ms around 0.12 ms of kernels for the headline estimate).  A Signal may instead HOLD its samples
    # on the device: `from_planar_f32` / `to_device()` put them there once, compute_transfer_function,
    # get_spectrogram, get_spectrum (Welch), get_csm, Filter / FilterBank.filter_signal, spectral_deconvolve and istft
    # read them in place, and results that are signals (filter outputs, impulse responses, reconstructions) come back
    # device-resident too.  `time_data` stays what it is in the reference -- (samples, channels) float64 -- and is
    # materialised (one download + widen) the first time somebody asks for it; assigning to it drops the device copy.
    @classmethod
    def from_planar_f32(cls, planar, sampling_rate_hz: int):
        """A signal whose samples are (channels, samples) float32 -- a host array (uploaded once, no cast, no
        transpose) or a `DevicePlanar` that is already in HBM.  constrain_amplitude is False (the peak of samples
        that never visit the host is not inspected)."""
        from .._lib import DevicePlanar, get_context
        dev = planar if isinstance(planar, DevicePlanar) else DevicePlanar.from_planar(get_context(), planar)
        obj = cls.__new__(cls)
        Signal._init_on_device(obj, dev, sampling_rate_hz)
        return obj

    def _init_on_device(self, dev, sampling_rate_hz: int):
        self.__constrain_amplitude = False
        self.calibrated_signal = False
        self.activate_cache = False
        self.__update_state()
        assert sampling_rate_hz is not None, "A sampling rate should be passed!"
        self.sampling_rate_hz = sampling_rate_hz
        self.__adopt(dev)
        self.set_spectrum_parameters()
        self.set_spectrogram_parameters()
        self._after_init()

    def _after_init(self) -> None:  # (ImpulseResponse: the FFT spectrum method)
        pass

    def __adopt(self, dev):
        self.__time_data = None
        self.__time_data_imaginary = None
        self.__device = dev
        self.__shape = (dev.n_samples, dev.n_ch)
        self.__amplitude_scale_factor = 1.0
        self.__update_state()
        self.clear_time_window()

    @property
    def on_device(self) -> bool:
        """Are the samples held in HBM (planar float32)?"""
        return getattr(self, "_Signal__device", None) is not None

    def to_device(self):
        """Put the samples in HBM (one cast + transpose + upload) and keep them there beside the host copy; the
        hot-path calls then read them in place.  Real signals only.  Returns self."""
        if not self.on_device:
            assert not self.is_complex_signal, "signals with imaginary time data stay on the host"
            from .._lib import DevicePlanar, get_context
            self.__device = DevicePlanar.from_planar(get_context(), backend._planar_f32(self.__time_data))
        return self

    @property
    def _has_host_copy(self) -> bool:
        return self.__time_data is not None

    @property
    def device_samples(self):
        """The `DevicePlanar` of a device-resident signal (None otherwise)."""
        return getattr(self, "_Signal__device", None)

    def __update_state(self):
        self.__spectrum_state_update = True
        self.__csm_state_update = True
        self.__spectrogram_state_update = True
        self.__time_vector_update = True

    # ---- properties --------------------------------------------------------
    @property
    def time_data(self) -> np.ndarray:
        if self.__time_data is None:  # device-born: one download + widen, kept from then on
            self.__time_data = backend._interleaved_f64(self.__device.to_planar())
        return self.__time_data

    @time_data.setter
    def time_data(self, new_time_data):
        self.__device = None  # new samples: the device copy (if any) no longer describes the signal
        new_time_data = np.atleast_2d(new_time_data).squeeze()
        assert new_time_data.ndim <= 2, (
            f"{new_time_data.ndim} are too many dimensions for time data. "
            "Dimensions should be [time samples, channels]")
        if new_time_data.ndim < 2:
            new_time_data = new_time_data[..., None]
        if new_time_data.shape[1] > new_time_data.shape[0]:  # more samples than channels, always
            new_time_data = new_time_data.T
        if np.iscomplexobj(new_time_data):
            new_imag = np.imag(new_time_data)
            new_time_data = np.real(new_time_data)
        else:
            new_imag = None
        self.__amplitude_scale_factor = 1.0
        if self.constrain_amplitude:
            peak = np.max(np.abs(new_time_data))
            if new_imag is not None:
                peak = max(peak, np.max(np.abs(new_imag)))
            if peak > 1.0:
                new_time_data = new_time_data / peak
                warn("Signal was over 0 dBFS, normalizing to 0 dBFS peak level was triggered")
                if new_imag is not None:
                    new_imag = new_imag / peak
                self.__amplitude_scale_factor = 1.0 / peak
        self.__time_data = new_time_data
        self.__shape = new_time_data.shape
        self.time_data_imaginary = new_imag
        self.__update_state()
        self.clear_time_window()

    @property
    def amplitude_scale_factor(self) -> float:
        return self.__amplitude_scale_factor

    @property
    def sampling_rate_hz(self) -> int:
        return self.__sampling_rate_hz

    @sampling_rate_hz.setter
    def sampling_rate_hz(self, new_sampling_rate_hz):
        assert type(new_sampling_rate_hz) is int, "Sampling rate can only be an integer"
        self.__sampling_rate_hz = new_sampling_rate_hz
        self.__update_state()

    @property
    def length_seconds(self) -> float:
        return len(self) / self.sampling_rate_hz

    @property
    def length_samples(self) -> int:
        return len(self)

    @property
    def time_vector_s(self) -> np.ndarray:
        if self.__time_vector_update:
            self.__time_vector_update = False
            self.__time_vector_s = np.linspace(0, len(self) / self.sampling_rate_hz, len(self))
        return self.__time_vector_s

    @property
    def time_data_imaginary(self):
        return self.__time_data_imaginary

    @time_data_imaginary.setter
    def time_data_imaginary(self, new_imag):
        if new_imag is not None:
            assert new_imag.shape == self.__shape, \
                "Shape of imaginary part time data does not match"
            self.__device = None  # (a complex signal lives on the host)
        self.__time_data_imaginary = new_imag

    @property
    def is_complex_signal(self) -> bool:
        return self.time_data_imaginary is not None

    @property
    def constrain_amplitude(self) -> bool:
        return self.__constrain_amplitude

    @constrain_amplitude.setter
    def constrain_amplitude(self, nca):
        assert type(nca) is bool, "constrain_amplitude must be of type boolean"
        self.__constrain_amplitude = nca
        if nca and hasattr(self, "time_data"):
            self.time_data = self.time_data

    @property
    def calibrated_signal(self) -> bool:
        return self.__calibrated_signal

    @calibrated_signal.setter
    def calibrated_signal(self, ncs):
        assert type(ncs) is bool, "calibrated_signal must be of type boolean"
        self.__calibrated_signal = ncs

    @property
    def metadata(self) -> dict:
        return dict(sampling_rate_hz=self.sampling_rate_hz,
                    number_of_channels=self.number_of_channels,
                    signal_length_samples=self.length_samples,
                    signal_length_seconds=self.length_seconds,
                    constrain_amplitude=self.constrain_amplitude,
                    amplitude_scale_factor=self.amplitude_scale_factor,
                    is_complex_signal=self.is_complex_signal)

    def __len__(self):
        return int(self.__shape[0])

    @property
    def number_of_channels(self) -> int:
        return int(self.__shape[1])

    def __iter__(self):
        return iter([self.time_data[:, x] for x in range(self.number_of_channels)])

    # ---- parameters ----------------------------------------------------------
    def set_spectrum_parameters(self, method: SpectrumMethod = SpectrumMethod.WelchPeriodogram,
                                smoothing: int = 0, pad_to_fast_length: bool = True,
                                window_length_samples: int = 1024,
                                window_type: Window = Window.Hann, overlap_percent: float = 50,
                                detrend: bool = True, average: str = "mean",
                                scaling: SpectrumScaling = SpectrumScaling.FFTBackward):
        new = dict(method=method, smoothing=smoothing, pad_to_fast_length=pad_to_fast_length,
                   window_length_samples=window_length_samples, window_type=window_type,
                   overlap_percent=overlap_percent, detrend=detrend, average=average,
                   scaling=scaling)
        if not hasattr(self, "_spectrum_parameters"):
            self._spectrum_parameters = new
            self.__spectrum_state_update = True
        elif not all(self._spectrum_parameters[k] == new[k] for k in self._spectrum_parameters):
            self._spectrum_parameters = new
            self.__spectrum_state_update = True
            self.__csm_state_update = True
        return self

    @property
    def spectrum_scaling(self) -> SpectrumScaling:
        return self._spectrum_parameters["scaling"]

    @spectrum_scaling.setter
    def spectrum_scaling(self, new_scaling: SpectrumScaling):
        assert isinstance(new_scaling, SpectrumScaling)
        self._spectrum_parameters["scaling"] = new_scaling
        self.__spectrum_state_update = True
        self.__csm_state_update = True

    @property
    def spectrum_method(self) -> SpectrumMethod:
        return self._spectrum_parameters["method"]

    @spectrum_method.setter
    def spectrum_method(self, new_method: SpectrumMethod):
        assert isinstance(new_method, SpectrumMethod)
        self._spectrum_parameters["method"] = new_method
        self.__spectrum_state_update = True
        self.__csm_state_update = True

    @property
    def spectrum_smoothing(self) -> float:
        return self._spectrum_parameters["smoothing"]

    @spectrum_smoothing.setter
    def spectrum_smoothing(self, new_smoothing):
        assert new_smoothing >= 0.0, "Smoothing must be positive or zero"
        self._spectrum_parameters["smoothing"] = float(new_smoothing)

    def set_spectrogram_parameters(self, window_length_samples: int = 1024,
                                   window_type: Window = Window.Hann,
                                   overlap_percent: float = 50.0, fft_length_samples=None,
                                   detrend: bool = False, padding: bool = True,
                                   scaling: SpectrumScaling = SpectrumScaling.FFTBackward):
        new = dict(window_length_samples=window_length_samples, window_type=window_type,
                   overlap_percent=overlap_percent, fft_length_samples=fft_length_samples,
                   detrend=detrend, padding=padding, scaling=scaling)
        if not hasattr(self, "_spectrogram_parameters"):
            self._spectrogram_parameters = new
            self.__spectrogram_state_update = True
        elif not all(self._spectrogram_parameters[k] == new[k]
                     for k in self._spectrogram_parameters):
            self._spectrogram_parameters = new
            self.__spectrogram_state_update = True
        return self

    def add_channel(self, path=None, new_time_data=None, sampling_rate_hz=None, allow_padding_trimming: bool = True):
        """Appends new channels (classes/signal.py:776-852); data of another length is zero-padded or trimmed at
        its end when `allow_padding_trimming`.  Reading from a file (path) is audio IO: out of scope here."""
        if path is not None:
            assert new_time_data is None, "Only path or new time data is accepted, not both."
            raise NotImplementedError("audio file IO is outside the GPU hot path: pass new_time_data")
        assert sampling_rate_hz == self.sampling_rate_hz, \
            f"{sampling_rate_hz} does not match {self.sampling_rate_hz} as the sampling rate"
        new_time_data = np.array(new_time_data)
        if new_time_data.ndim > 2:
            new_time_data = new_time_data.squeeze()
        assert new_time_data.ndim <= 2, (f"{new_time_data.ndim} are too many dimensions for time data. "
                                         "Dimensions should be (time samples, channels)")
        if new_time_data.ndim < 2:
            new_time_data = new_time_data[..., None]
        if new_time_data.shape[1] > new_time_data.shape[0]:
            new_time_data = new_time_data.T
        diff = new_time_data.shape[0] - self.time_data.shape[0]
        if diff != 0:
            if not allow_padding_trimming:
                raise AttributeError(f"{new_time_data.shape[0]} does not match {self.time_data.shape[0]}. "
                                     "Activate allow_padding_trimming for allowing this channel to be added")
            new_time_data = backend._pad_trim(new_time_data, self.time_data.shape[0])
            warn(("Padding" if diff < 0 else "Trimming") + " has been performed on the end of the new signal to "
                 "match original one.")
        self.time_data = np.concatenate([self.time_data, new_time_data], axis=1)
        self.__update_state()
        return self

    def clear_time_window(self):
        if hasattr(self, "window"):
            del self.window
        return self

    # ---- getters (device) ------------------------------------------------------
    def get_spectrum(self, force_computation=False):
        """-> (freqs_hz, spectrum) with the stored spectrum parameters."""
        if not (not hasattr(self, "spectrum") or self.__spectrum_state_update
                or force_computation):
            return self.spectrum[0].copy(), self.spectrum[1].copy()
        par = self._spectrum_parameters
        if self.spectrum_method == SpectrumMethod.WelchPeriodogram:
            if self.on_device and not self._short_estimate(par):
                spectrum = backend._welch_psd_device(self.device_samples, self.sampling_rate_hz, par["window_type"],
                                                     par["window_length_samples"], par["overlap_percent"],
                                                     par["detrend"], par["average"], par["scaling"])
            else:
                spectrum = backend._welch(self.time_data, None, self.sampling_rate_hz,
                                          par["window_type"], par["window_length_samples"],
                                          par["overlap_percent"], par["detrend"], par["average"],
                                          par["scaling"])
            if spectrum.ndim == 1:
                spectrum = spectrum[:, None]
            fft_length = par["window_length_samples"]
        else:
            fft_length = (next_fast_len(self.length_samples, True)
                          if par["pad_to_fast_length"] else self.length_samples)
            if par["smoothing"] != 0:
                raise NotImplementedError("spectrum smoothing is outside the GPU hot path")
            if hasattr(self, "window"):
                raise NotImplementedError("time-windowed signals are outside the GPU hot path")
            scaling = self.spectrum_scaling
            norm = scaling.fft_norm()
            scale = 1.0 if norm == "backward" else (
                1.0 / fft_length if norm == "forward" else fft_length**-0.5)
            spectrum = backend.rfft_spectrum(self.time_data, fft_length, scale)
            if scaling.has_physical_units():
                # helpers/spectrum_utilities.py:268-328 (per-bin scalars on the device result)
                factor = scaling.get_scaling_factor(fft_length, self.sampling_rate_hz, None)
                spectrum[0] /= 2**0.5
                if fft_length % 2 == 0:
                    spectrum[-1] /= 2**0.5
                if not scaling.is_amplitude_scaling():
                    spectrum = np.abs(spectrum) ** 2
                spectrum = spectrum * factor
        freqs = np.fft.rfftfreq(fft_length, 1 / self.sampling_rate_hz)
        if self.activate_cache:
            self.spectrum = [freqs.copy(), spectrum.copy()]
            self.__spectrum_state_update = False
        return freqs, spectrum

    def _short_estimate(self, par) -> bool:
        """Would backend._welch / _csm_welch send this signal's Welch estimate through the float64 kernels (fewer than
        128 frames ...)?  Those take the host arrays."""
        W = par["window_length_samples"]
        if W not in [2**k for k in range(3, 19)] or not (0 <= par["overlap_percent"] < 100):
            return True  # (let the host path raise the reference's assertion)
        hop = W - int(par["overlap_percent"] / 100 * W)
        n_frames = int(np.ceil(len(self) / hop))
        return backend._x64_short(backend.SPEC_PRECISION, self.number_of_channels, n_frames, W, par["average"])

    def get_csm(self, force_computation=False, on_device: bool = False):
        """-> (freqs_hz, csm (bins, channels, channels)).  on_device=True (an extension; Welch method only):
        the matrix stays in HBM, csm is a backend.DeviceCSM handle for the device beamformer map."""
        assert self.number_of_channels > 1, (
            "Cross spectral matrix can only be computed when at least two channels are available")
        if on_device:
            assert self.spectrum_method == SpectrumMethod.WelchPeriodogram, \
                "a device-resident CSM is built for the Welch method"
            par = self._spectrum_parameters
            dc = backend._csm_welch_device(self.device_samples if self.on_device else self.time_data,
                                           self.sampling_rate_hz, par["window_length_samples"],
                                           par["window_type"], par["overlap_percent"], par["detrend"],
                                           par["average"], par["scaling"])
            return dc.freqs_hz.copy(), dc
        if not (not hasattr(self, "csm") or force_computation or self.__csm_state_update):
            return self.csm[0].copy(), self.csm[1].copy()
        par = self._spectrum_parameters
        if self.spectrum_method == SpectrumMethod.WelchPeriodogram:
            f, csm = backend._csm_welch(self.time_data, self.sampling_rate_hz,
                                        par["window_length_samples"], par["window_type"],
                                        par["overlap_percent"], par["detrend"], par["average"],
                                        par["scaling"])
        else:
            # the spectrum is taken with FFTBackward scaling, the scaling is applied to the CSM
            old_scaling = self.spectrum_scaling
            self.spectrum_scaling = SpectrumScaling.FFTBackward
            try:
                f, sp = self.get_spectrum()
            finally:
                self.spectrum_scaling = old_scaling
            csm = backend._csm_fft(sp, old_scaling, None, self.sampling_rate_hz)
        if self.activate_cache:
            self.csm = [f.copy(), csm.copy()]
            self.__csm_state_update = False
        return f, csm

    def get_spectrogram(self, force_computation: bool = False, on_device: bool = False):
        """-> (time_s, freqs_hz, stft (bins, frames, channels)).  on_device=True (an extension): the spectrogram
        stays in HBM -- stft is a backend.DeviceSTFT for transforms.istft and the spectrogram features; its
        `to_host()` is the array."""
        par = self._spectrogram_parameters
        if on_device:
            return backend._stft_device(self.to_device().device_samples, self.sampling_rate_hz,
                                        par["window_length_samples"], par["window_type"], par["overlap_percent"],
                                        par["fft_length_samples"], par["detrend"], par["padding"], par["scaling"], True)
        if not (not hasattr(self, "spectrogram") or force_computation
                or self.__spectrogram_state_update):
            return tuple(a.copy() for a in self.spectrogram)
        if self.on_device:
            out = backend._stft_device(self.device_samples, self.sampling_rate_hz, par["window_length_samples"],
                                       par["window_type"], par["overlap_percent"], par["fft_length_samples"],
                                       par["detrend"], par["padding"], par["scaling"], False)
        else:
            out = backend._stft(self.time_data, self.sampling_rate_hz, par["window_length_samples"],
                                par["window_type"], par["overlap_percent"],
                                par["fft_length_samples"], par["detrend"], par["padding"],
                                par["scaling"])
        self.__spectrogram_state_update = False
        if self.activate_cache:
            self.spectrogram = deepcopy(out)
        return out[0], out[1], out[2]

    # ---- copies ---------------------------------------------------------------
    def copy(self):
        return deepcopy(self)

    def _get_data(self):
        return self.time_data

    def _set_data(self, data) -> None:
        self.time_data = data

    def _create_copy_with_new_data(self, data):
        return self.copy_with_new_time_data(data)

    def _update_state(self) -> None:
        self.__update_state()

    def _device_result(self, dev):
        """A signal of this object's type around device-resident result samples, with this signal's parameters --
        or, for a type that constrains its amplitude, around their host copy (the peak has to be inspected)."""
        if self.constrain_amplitude:
            return self.copy_with_new_time_data(backend._interleaved_f64(dev.to_planar()))
        new_signal = type(self).from_planar_f32(dev, self.sampling_rate_hz)
        new_signal.calibrated_signal = self.calibrated_signal
        new_signal.activate_cache = self.activate_cache
        new_signal._spectrum_parameters = deepcopy(self._spectrum_parameters)
        new_signal._spectrogram_parameters = deepcopy(self._spectrogram_parameters)
        new_signal._after_init()
        return new_signal

    def copy_with_new_time_data(self, new_time_data) -> "Signal":
        if isinstance(new_time_data, np.ndarray) and new_time_data.base is not None:
            new_time_data = new_time_data.copy()  # the new signal owns its samples
        new_signal = Signal.from_time_data(new_time_data, self.sampling_rate_hz,
                                           self.constrain_amplitude)
        new_signal.calibrated_signal = self.calibrated_signal
        new_signal.activate_cache = self.activate_cache
        new_signal._spectrum_parameters = deepcopy(self._spectrum_parameters)
        new_signal._spectrogram_parameters = deepcopy(self._spectrogram_parameters)
        return new_signal
