"""FilterBank (API mirror of dsptoolbox/classes/filterbank.py: ctor :33-66,
filter_signal :415-477 and the loop of filter_helpers.py:385-451, swap_filters :365-393,
filter_multiband_signal :479-532, get_ir :534-613, get_transfer_function :615-655).
A bank of equal-length FIR filters is applied in ONE device call: every input
block is transformed once and all band filters are applied on chip."""

from copy import deepcopy
from warnings import warn

import numpy as np

from .. import backend
from ..standard.enums import FilterBankMode
from .filter import Filter
from .multibandsignal import MultiBandSignal
from .signal import Signal


class FilterBank:
    def __init__(self, filters=None, same_sampling_rate: bool = True, info=None):
        self.same_sampling_rate = same_sampling_rate
        self.filters = filters if filters is not None else []
        self.info = info if info is not None else {}

    @property
    def filters(self):
        return self.__filters

    @filters.setter
    def filters(self, new_filters):
        assert type(new_filters) is list, "Filters have to be passed as a list"
        for f in new_filters:
            assert isinstance(f, Filter), f"{type(f)} is not a valid filter type"
        if new_filters and self.same_sampling_rate:
            fs = new_filters[0].sampling_rate_hz
            assert all(f.sampling_rate_hz == fs for f in new_filters), \
                "Not all filters have the same sampling rate"
        self.__filters = new_filters

    @property
    def sampling_rate_hz(self):
        if self.same_sampling_rate:
            return self.filters[0].sampling_rate_hz
        return [f.sampling_rate_hz for f in self.filters]

    @property
    def same_sampling_rate(self) -> bool:
        return self.__same_sampling_rate

    @same_sampling_rate.setter
    def same_sampling_rate(self, new_same):
        assert type(new_same) is bool, "Same sampling rate attribute must be a boolean"
        self.__same_sampling_rate = new_same

    @property
    def number_of_filters(self) -> int:
        return len(self.filters)

    def __len__(self):
        return len(self.filters)

    def __iter__(self):
        return iter(self.filters)

    def add_filter(self, filt: Filter, index: int = -1):
        fl = list(self.filters)
        fl.insert(len(fl) if index == -1 else index, filt)
        self.filters = fl
        return self

    def remove_filter(self, index: int = -1, return_filter: bool = False):
        fl = list(self.filters)
        f = fl.pop(index)
        self.filters = fl
        return f if return_filter else self

    def swap_filters(self, new_order):
        """Rearranges the filters in the new given order (filterbank.py:365-393)."""
        new_order = np.array(new_order).squeeze()
        assert new_order.ndim == 1, "Too many or too few dimensions are given in the new arrangement vector"
        assert self.number_of_filters == len(new_order), "The number of filters does not match"
        assert all(new_order < self.number_of_filters) and all(new_order >= 0), \
            f"Indexes of new filters have to be in [0, {self.number_of_filters - 1}]"
        assert len(np.unique(new_order)) == len(new_order), "There are repeated indexes in the new order vector"
        self.filters = [self.filters[i] for i in new_order]
        return self

    def copy(self):
        return deepcopy(self)

    def initialize_zi(self, number_of_channels: int = 1):
        """Initial state of every filter for the given number of channels (filterbank.py:126-138)."""
        for f in self.filters:
            f.initialize_zi(number_of_channels)
        return self

    def filter_signal(self, signal: Signal, mode: FilterBankMode, activate_zi: bool = False,
                      zero_phase: bool = False):
        """Parallel -> MultiBandSignal; Sequential / Summed -> Signal."""
        if type(signal) is MultiBandSignal:
            raise TypeError("This method only supports Signal objects. Use "
                            "filter_multiband_signal() for multirate parallel filtering")
        if mode in (FilterBankMode.Sequential, FilterBankMode.Summed):
            assert self.same_sampling_rate, \
                "Multirate filtering is not valid for sequential or summed filtering"
        assert np.all(signal.sampling_rate_hz == self.sampling_rate_hz), \
            "Sampling rates do not match"
        if zero_phase:
            assert not activate_zi, "Zero-phase filtering and zi cannot be used at the same time"
        if activate_zi:
            if not hasattr(self.filters[0], "zi"):
                self.initialize_zi(signal.number_of_channels)
            if len(self.filters[0].zi) != signal.number_of_channels:
                self.initialize_zi(signal.number_of_channels)
        if mode not in (FilterBankMode.Parallel, FilterBankMode.Sequential, FilterBankMode.Summed):
            raise ValueError("Invalid filter bank apply mode")
        for f in self.filters:
            if not f.is_fir:
                raise NotImplementedError("IIR filters are outside the FFT-batchable GPU hot path")
        if activate_zi or zero_phase:
            # per-filter state / two-pass filtering: the reference's own loop
            # (filter_helpers.py:385-451), one device convolution (or two) per filter
            if mode == FilterBankMode.Parallel:
                bands = [f.filter_signal(signal, activate_zi=activate_zi, zero_phase=zero_phase)
                         for f in self.filters]
                return MultiBandSignal(bands, same_sampling_rate=self.same_sampling_rate)
            if mode == FilterBankMode.Sequential:
                out = signal.copy()
                for f in self.filters:
                    out = f.filter_signal(out, activate_zi=activate_zi, zero_phase=zero_phase)
                return out
            acc = np.zeros((signal.time_data.shape[0], signal.number_of_channels, len(self.filters)))
            for n, f in enumerate(self.filters):
                acc[:, :, n] = f.filter_signal(signal, activate_zi=activate_zi,
                                               zero_phase=zero_phase).time_data
            return signal.copy_with_new_time_data(np.sum(acc, axis=-1))
        taps = [f.ba[0] for f in self.filters]
        n_taps = max(len(t) for t in taps)
        # zero-extending a FIR filter at the end does not change its output
        taps = [np.concatenate([t, np.zeros(n_taps - len(t))]) for t in taps]
        ds_mode = {FilterBankMode.Parallel: backend.DS_FB_PARALLEL,
                   FilterBankMode.Sequential: backend.DS_FB_SEQUENTIAL,
                   FilterBankMode.Summed: backend.DS_FB_SUMMED}[mode]
        if signal.on_device and not any(np.iscomplexobj(t) for t in taps):
            # device-resident samples: the whole bank in one call, its output stays in HBM -- a Parallel bank's
            # MultiBandSignal holds K device-resident bands (slices of ONE buffer), downloaded when asked for
            y = backend.fir_filter_bank_device(signal.device_samples, taps, ds_mode)
            if mode == FilterBankMode.Parallel:
                return MultiBandSignal([signal._device_result(b) for b in y], same_sampling_rate=self.same_sampling_rate)
            return signal._device_result(y)
        y = backend.fir_filter_bank(signal.time_data, taps, ds_mode)
        if mode == FilterBankMode.Parallel:
            bands = [signal.copy_with_new_time_data(np.ascontiguousarray(y[k]))
                     for k in range(len(taps))]
            return MultiBandSignal(bands, same_sampling_rate=self.same_sampling_rate)
        return signal.copy_with_new_time_data(y)

    def filter_multiband_signal(self, mbsignal: MultiBandSignal, activate_zi: bool = False,
                                zero_phase: bool = False) -> MultiBandSignal:
        """Band n of the MultiBandSignal through filter n, every channel (filterbank.py:479-532)."""
        assert np.all(mbsignal.sampling_rate_hz == self.sampling_rate_hz), "Sampling rates do not match"
        if zero_phase:
            assert not activate_zi, "Zero-phase filtering and zi cannot be used at the same time"
        if activate_zi:
            if not hasattr(self.filters[0], "zi"):
                self.initialize_zi(mbsignal.number_of_channels)
            if len(self.filters[0].zi) != mbsignal.number_of_channels:
                self.initialize_zi(mbsignal.number_of_channels)
        new_sig = mbsignal.copy()
        bands = list(new_sig.bands)
        for n in range(mbsignal.number_of_bands):
            bands[n] = self.filters[n].filter_signal(mbsignal.bands[n], channels=None, activate_zi=activate_zi,
                                                     zero_phase=zero_phase)
        new_sig.bands = bands
        return new_sig

    def get_ir(self, length_samples: int, mode: FilterBankMode, zero_phase: bool = False):
        """Impulse response of the bank: a unit impulse through filter_signal (filterbank.py:534-613).
        Parallel -> MultiBandSignal, Sequential / Summed -> ImpulseResponse."""
        from .impulse_response import ImpulseResponse

        def dirac(n, fs):
            d = np.zeros((n, 1))
            d[0, 0] = 1.0
            return ImpulseResponse(None, d, fs)

        if not self.same_sampling_rate:
            assert mode == FilterBankMode.Parallel, "Multirate filter bank can only deliver an IR in parallel mode"
            mb = MultiBandSignal(same_sampling_rate=False)
            for f, sr in zip(self.filters, self.sampling_rate_hz):
                mb.add_band(f.filter_signal(dirac(length_samples, sr), zero_phase=zero_phase))
            return mb
        max_order = max([0] + [b.order for b in self.filters])
        if max_order > length_samples:
            warn(f"Filter order {max_order} is longer than {length_samples}.The length will be adapted to be "
                 "100 samples longer than the longest filter")
            length_samples = max_order + 100
        return self.filter_signal(dirac(length_samples, self.sampling_rate_hz), mode, zero_phase=zero_phase)

    def get_transfer_function(self, frequency_vector_hz, mode: FilterBankMode) -> np.ndarray:
        """Complex transfer function of the bank (filterbank.py:615-655): Parallel -> (frequency, filter);
        Sequential -> the product; Summed -> ONE PLUS the sum (the reference starts its sum from ones)."""
        frequency_vector_hz = np.asarray(frequency_vector_hz)
        assert frequency_vector_hz.ndim == 1, "Frequency vector can only have one dimension"
        if mode not in (FilterBankMode.Parallel, FilterBankMode.Sequential, FilterBankMode.Summed):
            raise ValueError("No valid mode")
        for f in self.filters:
            assert frequency_vector_hz.max() <= f.sampling_rate_hz / 2, \
                "Queried frequency vector has values larger than nyquist"
            if not f.is_fir:
                raise NotImplementedError("IIR filters are outside the FFT-batchable GPU hot path")
        if self.same_sampling_rate:  # every filter in one device call
            h = backend.fir_transfer_function([f.ba[0] for f in self.filters], frequency_vector_hz, self.sampling_rate_hz)
        else:
            h = np.stack([f.get_transfer_function(frequency_vector_hz) for f in self.filters], axis=1)
        if mode == FilterBankMode.Parallel:
            return h
        if mode == FilterBankMode.Sequential:
            return np.prod(h, axis=1)
        return 1.0 + np.sum(h, axis=1)
