"""MultiBandSignal: the band outputs of a parallel filter bank
(API mirror of dsptoolbox/classes/multibandsignal.py:12-599, single-rate part)."""

from copy import deepcopy
from warnings import warn

import numpy as np

from .signal import Signal


class MultiBandSignal:
    def __init__(self, bands=None, same_sampling_rate: bool = True, info=None):
        self.same_sampling_rate = same_sampling_rate
        self.bands = bands if bands is not None else []
        self.info = info if info is not None else {}

    @property
    def sampling_rate_hz(self):
        return self.__sampling_rate_hz

    @sampling_rate_hz.setter
    def sampling_rate_hz(self, new_sampling_rate_hz):
        new_sampling_rate_hz = np.array(new_sampling_rate_hz)
        if self.same_sampling_rate:
            new_sampling_rate_hz = new_sampling_rate_hz.squeeze()
            assert new_sampling_rate_hz.ndim == 0, "MultiBandSignal has only one sample rate"
            self.__sampling_rate_hz = int(new_sampling_rate_hz)
        else:
            self.__sampling_rate_hz = [int(s) for s in np.atleast_1d(new_sampling_rate_hz)]

    @property
    def bands(self):
        return self.__bands

    @bands.setter
    def bands(self, new_bands):
        if new_bands is None:
            new_bands = []
        if type(new_bands) is tuple:
            new_bands = list(new_bands)
        assert type(new_bands) is list, "bands has to be a list"
        if new_bands:
            self.__number_of_channels = new_bands[0].number_of_channels
            complex_data = new_bands[0].time_data_imaginary is not None
            rates = []
            for s in new_bands:
                assert isinstance(s, Signal), f"{type(s)} is not a valid band type. Use Signal objects"
                assert s.number_of_channels == self.number_of_channels, \
                    "Signals have different number of channels. This behaviour is not supported"
                assert (s.time_data_imaginary is not None) == complex_data, \
                    "Some bands have imaginary time data and others do not. This behavior is not supported."
                rates.append(s.sampling_rate_hz)
            if self.same_sampling_rate:
                self.sampling_rate_hz = new_bands[0].sampling_rate_hz
                n0 = new_bands[0].length_samples
                for s in new_bands:
                    assert s.sampling_rate_hz == self.sampling_rate_hz, (
                        "Not all Signals have the same sampling rate. If you wish to create a "
                        "multirate system, set same_sampling_rate to False")
                    assert s.length_samples == n0, (
                        "The length of the bands is not always the same. This behaviour is not "
                        "supported if there is a constant sampling rate")
            else:
                self.sampling_rate_hz = rates
        self.__bands = new_bands

    @property
    def same_sampling_rate(self) -> bool:
        return self.__same_sampling_rate

    @same_sampling_rate.setter
    def same_sampling_rate(self, new_same):
        assert type(new_same) is bool, "Same sampling rate attribute must be a boolean"
        self.__same_sampling_rate = new_same

    @property
    def number_of_bands(self) -> int:
        return len(self.bands)

    @property
    def number_of_channels(self) -> int:
        return self.__number_of_channels

    @property
    def length_seconds(self) -> float:
        return self.bands[0].length_seconds if self.bands else 0.0

    @property
    def length_samples(self) -> int:
        return self.bands[0].length_samples if self.bands else 0

    @property
    def is_complex_signal(self) -> bool:
        """Do the bands carry imaginary time data?  False without bands (multibandsignal.py:262-274)."""
        if not self.bands:
            return False
        return self.bands[0].is_complex_signal

    def __iter__(self):
        return iter(self.bands)

    def __len__(self):
        return len(self.bands)

    def add_band(self, sig: Signal, index: int = -1):
        bands = list(self.bands)
        bands.insert(len(bands) if index == -1 else index, sig)
        self.bands = bands
        return self

    def remove_band(self, index: int = -1, return_band: bool = False):
        bands = list(self.bands)
        b = bands.pop(index)
        self.bands = bands
        return b if return_band else self

    def swap_bands(self, new_order):
        """Rearranges the bands in the new given order (multibandsignal.py:378-410)."""
        new_order = np.array(new_order).squeeze()
        assert new_order.ndim == 1, "Too many or too few dimensions are given in the new arrangement vector"
        assert self.number_of_bands == len(new_order), "The number of bands does not match"
        assert all(new_order < self.number_of_bands) and all(new_order >= 0), \
            f"Indexes of new bands have to be in [0, {self.number_of_bands - 1}]"
        assert len(np.unique(new_order)) == len(new_order), "There are repeated indexes in the new order vector"
        self.bands = [self.bands[i] for i in new_order]
        return self

    def get_all_bands(self, channel: int = 0):
        """ONE channel of every band as the channels of a Signal of the bands' own type (multibandsignal.py:463-518);
        bands of different sampling rates: (list of time vectors, list of sampling rates)."""
        complex_data = self.bands[0].time_data_imaginary is not None
        cols = [(b.time_data[:, channel] + 1j * b.time_data_imaginary[:, channel]) if complex_data
                else b.time_data[:, channel] for b in self.bands]
        if not self.same_sampling_rate:
            if complex_data:
                warn("Output is complex since signal data had imaginary part")
            return [c.copy() for c in cols], [b.sampling_rate_hz for b in self.bands]
        return type(self.bands[0])(None, np.stack(cols, axis=1), self.sampling_rate_hz)

    @property
    def on_device(self) -> bool:
        """Do all bands hold their samples in HBM (the output of a filter bank over a device-resident signal)?
        Nothing is downloaded until a band's `time_data` -- or `get_all_time_data()` -- is asked for."""
        return bool(self.bands) and all(b.on_device for b in self.bands)

    def collapse(self) -> Signal:
        """Sum of all bands as one Signal."""
        assert self.same_sampling_rate, "Collapsing is only available for same sampling rate bands"
        td, _ = self.get_all_time_data()
        return self.bands[0].copy_with_new_time_data(np.sum(td, axis=1))

    def get_all_time_data(self):
        """(time samples, band, channel) array and the sampling rate (multibandsignal.py:522-572).  Device-resident
        bands are materialised here, one band at a time (each download is dropped again unless the band had a
        host copy already: the whole bank's output exists once on the host, not twice)."""
        if not self.same_sampling_rate:
            return [(b.time_data, b.sampling_rate_hz) for b in self.bands]
        td = np.zeros((self.length_samples, self.number_of_bands, self.number_of_channels))
        for ind, b in enumerate(self.bands):
            if b.on_device and not b._has_host_copy:
                from .. import backend
                backend._interleaved_f64(b.device_samples.to_planar(), td[:, ind, :])
            else:
                td[:, ind, :] = b.time_data
        return td, self.sampling_rate_hz

    def copy(self):
        return deepcopy(self)
