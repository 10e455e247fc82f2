"""Channel-axis operations of the containers whose payload is a (length, channels) array.

Signal (samples x channels) and Spectrum (bins x channels) both keep one 2-D array with the
channels on the last axis; everything that only rearranges that axis lives here, written once
against three hooks the container supplies:

    _get_data()                      the (length, channels) array
    _set_data(array)                 install a rearranged array in place
    _create_copy_with_new_data(a)    a new container of the same kind around `a`
    _update_state()                  drop whatever the container cached for the old layout

Public names and failure type (AssertionError on a bad channel argument) follow the reference's
mixin, dsptoolbox/classes/_multichannel_data.py:6-118, so that user code moves over unchanged.
"""

from abc import ABC, abstractmethod

import numpy as np


class MultichannelData(ABC):
    # ---- hooks ---------------------------------------------------------------------------
    @abstractmethod
    def _get_data(self):
        raise NotImplementedError

    @abstractmethod
    def _set_data(self, data) -> None:
        raise NotImplementedError

    @abstractmethod
    def _create_copy_with_new_data(self, data):
        raise NotImplementedError

    @abstractmethod
    def _update_state(self) -> None:
        raise NotImplementedError

    # ---- shape ---------------------------------------------------------------------------
    @property
    def number_of_channels(self) -> int:
        return int(self._get_data().shape[-1])

    def __len__(self) -> int:
        return int(self._get_data().shape[0])

    # ---- helpers -------------------------------------------------------------------------
    def _channel_vector(self, channels) -> np.ndarray:
        """Any scalar / list / array selection as a flat integer index vector."""
        idx = np.asarray(channels)
        if idx.ndim != 1:
            idx = idx.reshape(-1) if idx.size == max(idx.shape, default=1) else idx
        assert idx.ndim == 1, "a channel selection is a scalar or a one-dimensional sequence"
        return idx

    def _replace(self, data: np.ndarray):
        self._set_data(data)
        self._update_state()
        return self

    # ---- operations ----------------------------------------------------------------------
    def remove_channel(self, channel_number: int = -1):
        """Drop one channel in place (the last one by default); the only channel cannot go."""
        n_ch = self.number_of_channels
        assert n_ch > 1, "the container has a single channel; it cannot be removed"
        victim = n_ch - 1 if channel_number == -1 else channel_number
        assert victim < n_ch, f"channel {victim} requested, valid channels are 0 ... {n_ch - 1}"
        keep = np.ones(n_ch, dtype=bool)
        keep[victim] = False
        return self._replace(self._get_data()[:, keep])

    def swap_channels(self, new_order):
        """Reorder the channels in place; `new_order` must be a permutation of 0 ... C-1."""
        order = self._channel_vector(new_order)
        n_ch = self.number_of_channels
        assert order.size == n_ch, f"{order.size} indices given for {n_ch} channels"
        seen = np.zeros(n_ch, dtype=np.int64)
        in_range = (order >= 0) & (order < n_ch)
        assert bool(in_range.all()), f"channel indices must lie in 0 ... {n_ch - 1}"
        np.add.at(seen, order, 1)
        assert bool((seen == 1).all()), "every channel has to appear exactly once in the new order"
        return self._replace(self._get_data()[:, order])

    def get_channels(self, channels):
        """A new container holding the selected channels (in the order given)."""
        return self._create_copy_with_new_data(self._get_data()[:, self._channel_vector(channels)])

    def sum_channels(self):
        """A new one-channel container: the sum over the channel axis."""
        total = self._get_data().sum(axis=1)
        return self._create_copy_with_new_data(total[:, None])
