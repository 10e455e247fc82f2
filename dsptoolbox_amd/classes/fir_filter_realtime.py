"""Block-streaming FIR filters behind the reference's real-time API
(dsptoolbox/classes/fir_filter_realtime.py:75-335).

The reference keeps, per channel, an FFT-sized input buffer (overlap-save) or a frequency-domain
delay line of uniform partitions, and returns for every block the last `blocksize` samples of
the circular convolution -- i.e. the next `blocksize` samples of the causal convolution of the
stream with the impulse response.  Here the stream state is the last T - 1 input samples of each
channel; a block is filtered by the device FIR kernels (ds_fir_ola, the same overlap-save kernels
as Filter.filter_signal) over [history | block] and its last `blocksize` outputs are returned.
Same results as the reference wherever the reference computes the convolution; two defects of the
reference are NOT reproduced (the test infrastructure restates them literally and pins them to
the reference's outputs):
  * FIRFilterOverlapSave: next_fast_len(T + blocksize) may be odd; the reference's irfft is called
    without a length and then returns L - 1 points, so its blocks are not the convolution
    (:137-139).  This class warns once in prepare() and returns the convolution.
  * FIRUniformPartitioned: ONE delay-line index for all channels (:227-229); driven channel after
    channel the reference is correct only when n_channels = 1 (mod n_partitions).  Here every
    channel has its own state.
"""

from __future__ import annotations

import abc
import warnings

import numpy as np
import scipy.fft as _sfft

from .. import backend
from ..standard.enums import FilterCoefficientsType
from .filter import Filter
from .signal import Signal


class RealtimeFilter(abc.ABC):
    """classes/realtime_filter.py:4-40"""

    @abc.abstractmethod
    def process_sample(self, x: float, channel: int):
        ...

    @abc.abstractmethod
    def reset_state(self):
        ...

    @abc.abstractmethod
    def set_n_channels(self, n_channels: int):
        ...


def _stream_block(fir: np.ndarray, hist: np.ndarray, block: np.ndarray):
    """fir (T,), hist (T-1, C), block (B, C) -> (y (B, C), new history).  Device FIR over
    [history | block]; the last B outputs are the block's share of the causal convolution."""
    xx = np.concatenate([hist, block], axis=0)
    y = backend._lfilter_fir(fir, [1.0], xx)
    return y[-block.shape[0]:, :], xx[xx.shape[0] - hist.shape[0]:, :].copy()


class FIRFilterOverlapSave(RealtimeFilter):
    """Convolution of an FIR filter with the overlap-save scheme, block by block."""

    def __init__(self, b):
        b = np.asarray(b)
        assert b.ndim == 1, "A single dimension should be provided"
        self.fir = b

    @staticmethod
    def from_filter(fir: Filter):
        assert fir.is_fir, "Only valid for FIR filters"
        b, _ = fir.get_coefficients(FilterCoefficientsType.Ba)
        return FIRFilterOverlapSave(b)

    def prepare(self, blocksize_samples: int, n_channels: int):
        self.blocksize = blocksize_samples
        self.total_length = _sfft.next_fast_len(len(self.fir) + blocksize_samples, True)
        if self.total_length % 2:
            warnings.warn(
                f"next_fast_len({len(self.fir)} + {blocksize_samples}) = {self.total_length} is odd: the "
                "reference's block output is not the convolution in this case (irfft without length); "
                "this implementation returns the convolution")
        self._hist = np.zeros((len(self.fir) - 1, n_channels))

    def process_block(self, block, channel: int):
        block = np.asarray(block, dtype=np.float64)
        assert block.ndim == 1 and len(block) == self.blocksize, "block must be 1D with the prepared block size"
        y, h = _stream_block(self.fir, self._hist[:, channel:channel + 1], block[:, None])
        self._hist[:, channel] = h[:, 0]
        return y[:, 0]

    def process_sample(self, x: float, channel: int):
        raise NotImplementedError("The convolution can only done via block-processing")

    def reset_state(self):
        self._hist.fill(0.0)

    def set_n_channels(self, n_channels: int):
        raise NotImplementedError("Use prepare method for setting the filter")


class FIRUniformPartitioned(FIRFilterOverlapSave):
    """Overlap-save FIR with uniform filter partitions (for long impulse responses)."""

    def __init__(self, fir):
        fir = np.asarray(fir)
        assert fir.ndim == 1
        self.fir = fir

    @staticmethod
    def from_filter(fir: Filter):
        assert fir.is_fir, "Only valid for FIR filters"
        b, _ = fir.get_coefficients(FilterCoefficientsType.Ba)
        return FIRUniformPartitioned(b)

    def prepare(self, blocksize_samples: int, n_channels: int):
        self.blocksize = blocksize_samples
        self.fft_size = blocksize_samples * 2
        self.n_partitions = len(self.fir) // self.blocksize + 1
        self._hist = np.zeros((len(self.fir) - 1, n_channels))


class FIRUniformPartitionedMultichannel(FIRUniformPartitioned):
    """Uniformly partitioned overlap-save FIR, one impulse response per channel, all channels
    per call: block (time samples, channels) -> (time samples, channels)."""

    def __init__(self, fir):
        # "standard form" exactly as the reference (:262): Signal's constructor orients the array
        # and scales it to a peak of 1 when the peak exceeds 1
        self.fir = Signal.from_time_data(fir, 10000).time_data

    def prepare(self, blocksize_samples: int):  # type: ignore[override]
        self.blocksize = blocksize_samples
        self.fft_size = blocksize_samples * 2
        self.n_partitions = self.fir.shape[0] // self.blocksize + 1
        self.n_channels = self.fir.shape[1]
        self._hist = np.zeros((self.fir.shape[0] - 1, self.n_channels))

    def process_block(self, block):  # type: ignore[override]
        block = np.asarray(block, dtype=np.float64)
        assert block.shape == (self.blocksize, self.n_channels), \
            "block must have shape (block size, channels) with all channels"
        out = np.empty_like(block)
        for ch in range(self.n_channels):
            y, h = _stream_block(self.fir[:, ch], self._hist[:, ch:ch + 1], block[:, ch:ch + 1])
            out[:, ch] = y[:, 0]
            self._hist[:, ch] = h[:, 0]
        return out
