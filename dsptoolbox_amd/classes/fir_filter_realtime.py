"""Block-streaming FIR filters behind the reference's real-time API
(dsptoolbox/classes/fir_filter_realtime.py:75-335).

Every class executes the reference's own algorithm, literally, on state that lives in device
buffers between calls (ds_fir_ols_step_dev / ds_fir_part_step_dev): the time-domain input
buffers, the frequency-domain delay line of partition spectra and the tap spectra are uploaded
or zeroed once in prepare(); a process_block call moves one block up and one block down.
"Literally" includes the two places where the reference's blocks are not the causal convolution
(both pinned by tests/golden/fir_stream.npz, outputs of the real reference):
  * FIRFilterOverlapSave calls irfft without a length (:137-139), so when
    next_fast_len(T + blocksize) is odd the inverse transform is one point short;
  * FIRUniformPartitioned keeps ONE delay-line index for all channels (:227-229), so driven
    channel after channel it is the convolution only when n_channels = 1 (mod n_partitions).
FIRUniformPartitionedMultichannel processes all channels in one launch sequence.
"""

from __future__ import annotations

import abc
import ctypes as C

import numpy as np
import scipy.fft as _sfft

from .._lib import DeviceBuffer, get_context
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


def _dev_zeros(ctx, nbytes: int) -> DeviceBuffer:
    buf = DeviceBuffer(ctx, max(int(nbytes), 4))
    ctx.check(ctx.lib.ds_memset(ctx.handle, C.c_void_p(buf.ptr), 0, max(int(nbytes), 4)), "ds_memset")
    return buf


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
        self.fir_spectrum = _sfft.rfft(self.fir, n=self.total_length, axis=0)
        self._n_channels = n_channels
        ctx = self._ctx = get_context()
        self._d_h = DeviceBuffer.from_array(ctx, np.ascontiguousarray(self.fir_spectrum, dtype=np.complex64))
        self._d_buf = _dev_zeros(ctx, 4 * self.total_length * n_channels)  # buffer[c][L], the reference's (L, C)
        self._d_blk = DeviceBuffer(ctx, 4 * blocksize_samples)
        self._d_out = DeviceBuffer(ctx, 4 * blocksize_samples)

    def process_block(self, block, channel: int):
        blk = np.ascontiguousarray(block, dtype=np.float32)
        assert blk.ndim == 1 and len(blk) == self.blocksize, "block must be 1D with the prepared block size"
        ctx = self._ctx
        ctx.upload(self._d_blk.ptr, blk)
        ctx.check(ctx.lib.ds_fir_ols_step_dev(
            ctx.handle, C.c_void_p(self._d_buf.ptr + 4 * self.total_length * int(channel)),
            C.c_void_p(self._d_blk.ptr), self.blocksize, self.total_length, C.c_void_p(self._d_h.ptr),
            C.c_void_p(self._d_out.ptr)), "ds_fir_ols_step_dev")
        return self._d_out.to_array((self.blocksize,), np.float32).astype(np.float64)

    def process_sample(self, x: float, channel: int):
        raise NotImplementedError("The convolution can only done via block-processing")

    def reset_state(self):
        ctx = self._ctx
        ctx.check(ctx.lib.ds_memset(ctx.handle, C.c_void_p(self._d_buf.ptr), 0,
                                    4 * self.total_length * self._n_channels), "ds_memset")

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
        self._prepare_partitions(np.asarray(self.fir)[:, None], n_channels)

    def _prepare_partitions(self, fir2d: np.ndarray, n_channels: int):
        """:189-212 / :284-309: partition spectra, the delay line and the input buffers."""
        bs = self.blocksize
        self.n_partitions = fir2d.shape[0] // bs + 1
        n_fir_ch = fir2d.shape[1]
        partitioned = np.zeros((bs, self.n_partitions, n_fir_ch))
        for n in range(self.n_partitions):
            partition = fir2d[n * bs:(n + 1) * bs]
            partitioned[:len(partition), n, :] = partition
        self.partitioned_spectrum = _sfft.rfft(partitioned, axis=0, n=self.fft_size)  # (bs + 1, P, Cf)
        self.buffer_ind = 0  # ONE index for all channels, as in the reference
        self._n_channels, self._n_fir_ch = n_channels, n_fir_ch
        ctx = self._ctx = get_context()
        self._d_h = DeviceBuffer.from_array(ctx, np.ascontiguousarray(self.partitioned_spectrum, dtype=np.complex64))
        self._delay_bytes = 8 * (bs + 1) * self.n_partitions * n_channels
        self._d_delay = _dev_zeros(ctx, self._delay_bytes)          # buffer_spectra (bs + 1, P, C)
        self._d_in = _dev_zeros(ctx, 4 * self.fft_size * n_channels)  # input_buffer as [c][2 bs]
        self._d_blk = DeviceBuffer(ctx, 4 * bs * n_channels)
        self._d_out = DeviceBuffer(ctx, 4 * bs * n_channels)

    def _step(self, blk_cn: np.ndarray, ch0: int):
        """blk_cn (n_call, bs) float32 -> (n_call, bs) float32; advances the shared index."""
        ctx, n_call = self._ctx, blk_cn.shape[0]
        ctx.upload(self._d_blk.ptr, blk_cn)
        ctx.check(ctx.lib.ds_fir_part_step_dev(
            ctx.handle, C.c_void_p(self._d_in.ptr), C.c_void_p(self._d_blk.ptr), self.blocksize, self._n_channels,
            int(ch0), n_call, C.c_void_p(self._d_h.ptr), self.n_partitions, self._n_fir_ch,
            C.c_void_p(self._d_delay.ptr), self.buffer_ind, C.c_void_p(self._d_out.ptr)), "ds_fir_part_step_dev")
        self.buffer_ind = (self.buffer_ind + 1) % self.n_partitions
        return self._d_out.to_array((n_call, self.blocksize), np.float32)

    def process_block(self, block, channel: int):
        blk = np.ascontiguousarray(block, dtype=np.float32)
        assert blk.ndim == 1 and len(blk) == self.blocksize, "block must be 1D with the prepared block size"
        return self._step(blk[None, :], channel)[0].astype(np.float64)

    def reset_state(self):
        ctx = self._ctx  # (the reference leaves buffer_ind where it is, :185-187)
        ctx.check(ctx.lib.ds_memset(ctx.handle, C.c_void_p(self._d_delay.ptr), 0, self._delay_bytes), "ds_memset")
        ctx.check(ctx.lib.ds_memset(ctx.handle, C.c_void_p(self._d_in.ptr), 0,
                                    4 * self.fft_size * self._n_channels), "ds_memset")


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
        self.n_channels = self.fir.shape[1]
        self._prepare_partitions(self.fir, self.n_channels)

    def process_block(self, block):  # type: ignore[override]
        block = np.asarray(block)
        assert block.shape == (self.blocksize, self.n_channels), \
            "block must have shape (block size, channels) with all channels"
        out = self._step(np.ascontiguousarray(block.T, dtype=np.float32), 0)  # every channel in one go
        return out.T.astype(np.float64)
