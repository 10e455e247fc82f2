"""Multi-GPU layer: one process per GPU, independent units sharded across ranks.

The hot path has no exchange step: output channels
(Welch / H1, FIR), items (deconvolution batch) or frequency bins are independent,
so every rank runs the single-GPU path on its contiguous shard.  The collectives
are a broadcast of the shared input (sweep channel, FIR taps, inverse spectrum)
and an optional gather of the results -- on device buffers both go over RCCL /
xGMI through the library's own communicator (ds_bcast, ds_allgather).  What the
ranks exchange on the HOST (the 128-byte RCCL id, shapes, small result arrays) goes
through `rendezvous.Exchange` -- a standard-library TCP star by default; the package
imports no torch.
"""

from __future__ import annotations

import ctypes as C
import os
import struct

import numpy as np

from . import rendezvous

_EX: rendezvous.Exchange | None = None


def init(exchange: rendezvous.Exchange | None = None) -> rendezvous.Exchange:
    """Install the host-side exchange of this process (default: the launcher's environment,
    rendezvous.from_environment).  Idempotent without arguments."""
    global _EX
    if exchange is not None:
        _EX = exchange
    elif _EX is None:
        _EX = rendezvous.from_environment()
    return _EX


def shutdown() -> None:
    global _EX
    if _EX is not None:
        _EX.close()
    _EX = None


def shard_range(n_units: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous balanced split of n_units: the first (n_units % world) ranks get one
    more.  Empty shards are allowed (start == stop)."""
    assert world_size >= 1 and 0 <= rank < world_size
    base, extra = divmod(n_units, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def world() -> tuple[int, int]:
    """(rank, world_size): the installed exchange, else the launcher's environment."""
    if _EX is not None:
        return _EX.rank, _EX.world
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


# Wire format of a host array: a fixed binary header -- magic, dtype code, ndim, dims -- then the
# raw bytes.  Nothing in it is evaluated: the dtype comes from a closed table and the dimensions
# are checked against the payload length, so a peer can at worst send a wrong array.
_WIRE_MAGIC = b"DSA1"
_WIRE_DTYPES = ("<f4", "<f8", "<c8", "<c16", "<i4", "<i8", "<u4", "<u8", "|u1", "|i1", "|b1", "<i2", "<u2")
_WIRE_MAX_NDIM = 8


def _pack(arr: np.ndarray) -> bytes:
    a = np.asarray(arr)
    if not a.flags.c_contiguous:  # (ascontiguousarray would turn a 0-d array into 1-d)
        a = np.ascontiguousarray(a)
    code = a.dtype.newbyteorder("<").str if a.dtype.byteorder == ">" else a.dtype.str
    if a.dtype.byteorder == ">":
        a = a.astype(a.dtype.newbyteorder("<"))
    if code not in _WIRE_DTYPES:
        raise TypeError(f"dtype {a.dtype} cannot travel through the host exchange")
    if a.ndim > _WIRE_MAX_NDIM:
        raise ValueError("too many dimensions for the host exchange")
    head = _WIRE_MAGIC + struct.pack("<BB", _WIRE_DTYPES.index(code), a.ndim) + struct.pack(f"<{a.ndim}q", *a.shape)
    return head + a.tobytes()


def _unpack(b: bytes) -> np.ndarray:
    if len(b) < 6 or b[:4] != _WIRE_MAGIC:
        raise ValueError("host exchange: not an array message")
    code, ndim = struct.unpack_from("<BB", b, 4)
    if code >= len(_WIRE_DTYPES) or ndim > _WIRE_MAX_NDIM or len(b) < 6 + 8 * ndim:
        raise ValueError("host exchange: malformed array header")
    shape = struct.unpack_from(f"<{ndim}q", b, 6)
    dtype = np.dtype(_WIRE_DTYPES[code])
    count = 1
    for d in shape:
        if d < 0:
            raise ValueError("host exchange: negative dimension")
        count *= d
    off = 6 + 8 * ndim
    if count * dtype.itemsize != len(b) - off:
        raise ValueError("host exchange: array header does not match the payload length")
    return np.frombuffer(b, dtype=dtype, count=count, offset=off).reshape(shape).copy()


def broadcast_array(arr: np.ndarray | None, src: int = 0) -> np.ndarray:
    """Host-array broadcast through the host exchange (small shared inputs; a device buffer
    travels with broadcast_device over xGMI instead)."""
    rank, ws = world()
    if ws == 1:
        return arr
    ex = init()
    return _unpack(ex.broadcast_bytes(_pack(arr) if rank == src else None, src=src))


def gather_channel_shards(local: np.ndarray, n_total: int, axis: int = -1) -> np.ndarray:
    """All-gather per-rank host result slices (split by shard_range along `axis`) into the full
    array on every rank.  Results that live on the device are gathered there: allgather_device."""
    rank, ws = world()
    if ws == 1:
        return local
    parts = [_unpack(b) for b in init().allgather_bytes(_pack(local))]
    out = np.concatenate([p for p in parts if p.shape[axis] > 0], axis=axis)
    assert out.shape[axis] == n_total, (out.shape, n_total)
    return out


def welch_transfer_function_sharded(output_td, input_td, fs_hz: int, window_length_samples: int,
                                    mode: str, compute=None, gather: bool = True, **params):
    """compute_transfer_function with the output channels sharded across ranks.
    `compute(output_shard, input, fs, W, mode, **params) -> (tf, coh)` defaults to the
    HIP path; a one-channel input is used by every rank (broadcast it first if only rank 0
    holds it), a per-channel input is sharded together with the output."""
    if compute is None:
        from . import backend
        compute = backend.welch_transfer_function
    rank, ws = world()
    n_cy = output_td.shape[1]
    a, b = shard_range(n_cy, ws, rank)
    x = input_td if input_td.shape[1] == 1 else input_td[:, a:b]
    nb = window_length_samples // 2 + 1
    if b > a:
        tf, coh = compute(output_td[:, a:b], x, fs_hz, window_length_samples, mode, **params)
    else:
        tf, coh = np.zeros((nb, 0), dtype=np.complex128), np.zeros((nb, 0))
    if not gather:
        return tf, coh
    return gather_channel_shards(tf, n_cy, 1), gather_channel_shards(coh, n_cy, 1)


def csm_welch_sharded(time_data, sampling_rate_hz: int, window_length_samples: int, compute=None,
                      gather: bool = True, **params):
    """Welch cross-spectral matrix split by frequency bins: the
    matrix does not shard by channel (all pairs are needed), so every rank transforms all
    channels and keeps the bins of its shard -- redundant STFTs, no reduction between ranks.
    `compute(td, fs, W, bin_start, bin_stop, **params) -> (bins, C, C)` defaults to the HIP path.
    -> (f, csm (B, C, C)) on every rank (or this rank's rows with gather=False)."""
    if compute is None:
        from . import backend
        from .standard.enums import SpectrumScaling, Window

        def compute(td, fs, W, b0, b1, window_type=Window.Hann, overlap_percent=50.0, detrend=True,
                    scaling=SpectrumScaling.FFTBackward):
            return backend._csm_welch_bins(td, fs, W, window_type, overlap_percent, detrend, scaling, b0, b1)
    rank, ws = world()
    nb = window_length_samples // 2 + 1
    a, b = shard_range(nb, ws, rank)
    local = compute(time_data, sampling_rate_hz, window_length_samples, a, b, **params)
    f = np.fft.rfftfreq(window_length_samples, 1 / sampling_rate_hz)
    if not gather:
        return f[a:b], local
    return f, gather_channel_shards(local, nb, axis=0)


def fir_filter_bank_sharded(x, taps_list, mode: int, compute=None, gather: bool = True):
    """FIR filter bank with the independent units sharded across ranks (SURVEY section 8(e),
    config 3): Parallel -> the bands, Sequential / Summed -> the channels.  `compute(x, taps, mode)`
    defaults to the HIP path (backend.fir_filter_bank: Parallel -> (K, N, C), else (N, C)); the
    input / taps are expected on every rank (broadcast them first if only rank 0 holds them)."""
    from . import backend
    if compute is None:
        compute = backend.fir_filter_bank
    rank, ws = world()
    x = np.asarray(x)
    n, n_ch = x.shape
    if mode == backend.DS_FB_PARALLEL:
        a, b = shard_range(len(taps_list), ws, rank)
        local = compute(x, taps_list[a:b], mode) if b > a else np.zeros((0, n, n_ch))
        return gather_channel_shards(local, len(taps_list), axis=0) if gather else local
    a, b = shard_range(n_ch, ws, rank)
    local = compute(x[:, a:b], taps_list, mode) if b > a else np.zeros((n, 0))
    return gather_channel_shards(local, n_ch, axis=1) if gather else local


def spectral_division_sharded(num_td, n_fft: int, inverse_spectrum, n_out: int, compute=None,
                              gather: bool = True):
    """Batched spectral division (M, N, C) with the items sharded across ranks (config 5); the
    inverse spectrum (32 KB) is expected on every rank."""
    if compute is None:
        from . import backend
        compute = backend.spectral_division
    rank, ws = world()
    num_td = np.asarray(num_td)
    assert num_td.ndim == 3, "a batch of items (M, N, C) is sharded; single items are not"
    m = num_td.shape[0]
    a, b = shard_range(m, ws, rank)
    local = compute(num_td[a:b], n_fft, inverse_spectrum, n_out) if b > a else \
        np.zeros((0, n_out, num_td.shape[2]))
    return gather_channel_shards(local, m, axis=0) if gather else local


def init_library_comm(ctx) -> bool:
    """Create the library's RCCL communicator (ds_comm_*): rank 0 makes the 128-byte id, the
    host exchange hands it to every rank.  Returns False for a single-rank world."""
    rank, ws = world()
    if ws == 1:
        return False
    ident = C.create_string_buffer(128)
    if rank == 0:
        ctx.check(ctx.lib.ds_comm_unique_id(ident), "ds_comm_unique_id")
    raw = init().broadcast_bytes(ident.raw if rank == 0 else None, src=0)
    ctx.check(ctx.lib.ds_comm_init(ctx.handle, ws, rank, raw), "ds_comm_init")
    return True


def broadcast_device(ctx, dptr: int, nbytes: int, root: int = 0):
    """RCCL broadcast of a device buffer on the context's stream (xGMI)."""
    ctx.check(ctx.lib.ds_bcast(ctx.handle, C.c_void_p(dptr), nbytes, root), "ds_bcast")


def allgather_device(ctx, send_ptr: int, recv_ptr: int, nbytes_per_rank: int):
    """RCCL all-gather of equal-sized device shards on the context's stream: rank r's
    `nbytes_per_rank` bytes land at recv + r * nbytes_per_rank on every rank.  Uneven shards
    (shard_range) are padded to the largest by the caller."""
    ctx.check(ctx.lib.ds_allgather(ctx.handle, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), nbytes_per_rank),
              "ds_allgather")
