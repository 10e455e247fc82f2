"""ctypes binding of the C-ABI in include/dsptoolbox_amd.h.

The HIP library is the product: if it is missing, or there is no GPU, every
compute call raises -- there is no CPU fallback anywhere in this package.
"""

import ctypes as C
import os
import threading

import numpy as np

from ._build import LIB_PATH

c32_p = C.c_void_p  # ds_c32* (numpy complex64 buffers)
f32_p = C.c_void_p
ctx_p = C.c_void_p
i64 = C.c_int64

SIGNATURES = {
    "ds_version": (C.c_int, []),
    "ds_device_count": (C.c_int, []),
    "ds_init": (C.c_int, [C.c_int, C.POINTER(ctx_p)]),
    "ds_destroy": (None, [ctx_p]),
    "ds_last_error": (C.c_char_p, [ctx_p]),
    "ds_malloc": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "ds_free": (C.c_int, [ctx_p, C.c_void_p]),
    "ds_host_alloc": (C.c_int, [ctx_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "ds_host_free": (C.c_int, [ctx_p, C.c_void_p]),
    "ds_host_planar_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int]),
    "ds_host_interleave_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_int]),
    "ds_host_widen_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "ds_upload": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ds_download": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ds_memset": (C.c_int, [ctx_p, C.c_void_p, C.c_int, C.c_size_t]),
    "ds_sync": (C.c_int, [ctx_p]),
    "ds_timer_start": (C.c_int, [ctx_p]),
    "ds_timer_stop": (C.c_int, [ctx_p, C.POINTER(C.c_float)]),
    "ds_max_fft_len": (C.c_int, []),
    "ds_profile_enable": (C.c_int, [ctx_p, C.c_int]),
    "ds_profile_report": (C.c_char_p, [ctx_p]),
    "ds_routes": (C.c_char_p, [ctx_p]),
    "ds_profile_only": (C.c_int, [ctx_p, C.c_char_p]),
    "ds_profile_stride": (C.c_int, [ctx_p, C.c_int]),
    "ds_profile_overhead": (C.c_int, [ctx_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "ds_stft_r2c_dev": (C.c_int, [ctx_p, f32_p, i64, C.c_int, i64, C.c_int, C.c_int, C.c_int, i64,
                                  C.c_int, f32_p, C.c_int, C.c_float, C.c_float, C.c_int, c32_p]),
    "ds_stft_r2c": (C.c_int, [ctx_p, f32_p, i64, C.c_int, C.c_int, C.c_int, C.c_int, i64, C.c_int,
                              f32_p, C.c_int, C.c_float, C.c_float, C.c_int, c32_p]),
    "ds_welch_tf_dev": (C.c_int, [ctx_p, f32_p, C.c_int, i64, f32_p, C.c_int, i64, i64, C.c_int,
                                  C.c_int, C.c_int, f32_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_double, C.c_double, C.c_int, c32_p, f32_p]),
    "ds_welch_tf": (C.c_int, [ctx_p, f32_p, C.c_int, f32_p, C.c_int, i64, C.c_int, C.c_int, C.c_int,
                              f32_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                              C.c_int, c32_p, f32_p]),
    "ds_welch_tf_f64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, i64, C.c_int, C.c_int, C.c_int,
                                  f32_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                  C.c_int, c32_p, f32_p]),
    "ds_welch_psd_dev": (C.c_int, [ctx_p, f32_p, C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, f32_p,
                                   C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, f32_p]),
    "ds_welch_psd": (C.c_int, [ctx_p, f32_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, f32_p, C.c_int,
                               C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, f32_p]),
    "ds_stft_r2c_f64": (C.c_int, [ctx_p, C.c_void_p, i64, C.c_int, C.c_int, C.c_int, C.c_int, i64, C.c_int, f32_p,
                                  C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "ds_fir_ola_f64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, i64, f32_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ds_welch_psd_f64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, f32_p, C.c_int,
                                   C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, f32_p]),
    "ds_csm_f64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, f32_p, C.c_int,
                             C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, c32_p]),
    "ds_welch_csd": (C.c_int, [ctx_p, f32_p, f32_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, f32_p,
                               C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, c32_p]),
    "ds_welch_csd_f64": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, f32_p,
                                   C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, c32_p]),
    "ds_band_power_dev": (C.c_int, [ctx_p, c32_p, C.c_int, i64, f32_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, f32_p]),
    "ds_band_power": (C.c_int, [ctx_p, c32_p, C.c_int, i64, f32_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                C.c_int, f32_p]),
    "ds_das_map_dev": (C.c_int, [ctx_p, c32_p, c32_p, C.c_int, C.c_int, C.c_int, f32_p]),
    "ds_das_map": (C.c_int, [ctx_p, c32_p, c32_p, C.c_int, C.c_int, C.c_int, f32_p]),
    "ds_csm_das_prepare_dev": (C.c_int, [ctx_p, c32_p, C.c_int, C.c_int, C.c_double, C.c_int, c32_p]),
    "ds_istft_dev": (C.c_int, [ctx_p, c32_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, f32_p, C.c_float, i64, f32_p, i64]),
    "ds_istft": (C.c_int, [ctx_p, c32_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.c_int, f32_p, C.c_float, i64, f32_p]),
    "ds_csm_dev": (C.c_int, [ctx_p, f32_p, C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, f32_p,
                             C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, c32_p]),
    "ds_csm_bins_dev": (C.c_int, [ctx_p, f32_p, C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, f32_p,
                                  C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, c32_p]),
    "ds_csm": (C.c_int, [ctx_p, f32_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, f32_p, C.c_int,
                         C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, c32_p]),
    "ds_csm_spec_dev": (C.c_int, [ctx_p, c32_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                  C.c_int, c32_p]),
    "ds_csm_spec": (C.c_int, [ctx_p, c32_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                              C.c_int, c32_p]),
    "ds_rfft_dev": (C.c_int, [ctx_p, f32_p, C.c_int, i64, i64, C.c_int, C.c_float, c32_p]),
    "ds_rfft": (C.c_int, [ctx_p, f32_p, C.c_int, i64, C.c_int, C.c_float, c32_p]),
    "ds_deconv_inverse_dev": (C.c_int, [ctx_p, c32_p, C.c_int, C.c_int, f32_p, c32_p]),
    "ds_deconv_dev": (C.c_int, [ctx_p, f32_p, C.c_int, C.c_int, i64, i64, C.c_int, c32_p, C.c_int,
                                i64, i64, f32_p]),
    "ds_deconv": (C.c_int, [ctx_p, f32_p, C.c_int, C.c_int, i64, C.c_int, c32_p, C.c_int, i64, f32_p]),
    "ds_fir_ola_dev": (C.c_int, [ctx_p, f32_p, C.c_int, i64, i64, f32_p, C.c_int, C.c_int, C.c_int,
                                 f32_p, i64]),
    "ds_fir_ola": (C.c_int, [ctx_p, f32_p, C.c_int, i64, f32_p, C.c_int, C.c_int, C.c_int, f32_p]),
    "ds_welch_tf_x64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, i64, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                  C.c_void_p, C.c_void_p]),
    "ds_welch_spec_x64": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "ds_csm_x64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, i64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                             C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "ds_fir_part_step_dev": (C.c_int, [ctx_p, f32_p, f32_p, C.c_int, C.c_int, C.c_int, C.c_int, c32_p, C.c_int,
                                       C.c_int, c32_p, C.c_int, f32_p]),
    "ds_fir_ols_step_dev": (C.c_int, [ctx_p, f32_p, f32_p, C.c_int, i64, c32_p, f32_p]),
    "ds_rfft_f64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, i64, C.c_int, C.c_float, C.c_void_p]),
    "ds_deconv_f64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, i64, C.c_int, c32_p, C.c_int, i64, C.c_void_p]),
    "ds_istft_f64": (C.c_int, [ctx_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               f32_p, C.c_float, i64, C.c_void_p]),
    "ds_fir_freqz": (C.c_int, [ctx_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "ds_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ds_comm_init": (C.c_int, [ctx_p, C.c_int, C.c_int, C.c_char_p]),
    "ds_comm_count": (C.c_int, [ctx_p, C.POINTER(C.c_int)]),
    "ds_bcast": (C.c_int, [ctx_p, C.c_void_p, C.c_size_t, C.c_int]),
    "ds_allgather": (C.c_int, [ctx_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ds_comm_destroy": (C.c_int, [ctx_p]),
    "ds_measure_copy": (C.c_int, [ctx_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "ds_mem_info": (C.c_int, [ctx_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
}

_lib = None
_lock = threading.Lock()


class DeviceError(RuntimeError):
    """Raised for every non-zero return of the HIP library."""


def load_library(path: str | None = None):
    """Load libdsptoolbox_amd.so and bind every symbol of the header.  Raises if
    the library has not been built (python -m dsptoolbox_amd._build)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = path or os.environ.get("DSPTOOLBOX_AMD_LIB", LIB_PATH)
        if not os.path.exists(path):
            raise DeviceError(
                f"{path} not found: the HIP extension is required (no CPU fallback). "
                "Build it with `python -m dsptoolbox_amd._build`.")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class Context:
    """One ds_ctx: one device, one HIP stream."""

    def __init__(self, device: int | None = None):
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
            n = self.lib.ds_device_count()
            if n > 0:
                device %= n
        h = ctx_p()
        rc = self.lib.ds_init(int(device), C.byref(h))
        if rc != 0:
            msg = self.lib.ds_last_error(None)
            raise DeviceError(f"ds_init({device}) failed [{rc}]: {msg.decode() if msg else ''}")
        self.handle = h
        self.device = device

    def check(self, rc: int, what: str = ""):
        if rc == 0:
            return
        msg = self.lib.ds_last_error(self.handle)
        text = f"{what} failed [{rc}]: {msg.decode() if msg else ''}"
        if rc == -2:
            raise NotImplementedError(text)
        if rc == -1:
            raise ValueError(text)
        raise DeviceError(text)

    # ---- raw device memory -------------------------------------------------
    def last_error(self) -> str:
        msg = self.lib.ds_last_error(self.handle)
        return msg.decode() if msg else ""

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.check(self.lib.ds_malloc(self.handle, C.byref(p), nbytes), "ds_malloc")
        return p.value

    def free(self, dptr: int):
        self.check(self.lib.ds_free(self.handle, C.c_void_p(dptr)), "ds_free")

    def upload(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.ds_upload(self.handle, C.c_void_p(dptr), arr.ctypes.data, arr.nbytes),
                   "ds_upload")

    def download(self, dptr: int, arr: np.ndarray):
        assert arr.flags.c_contiguous
        self.check(self.lib.ds_download(self.handle, arr.ctypes.data, C.c_void_p(dptr), arr.nbytes),
                   "ds_download")

    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        d = self.malloc(arr.nbytes)
        self.upload(d, arr)
        return d

    def staging(self, nbytes: int) -> np.ndarray:
        """A page-locked host byte buffer of at least `nbytes`, owned by the context and reused by every call
        (grown when needed): results are downloaded into it at the link's rate and copied / widened out of it
        before the next call.  One buffer per context = per thread."""
        cur = getattr(self, "_staging", None)
        if cur is None or cur[1] < nbytes:
            if cur is not None:
                self.check(self.lib.ds_host_free(self.handle, C.c_void_p(cur[0])), "ds_host_free")
                self._staging = None
            size = max(int(nbytes), 1 << 22)
            p = C.c_void_p()
            self.check(self.lib.ds_host_alloc(self.handle, C.byref(p), size), "ds_host_alloc")
            arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(size,))
            self._staging = cur = (p.value, size, arr)
        return cur[2][:nbytes]

    def download_result(self, dptr: int, shape, dtype) -> np.ndarray:
        """Device -> a page-locked host block the CALLER owns: the array returned is backed by a block of this
        context's result pool, which takes the block back when the last array referring to it is gone (its reference
        count says so).  For the small results of device-resident calls: no staging copy, no page faults of a fresh
        allocation, the link's full rate."""
        import sys
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        size = max(4096, 1 << (n - 1).bit_length())
        pool = self.__dict__.setdefault("_result_pool", {})
        lst = pool.setdefault(size, [])
        if len(lst) > 64:
            # many results of this size were alive at once (a caller keeping hundreds of spectra): of the blocks whose
            # arrays are gone by now all but 16 go back to the driver instead of staying page-locked for good
            keep, spare = [], 0
            for q, b in lst:
                if sys.getrefcount(b) <= 3:  # the tuple, the loop variable, getrefcount's argument
                    spare += 1
                    if spare > 16:
                        self.lib.ds_host_free(self.handle, C.c_void_p(q))
                        continue
                keep.append((q, b))
            del q, b
            lst[:] = keep
        blk = None
        for cand in lst:
            if sys.getrefcount(cand[1]) <= 2:  # the tuple in the list and getrefcount's argument
                blk = cand[1]
                break
        if blk is None:
            p = C.c_void_p()
            self.check(self.lib.ds_host_alloc(self.handle, C.byref(p), size), "ds_host_alloc")
            blk = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(size,))
            lst.append((p.value, blk))
        self.check(self.lib.ds_download(self.handle, blk.ctypes.data, C.c_void_p(dptr), n), "ds_download")
        return blk[:n].view(dtype).reshape(shape)

    def download_staged(self, dptr: int, shape, dtype) -> np.ndarray:
        """Device -> page-locked staging buffer; the returned array is a VIEW of that buffer (valid until the
        next staged download of this context)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        view = self.staging(n)
        self.check(self.lib.ds_download(self.handle, view.ctypes.data, C.c_void_p(dptr), n), "ds_download")
        return view.view(dtype).reshape(shape)

    def sync(self):
        self.check(self.lib.ds_sync(self.handle), "ds_sync")

    def timer_start(self):
        self.check(self.lib.ds_timer_start(self.handle), "ds_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self.check(self.lib.ds_timer_stop(self.handle, C.byref(ms)), "ds_timer_stop")
        return float(ms.value)

    def profile_enable(self, on: bool = True):
        self.check(self.lib.ds_profile_enable(self.handle, int(on)), "ds_profile_enable")

    def profile_only(self, kernel_name: str | None):
        """Bracket only launches of this kernel name (None: all kernels)."""
        self.check(self.lib.ds_profile_only(self.handle, kernel_name.encode() if kernel_name else None),
                   "ds_profile_only")

    def profile_stride(self, every: int):
        """Bracket only every `every`-th matching launch (1: all)."""
        self.check(self.lib.ds_profile_stride(self.handle, int(every)), "ds_profile_stride")

    def profile_overhead(self, reps: int = 100) -> float:
        """Lower bound (ms) of what an event pair adds to the kernel it brackets: with b1 / b2 the
        brackets of one / two empty kernels, 2 b1 - b2 (b2 - b1 is an empty kernel's cost in the stream)."""
        b = []
        for n in (1, 2):
            ms = C.c_double(0.0)
            self.check(self.lib.ds_profile_overhead(self.handle, int(reps), n, C.byref(ms)), "ds_profile_overhead")
            b.append(ms.value)
        return max(0.0, 2.0 * b[0] - b[1])

    def profile_report(self) -> dict:
        """{kernel: (total_ms, launches)} since the previous report."""
        text = self.lib.ds_profile_report(self.handle).decode()
        out = {}
        for line in text.splitlines():
            name, ms, cnt = line.split()
            out[name] = (float(ms), int(cnt))
        return out

    def routes(self) -> set:
        """Launch names since the previous call: "group" or "group@variant" (which kernel family ran)."""
        return set(self.lib.ds_routes(self.handle).decode().split())

    def close(self):
        if getattr(self, "handle", None):
            import sys
            cur = getattr(self, "_staging", None)
            if cur is not None:
                self.lib.ds_host_free(self.handle, C.c_void_p(cur[0]))
                self._staging = None
            for lst in self.__dict__.pop("_result_pool", {}).values():
                for ptr, blk in lst:
                    if sys.getrefcount(blk) <= 3:  # (a block somebody still reads stays mapped until the process ends)
                        self.lib.ds_host_free(self.handle, C.c_void_p(ptr))
            self.lib.ds_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()


def get_context() -> Context:
    """Default context of the CALLING THREAD (device = LOCAL_RANK, one process per GPU).
    A ds_ctx is bound to one stream and one scratch workspace and is not re-entrant
    (INTEGRATION.md section 5), while ctypes releases the GIL during a call -- so every Python
    thread gets its own context, like the thread-safe numpy reference it stands in for."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        ctx = _tls.ctx = Context()
    return ctx


def reset_context() -> Context:
    """Close the calling thread's default context and open a new one.  A context reads the
    DSPTOOLBOX_AMD_* switches ONCE, in ds_init (csrc/config.hpp): this is how a process
    changes them afterwards (tests, A/B timing tools).  Device buffers of the old context
    must not be used after the call."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is not None:
        ctx.close()
    _tls.ctx = Context()
    return _tls.ctx


class DevicePlanar:
    """Planar float32 samples that live in HBM: channel c at `ptr + 4 * c * ld`, `n_samples` valid samples each
    (the layout every `_dev` entry point of the C-ABI takes).  `owner` is the DeviceBuffer the bytes belong to --
    several DevicePlanar objects may share one (the bands of a filter bank's output are slices of ONE buffer) and it
    is freed when the last of them goes.  Treated as immutable: nothing writes into a buffer once it is handed out."""

    def __init__(self, owner: "DeviceBuffer", n_ch: int, n_samples: int, ld: int | None = None, offset_bytes: int = 0):
        self.owner = owner
        self.n_ch, self.n_samples = int(n_ch), int(n_samples)
        self.ld = int(n_samples if ld is None else ld)
        self.offset_bytes = int(offset_bytes)
        assert self.offset_bytes + 4 * ((self.n_ch - 1) * self.ld + self.n_samples) <= owner.nbytes

    @property
    def ctx(self):
        return self.owner.ctx

    @property
    def ptr(self) -> int:
        return self.owner.ptr + self.offset_bytes

    @classmethod
    def from_planar(cls, ctx: "Context", planar: np.ndarray) -> "DevicePlanar":
        """Upload a (channels, samples) float32 array (one copy over the link, no cast)."""
        planar = np.ascontiguousarray(planar, dtype=np.float32)
        assert planar.ndim == 2, "planar samples are (channels, samples)"
        return cls(DeviceBuffer.from_array(ctx, planar), planar.shape[0], planar.shape[1])

    def to_planar(self) -> np.ndarray:
        """(channels, samples) float32 on the host."""
        out = np.empty((self.n_ch, self.n_samples), dtype=np.float32)
        if self.ld == self.n_samples:
            self.ctx.download(self.ptr, out)
        else:
            for c in range(self.n_ch):
                self.ctx.download(self.ptr + 4 * c * self.ld, out[c])
        return out

    def __deepcopy__(self, memo):
        return self  # immutable and reference counted: copies of a Signal share the device samples


class DeviceBuffer:
    """Owned device allocation (freed explicitly or at garbage collection)."""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.ptr = ctx.malloc(self.nbytes)

    @classmethod
    def from_array(cls, ctx: Context, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        buf = cls(ctx, arr.nbytes)
        ctx.upload(buf.ptr, arr)
        return buf

    def to_array(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.download(self.ptr, out)
        return out

    def free(self):
        if self.ptr is not None and self.ctx.handle:
            self.ctx.free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
