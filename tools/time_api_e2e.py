"""Dev tool: end-to-end time of the Python API for config 2 (host float64 (N, C) arrays in,
float64/complex128 out) split into host preparation, the C-ABI call (H2D + kernels + D2H) and the
rest."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsptoolbox_amd as dsp  # noqa: E402
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd.generators import sweep_and_responses  # noqa: E402

x, y = sweep_and_responses(2**20, 64, 48000)
X = dsp.Signal(None, x, 48000)
Y = dsp.Signal(None, y, 48000)
X.set_spectrum_parameters(window_length_samples=4096, overlap_percent=50, detrend=True)
for it in range(3):
    t0 = time.perf_counter()
    H = dsp.transfer_functions.compute_transfer_function(Y, X, 4096, dsp.TransferFunctionType.H1)
    t1 = time.perf_counter()
    print(f"compute_transfer_function end to end: {(t1 - t0) * 1e3:8.1f} ms")
t0 = time.perf_counter()
yp = backend._planar_f32(y)
t1 = time.perf_counter()
print(f"  host (N, C) float64 -> planar float32: {(t1 - t0) * 1e3:8.1f} ms for {y.nbytes / 1e6:.0f} MB")

# STFT of the CSM shape and a 4-band FIR bank, float64 in and out
rng = np.random.default_rng(1)
mic = rng.standard_normal((512000, 64)) * 0.1
S = dsp.Signal(None, mic, 48000)
for it in range(3):
    S = dsp.Signal(None, mic, 48000)
    t0 = time.perf_counter()
    t, f, st = S.get_spectrogram()
    t1 = time.perf_counter()
    print(f"get_spectrogram 64 ch x 512 000 (-> {st.nbytes / 1e6:.0f} MB complex128): {(t1 - t0) * 1e3:8.1f} ms")
sig = rng.standard_normal((2**22, 8)) * 0.1
taps = [rng.standard_normal(4097) * 0.01 for _ in range(4)]
for it in range(3):
    t0 = time.perf_counter()
    out = backend.fir_filter_bank(sig, taps, backend.DS_FB_PARALLEL)
    t1 = time.perf_counter()
    print(f"fir_filter_bank 4 x 4097 taps, 8 ch x 2^22 (-> {out.nbytes / 1e6:.0f} MB float64): {(t1 - t0) * 1e3:8.1f} ms")

# round 4: whole-signal spectra, one-item deconvolution and the inverse STFT through the pinned pipelines
# (ds_rfft_f64 / ds_deconv_f64 / ds_istft_f64) against the host-cast path (a Fortran-ordered view is not "fusable")
big = rng.standard_normal((2**21, 8)) * 0.1
for name, arr in (("pinned pipeline", big), ("host cast", np.asfortranarray(big))):
    for it in range(2):
        t0 = time.perf_counter()
        sp = backend.rfft_spectrum(arr, 2**21)
        t1 = time.perf_counter()
    print(f"rfft_spectrum 8 ch x 2^21 (-> {sp.nbytes / 1e6:.0f} MB complex128), {name}: {(t1 - t0) * 1e3:8.1f} ms")
inv = (rng.standard_normal(2**20 + 1) + 1j * rng.standard_normal(2**20 + 1)).astype(np.complex64)
for name, arr in (("pinned pipeline", big), ("host cast", np.asfortranarray(big))):
    for it in range(2):
        t0 = time.perf_counter()
        ir = backend.spectral_division(arr, 2**21, inv, 2**21)
        t1 = time.perf_counter()
    print(f"spectral_division 8 ch x 2^21 (-> {ir.nbytes / 1e6:.0f} MB float64), {name}: {(t1 - t0) * 1e3:8.1f} ms")
S = dsp.Signal(None, mic, 48000)
t, f, st = S.get_spectrogram()
for name, arr in (("pinned pipeline", st), ("host cast", np.asfortranarray(st))):
    for it in range(2):
        t0 = time.perf_counter()
        back = dsp.transforms.istft(arr, original_signal=S)
        t1 = time.perf_counter()
    print(f"istft of the {st.nbytes / 1e6:.0f} MB spectrogram (-> {back.time_data.nbytes / 1e6:.0f} MB float64), {name}: {(t1 - t0) * 1e3:8.1f} ms")
