"""Round 5: the reference-shaped API over DEVICE-RESIDENT signals (Signal.to_device / from_planar_f32) against the same
calls over host arrays.  Wall time per call, host-side clock around the call (results that are small arrays are on the host
when the call returns; results that are signals stay in HBM, the call returns when the device is done).
    python tools/time_api_resident.py > gpurun_out/r05_api_resident.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsptoolbox_amd as dsp  # noqa: E402
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd._lib import get_context  # noqa: E402
from dsptoolbox_amd.generators import fir_bank_taps, sweep_and_responses  # noqa: E402
from dsptoolbox_amd.standard.enums import FilterBankMode  # noqa: E402

ctx = get_context()


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ctx.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


fs = 48000
x, y = sweep_and_responses(2**20, 64, fs)
X, Y = dsp.Signal(None, x, fs), dsp.Signal(None, y, fs)
X.set_spectrum_parameters(window_length_samples=4096, overlap_percent=50, detrend=True)
H1 = dsp.TransferFunctionType.H1
t0 = time.perf_counter()
Xd = dsp.Signal.from_planar_f32(backend._planar_f32(x), fs)
Yd = dsp.Signal(None, y, fs).to_device()
ctx.sync()
print(f"one-off: 65 channels x 2^20 float64 -> planar float32 -> HBM: {(time.perf_counter() - t0) * 1e3:.1f} ms")
Xd.set_spectrum_parameters(window_length_samples=4096, overlap_percent=50, detrend=True)
for W in (4096, 1024):
    for s in (X, Xd):
        s.set_spectrum_parameters(window_length_samples=W, overlap_percent=50, detrend=True)
    m_h = timed(lambda: dsp.transfer_functions.compute_transfer_function(Y, X, W, H1), reps=5, warm=1)
    m_d = timed(lambda: dsp.transfer_functions.compute_transfer_function(Yd, Xd, W, H1), reps=200, warm=20)
    print(f"compute_transfer_function 64 + 1 ch x 2^20, window {W}:  host arrays {m_h[0]:8.3f} ms   resident {m_d[0]:8.3f} ms (min {m_d[1]:.3f})")
assert not Xd._has_host_copy  # (Yd came from a host array and keeps it)

rng = np.random.default_rng(1)
mic = rng.standard_normal((512000, 64)) * 0.1
S, Sd = dsp.Signal(None, mic, fs), dsp.Signal.from_planar_f32(backend._planar_f32(mic), fs)
m_h = timed(lambda: S.get_spectrogram(force_computation=True), reps=3, warm=1)
m_d = timed(lambda: Sd.get_spectrogram(force_computation=True), reps=3, warm=1)
m_k = timed(lambda: Sd.get_spectrogram(on_device=True), reps=50, warm=5)
print(f"get_spectrogram 64 ch x 512 000, window 1024:  host arrays {m_h[0]:8.2f} ms   resident in, array out {m_d[0]:8.2f} ms   resident in and out {m_k[0]:8.3f} ms")
_, _, handle = Sd.get_spectrogram(on_device=True)
_, _, st = S.get_spectrogram()
m_h = timed(lambda: dsp.transforms.istft(st, original_signal=S), reps=3, warm=1)
m_k = timed(lambda: dsp.transforms.istft(handle, original_signal=Sd), reps=50, warm=5)
print(f"istft of that spectrogram:  host arrays {m_h[0]:8.2f} ms   resident in and out {m_k[0]:8.3f} ms")
m_h = timed(lambda: S.get_csm(force_computation=True), reps=3, warm=1)
m_k = timed(lambda: Sd.get_csm(on_device=True), reps=50, warm=5)
print(f"get_csm 64 mics, 1000 frames:  host arrays {m_h[0]:8.2f} ms   resident in and out {m_k[0]:8.3f} ms")

sig = rng.standard_normal((2**22, 8)) * 0.1
G, Gd = dsp.Signal(None, sig, fs), dsp.Signal.from_planar_f32(backend._planar_f32(sig), fs)
for K in (4, 32):
    taps = fir_bank_taps(K, 4097, fs)
    fb = dsp.FilterBank([dsp.Filter.from_ba(t, [1.0], fs) for t in taps])
    m_k = timed(lambda: fb.filter_signal(Gd, FilterBankMode.Parallel), reps=10, warm=2)
    line = f"FilterBank.filter_signal Parallel, {K} x 4097 taps, 8 ch x 2^22 ({K * sig.size * 4 / 1e9:.2f} GB of float32 bands):  resident {m_k[0]:8.3f} ms"
    if K == 4:
        m_h = timed(lambda: fb.filter_signal(G, FilterBankMode.Parallel), reps=2, warm=1)
        line += f"   host arrays {m_h[0]:8.1f} ms"
    else:
        out = fb.filter_signal(Gd, FilterBankMode.Parallel)
        t0 = time.perf_counter()
        one = out.bands[5].time_data
        line += f"   (nothing downloaded; one band on demand: {(time.perf_counter() - t0) * 1e3:.1f} ms for {one.nbytes / 1e6:.0f} MB float64)"
        del out
    print(line)
f1 = dsp.Filter.from_ba(fir_bank_taps(1, 4097, fs)[0], [1.0], fs)
m_h = timed(lambda: f1.filter_signal(G), reps=2, warm=1)
m_k = timed(lambda: f1.filter_signal(Gd), reps=20, warm=3)
print(f"Filter.filter_signal 4097 taps, 8 ch x 2^22:  host arrays {m_h[0]:8.1f} ms   resident {m_k[0]:8.3f} ms")

n = 2**17
xs, ys = sweep_and_responses(n, 2, fs)
A, B = dsp.Signal(None, ys, fs), dsp.Signal(None, xs, fs)
Ad, Bd = dsp.Signal.from_planar_f32(backend._planar_f32(ys), fs), dsp.Signal.from_planar_f32(backend._planar_f32(xs), fs)
m_h = timed(lambda: dsp.transfer_functions.spectral_deconvolve(A, B), reps=5, warm=1)
m_k = timed(lambda: dsp.transfer_functions.spectral_deconvolve(Ad, Bd), reps=20, warm=3)
print(f"spectral_deconvolve 2 ch x 2^17:  host arrays {m_h[0]:8.2f} ms   resident {m_k[0]:8.3f} ms")
