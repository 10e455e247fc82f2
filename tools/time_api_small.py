"""Dev tool: latency of the reference-shaped Python API on a typical small input (192 000-sample
stereo signal, the size of the reference's example chirps): what a drop-in user sees per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsptoolbox_amd as dsp  # noqa: E402

rng = np.random.default_rng(0)
n, fs = 192000, 48000
x = rng.standard_normal((n, 1)) * 0.1
y = np.stack([np.convolve(x[:, 0], rng.standard_normal(64))[:n] for _ in range(2)], axis=1)


def bench(name, fn, reps=20):
    fn()
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    print(f"{name:44s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms")


def spectrum():
    s = dsp.Signal(None, y, fs)
    return s.get_spectrum()


def spectrogram():
    s = dsp.Signal(None, y, fs)
    return s.get_spectrogram()


def csm():
    s = dsp.Signal(None, y, fs)
    return s.get_csm()


def tf():
    return dsp.transfer_functions.compute_transfer_function(dsp.Signal(None, y, fs), dsp.Signal(None, x, fs), 4096,
                                                            dsp.TransferFunctionType.H1)


def deconv():
    return dsp.transfer_functions.spectral_deconvolve(dsp.Signal(None, y, fs), dsp.Signal(None, x, fs))


fir = dsp.Filter.fir_filter(1024, 1000.0, dsp.FilterPassType.Lowpass, fs)


def filt():
    return fir.filter_signal(dsp.Signal(None, y, fs))


bench("Signal(...) construction only", lambda: dsp.Signal(None, y, fs))
bench("get_spectrum (Welch 1024)", spectrum)
bench("get_spectrogram (1024)", spectrogram)
bench("get_csm (Welch 1024)", csm)
bench("compute_transfer_function (4096, H1)", tf)
bench("spectral_deconvolve (192 000, Bluestein)", deconv)
bench("Filter.filter_signal (1025 taps)", filt)
