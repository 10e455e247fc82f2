"""Seeded synthetic inputs for tests and benchmarks; own
formulas, fp64 on the host."""

import numpy as np
from scipy.signal import firwin


def exponential_sweep(n_samples: int, fs_hz: int, f0: float = 20.0, f1: float = 20000.0,
                      peak: float = 0.5) -> np.ndarray:
    """Exponential sine sweep f0 -> f1 over the whole length, no fades."""
    t = np.arange(n_samples) / fs_hz
    dur = n_samples / fs_hz
    k = np.log(f1 / f0)
    return peak * np.sin(2 * np.pi * f0 * dur / k * (np.exp(t / dur * k) - 1.0))


def sweep_and_responses(n_samples: int, n_channels: int, fs_hz: int = 48000,
                        noise: float = 1e-3):
    """x (N, 1) sweep; y[:, c] = (x * h_c)[:N] + noise, h_c a 256-tap decaying
    Gaussian-noise response (seed 1000 + c), noise seed 2000 + c."""
    x = exponential_sweep(n_samples, fs_hz)
    y = np.empty((n_samples, n_channels))
    decay = np.exp(-np.arange(256) / 40.0)
    nfft = 1 << int(np.ceil(np.log2(n_samples + 256)))
    X = np.fft.rfft(x, nfft)
    for c in range(n_channels):
        h = np.random.default_rng(1000 + c).standard_normal(256) * decay
        yc = np.fft.irfft(X * np.fft.rfft(h, nfft), nfft)[:n_samples]
        y[:, c] = yc + noise * np.random.default_rng(2000 + c).standard_normal(n_samples)
    return x[:, None], y


def fir_bank_taps(n_bands: int, n_taps: int, fs_hz: int = 48000, f_lo: float = 50.0,
                  f_hi: float = 20000.0) -> np.ndarray:
    """n_bands linear-phase (type I) band-pass filters, log-spaced edges. (K, T)."""
    edges = np.geomspace(f_lo, f_hi, n_bands + 1)
    return np.stack([firwin(n_taps, [edges[k], edges[k + 1]], window="hamming",
                            pass_zero="bandpass", fs=fs_hz) for k in range(n_bands)])


def mic_array_noise(n_samples: int, n_mics: int, seed: int = 4) -> np.ndarray:
    """0.1 * N(0,1) per microphone + 0.2 * common N(0,1)."""
    rng = np.random.default_rng(seed)
    return 0.1 * rng.standard_normal((n_samples, n_mics)) + 0.2 * rng.standard_normal(n_samples)[:, None]
