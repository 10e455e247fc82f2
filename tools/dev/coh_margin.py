"""Dev tool (GPU): where does the coherence error of the long-window register kernels sit?
Re-draws the class of case that failed in gpurun_out/sweeps_r02.log (W = 8192 / 16384, ~50 frames,
overlaps other than 50 %) and prints, per case, the worst coherence / tf error, its bin, channel,
the reference coherence there -- and the same error for a float32 numpy restatement of the estimate
(the yardstick of "plain fp32 arithmetic").  usage: python tools/dev/coh_margin.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def f32_estimate(y, x, W, ov, det):
    """coherence of the raw frame sums with float32 transforms and float32 accumulation"""
    hop = W - int(ov / 100 * W)
    n = x.shape[0]
    F = -(-n // hop)
    pad = W - (n % hop)
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(W) / W)).astype(np.float32)

    def spec(a):
        a = np.concatenate([a.astype(np.float32), np.zeros((pad,) + a.shape[1:], np.float32)])
        fr = np.stack([a[k * hop:k * hop + W] for k in range(F)], 0) * w[None, :, None]
        if det:
            fr = fr - fr.mean(axis=1, keepdims=True)
        return np.fft.rfft(fr.astype(np.float32), axis=1).astype(np.complex64)

    X, Y = spec(x), spec(y)
    sxx = np.sum((X.real**2 + X.imag**2).astype(np.float32), axis=0, dtype=np.float32)
    syy = np.sum((Y.real**2 + Y.imag**2).astype(np.float32), axis=0, dtype=np.float32)
    sxy = np.sum((np.conj(X) * Y).astype(np.complex64), axis=0, dtype=np.complex64)
    return (np.abs(sxy.astype(np.complex128))**2 / sxx.astype(np.float64) / syy.astype(np.float64))


for it in range(n_cases):
    W = int(rng.choice([8192, 8192, 16384, 4096]))
    ov = float(rng.choice([25, 50, 75]))
    hop = W - int(ov / 100 * W)
    frames = int(rng.integers(45, 70))
    n = frames * hop + int(rng.integers(-hop + 1, hop))
    n_ch = int(rng.choice([8, 16, 33]))
    det = bool(rng.integers(0, 2))
    sc = list(SpectrumScaling)[int(rng.integers(0, len(list(SpectrumScaling))))]
    mode = str(rng.choice(["H1", "H3"]))
    x = rng.standard_normal((n, 1)) * 0.3 + 0.05
    h = rng.standard_normal((32, n_ch)) * np.exp(-np.arange(32) / 6.0)[:, None]
    y = np.stack([np.convolve(x[:, 0], h[:, c])[:n] for c in range(n_ch)], axis=1) + 0.05 * rng.standard_normal((n, n_ch))
    tf, coh = backend.welch_transfer_function(y, x, 48000, W, mode, overlap_percent=ov, detrend=det, scaling=sc,
                                              precision="f32")
    rt, rc = orc.compute_transfer_function_batched(y, x, 48000, W, mode, overlap_percent=ov, detrend=det, scaling=sc.name)
    lo = 1 if det else 0
    dc = np.abs(coh - rc)[lo:]
    b, c = np.unravel_index(np.argmax(dc), dc.shape)
    dt = np.abs(tf - rt)[lo:]
    bt, ct = np.unravel_index(np.argmax(dt), dt.shape)
    c32 = f32_estimate(y, x, W, ov, det)
    if sc.is_amplitude_scaling():
        c32 = np.sqrt(c32)
    d32 = np.abs(c32 - rc)[lo:]
    b32, c32c = np.unravel_index(np.argmax(d32), d32.shape)
    print(f"W={W} ov={ov:.0f} frames={frames} ch={n_ch} det={det} {sc.name} {mode}: coh err {dc.max() / rc[lo:].max():.2e} at bin {b + lo} ch {c} "
          f"(coh_ref {rc[b + lo, c]:.4f}); tf err {dt.max() / np.abs(rt[lo:]).max():.2e} at bin {bt + lo} (|ref| / max "
          f"{abs(rt[bt + lo, ct]) / np.abs(rt[lo:]).max():.3f}); numpy-f32 coh err {d32.max() / rc[lo:].max():.2e} "
          f"(coh_ref {rc[b32 + lo, c32c]:.4f}); rms coh err ours {np.sqrt(np.mean(dc**2)):.2e} numpy-f32 {np.sqrt(np.mean(d32**2)):.2e}",
          flush=True)
