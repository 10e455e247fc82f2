"""Edge shapes of the Welch register kernels: one / two / three frames, one channel, paired inputs, cross spectra."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
from dsptoolbox_amd import backend
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window
from oracle import dsp_oracle as orc
warnings.simplefilter("ignore")
rng = np.random.default_rng(5)
worst = 0.0
def rel(a, b, lo):
    a, b = np.asarray(a)[lo:], np.asarray(b)[lo:]
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
for W in (32, 64, 128, 256, 1024, 2048, 4096, 8192, 16384):
    for n in (W, W + 1, W + W // 2, 2 * W, 2 * W + 5, 3 * W - 1, 5 * W + 17):
        for C in (1, 2, 3):
            for det in (False, True):
                x = rng.standard_normal((n, C)) * 0.3
                y = np.stack([np.convolve(x[:, i], rng.standard_normal(5))[:n] for i in range(C)], axis=1) + 0.01 * rng.standard_normal((n, C))
                lo = 1 if det else 0
                a = backend._welch(y, None, 48000, Window.Hann, W, 50, det, "mean", SpectrumScaling.FFTBackward)
                r = orc.welch(y, None, 48000, "hann", W, 50, det, "mean", "FFTBackward")
                e = [rel(a, r, lo)]
                k = backend._welch(x, y, 48000, Window.Hann, W, 50, det, "mean", SpectrumScaling.FFTBackward)
                r = orc.welch(x, y, 48000, "hann", W, 50, det, "mean", "FFTBackward")
                e.append(rel(k, r, lo))
                for xin in (x, x[:, :1]):
                    yy = y if xin.shape[1] == C else np.stack([np.convolve(x[:, 0], rng.standard_normal(5))[:n] for _ in range(C)], axis=1)
                    tf, coh = backend.welch_transfer_function(yy, xin, 48000, W, "H1", detrend=det, precision="f32")
                    rt, rc = orc.compute_transfer_function(yy, xin, 48000, W, "H1", detrend=det)
                    e.append(rel(tf, rt, lo)); e.append(rel(coh, rc, lo))
                m = max(e)
                worst = max(worst, m)
                if not np.isfinite(m) or m > 2e-5:
                    print("BAD", W, n, C, det, [f"{v:.2e}" for v in e])
print("worst", worst)
