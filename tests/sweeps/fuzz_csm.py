"""Dev tool: randomized Welch cross-spectral matrices (channel counts 2 ... 70, even and odd; 8 ... 300
frames; windows 64 ... 2048; every scaling; bin ranges) against the oracle -- the bf16-triple
kernel (up to 64 channels; odd counts read one value past the row into a tile row that is not stored)
and the channel-group kernels (> 64); DSPTOOLBOX_AMD_CSM_F32=1 puts the fp32-matrix-instruction kernel in its place."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
scalings = list(SpectrumScaling)
worst, fails = {}, []
for it in range(n_cases):
    W = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
    C = int(rng.choice([2, 3, 4, 6, 8, 16, 30, 32, 33, 34, 40, 62, 63, 64, 70, 97, 130]))
    F = int(rng.integers(8, 300))
    ov = float(rng.choice([0.0, 50.0, 75.0]))
    hop = max(1, int(W * (1 - ov / 100)))
    n = hop * (F - 1) + W + int(rng.integers(0, hop))
    n = min(n, 400000)
    det = bool(rng.integers(0, 2))
    sc = scalings[int(rng.integers(0, len(scalings)))]
    level = float(10.0 ** rng.uniform(-4, 2))
    if rng.integers(0, 2):
        x = level * (0.3 * rng.standard_normal((n, C)) + 0.5 * rng.standard_normal(n)[:, None])
    else:  # one source through responses of either sign: negative real cross spectra at DC / Nyquist
        src = rng.standard_normal(n) * 0.3 + 0.05
        h = rng.standard_normal((32, C)) * np.exp(-np.arange(32) / 6.0)[:, None]
        x = level * (np.stack([np.convolve(src, h[:, c])[:n] for c in range(C)], axis=1) + 0.05 * rng.standard_normal((n, C)))
    info = (W, C, n, ov, det, sc.name, f"{level:.1e}")
    try:
        f, csm = backend._csm_welch(x, 48000, W, Window.Hann, ov, det, "mean", sc)
        fr, ref = orc.csm_welch_batched(x, 48000, W, "hann", ov, det, sc.name)
        lo = 1 if det else 0
        e = orc.rel_max(csm[lo:], ref[lo:])
        kind = ("b3" if C % 2 == 0 else "b3 odd") if C <= 64 else "groups"
        if rng.integers(0, 3) == 0 and W >= 128:
            a = int(rng.integers(0, W // 2))
            b = int(rng.integers(a + 1, W // 2 + 2))
            part = backend._csm_welch_bins(x, 48000, W, Window.Hann, ov, det, sc, a, b)
            l2 = 1 if (det and a == 0) else 0
            if b - a > l2:
                e = max(e, orc.rel_max(part[l2:], ref[a + l2:b]))
            kind += "+range"
        worst[kind] = max(worst.get(kind, 0.0), e)
        if not np.isfinite(e) or e > 1e-6:
            fails.append((info, kind, e))
    except Exception as ex:  # noqa: BLE001
        fails.append((info, "exception", repr(ex)[:200]))
print("worst", worst, "failures", len(fails))
for f_ in fails[:20]:
    print("  ", f_)
