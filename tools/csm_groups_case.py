"""Dev tool: the one case of tests/sweeps/fuzz_csm.py (60 cases, seed 12) that read 0.11 on the channel-group kernels --
window 2048, 130 channels, 400000 samples, no overlap -- replayed from the sweep's own random stream (the draws of the
earlier cases are made, their GPU work is not), then taken apart: error by 64-channel block and by bin, and the same data
through the generic kernel (DSPTOOLBOX_AMD_CSM_GENERIC=1 in the environment) for comparison."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd._lib import get_context  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
ctx = get_context()
rng = np.random.default_rng(12)
scalings = list(SpectrumScaling)
for it in range(60):
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
    kind = int(rng.integers(0, 2))
    if kind:
        x = level * (0.3 * rng.standard_normal((n, C)) + 0.5 * rng.standard_normal(n)[:, None])
    else:
        src = rng.standard_normal(n) * 0.3 + 0.05
        h = rng.standard_normal((32, C)) * np.exp(-np.arange(32) / 6.0)[:, None]
        noise = 0.05 * rng.standard_normal((n, C))
        if (W, C, n) == (2048, 130, 400000):
            x = level * (np.stack([np.convolve(src, h[:, c])[:n] for c in range(C)], axis=1) + noise)
    hit = (W, C, n) == (2048, 130, 400000)
    if hit:
        print("case", it, (W, C, n, ov, det, sc.name, level), "source kind", kind, "frames", int(np.ceil(n / hop)))
        ctx.routes()
        f, csm = backend._csm_welch(x, 48000, W, Window.Hann, ov, det, "mean", sc)
        print("routes", sorted(ctx.routes()))
        fr, ref = orc.csm_welch_batched(x, 48000, W, "hann", ov, det, sc.name)
        err = np.abs(csm - ref) / np.abs(ref).max()
        ng = (C + 63) // 64
        print("max", err.max(), "blocks", {(i, j): f"{err[:, 64 * i:64 * i + 64, 64 * j:64 * j + 64].max():.1e}" for i in range(ng) for j in range(ng)})
        per_bin = err.reshape(err.shape[0], -1).max(axis=1)
        bad = np.nonzero(per_bin > 1e-6)[0]
        print("bad bins", len(bad), bad[:12], bad[-6:])
        print("|ref| max per bin (first 6, last 3)", np.abs(ref).reshape(ref.shape[0], -1).max(axis=1)[[0, 1, 2, 3, 4, 5, -3, -2, -1]])
        for b in bad[:3]:
            ij = np.argwhere(err[b] > 1e-6)
            i, j = ij[0]
            print("  bin", b, "bad elements", len(ij), "rows", sorted(set(ij[:, 0]))[:8], "cols", sorted(set(ij[:, 1]))[:8],
                  "e.g.", (i, j), csm[b, i, j], ref[b, i, j])
        part = backend._csm_welch_bins(x, 48000, W, Window.Hann, ov, det, sc, 0, 8)
        print("bins 0..7 as a bin range:", float(np.abs(part - ref[:8]).max() / np.abs(ref).max()))
        break
    # the sweep's remaining draws of this case
    if rng.integers(0, 3) == 0 and W >= 128:
        a = int(rng.integers(0, W // 2))
        b = int(rng.integers(a + 1, W // 2 + 2))
