"""Dev tool: where does the fp32 Welch estimate lose the coherence of a bin the output barely excites?
tests/sweeps/fuzz_api3.py found 2.2e-6 in the coherence at the Nyquist bin (4096-sample window, 216 frames, one input
channel per output channel, |H| there 34 x below its maximum).  For 8-tap responses with a prescribed value g at the
Nyquist bin: the error of tf and coherence at that bin and over all bins, against the frame count (does it average
down?) and against the kernel family (environment switches are read at ds_init: run once per switch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd import backend  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fs = 48000
rng = np.random.default_rng(5)
print("switches:", {k: v for k, v in os.environ.items() if k.startswith("DSPTOOLBOX_AMD")})
for paired in (True, False):
    for n_frames in (216, 864):
        n = (n_frames + 1) * (W // 2) - 7
        for g in (1.0, 0.1, 0.03, 0.01):
            C = 4
            x = rng.standard_normal((n, C if paired else 1)) * 0.4
            h = rng.standard_normal((8, C)) * 0.5 + 1.0
            sign = (-1.0) ** np.arange(8)
            h[7] -= (sign @ h - g) / sign[7]  # the response at the Nyquist bin is g
            y = np.stack([np.convolve(x[:, j if paired else 0], h[:, j])[:n] for j in range(C)], axis=1)
            y += 0.01 * rng.standard_normal((n, C))
            tf, coh = backend.welch_transfer_function(y, x, fs, W, "H1")
            rt, rc = orc.compute_transfer_function(y, x, fs, W, "H1")
            et, ec = np.abs(tf - rt), np.abs(coh - rc)
            b = W // 2
            print(f"paired={int(paired)} frames={n_frames:4d} g={g:5.2f}: |tf| max {np.abs(rt[1:]).max():.2f} nyq {np.abs(rt[b]).max():.3f}  "
                  f"tf err nyq {et[b].max():.2e} dc+1 {et[1].max():.2e} median bin {np.median(et[1:b].max(axis=1)):.2e} max {et[1:b].max():.2e} | "
                  f"coh nyq {rc[b].min():.3f} err nyq {ec[b].max():.2e} median {np.median(ec[1:b].max(axis=1)):.2e} max {ec[1:b].max():.2e} "
                  f"at bin {np.unravel_index(np.argmax(ec[1:b]), ec[1:b].shape)[0] + 1}")
