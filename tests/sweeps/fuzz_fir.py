"""Dev tool: randomized FIR bank shapes (tap counts across every block-size regime, signal lengths
around block boundaries, channel counts, bank modes) against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dsptoolbox_amd import backend  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, fails = 0.0, []
for it in range(n_cases):
    T = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 1100), rng.integers(1024, 2100),
                        rng.integers(2049, 4200), rng.integers(4097, 8300), 1025, 2049, 4097, 8193, 3001]))
    k = int(rng.integers(1, 4))
    c = int(rng.integers(1, 6))
    L = 16384 - (T - 1) if T > 1024 else 4096
    n = int(rng.choice([rng.integers(1, 2000), rng.integers(2000, 60000), L, L + 1, 2 * L, 3 * L - 1, rng.integers(60000, 150000)]))
    mode, name = [(backend.DS_FB_PARALLEL, "Parallel"), (backend.DS_FB_SUMMED, "Summed"),
                  (backend.DS_FB_SEQUENTIAL, "Sequential")][int(rng.integers(0, 3))]
    x = rng.standard_normal((n, c)) * 0.2
    taps = [rng.standard_normal(T) * np.exp(-np.arange(T) / max(2.0, T / 4.0)) * 0.3 for _ in range(k)]
    try:
        y = backend.fir_filter_bank(x, taps, mode)
        r = orc.filterbank_fir(taps, x, name)
        if name == "Parallel":
            r = np.transpose(r, (2, 0, 1))
        e = float(np.max(np.abs(y - r)) / max(np.max(np.abs(r)), 1e-300))
    except Exception as ex:  # noqa: BLE001
        fails.append((T, k, c, n, name, repr(ex)[:160]))
        continue
    worst = max(worst, e)
    lim = 1e-6 if name != "Sequential" else 5e-6  # a cascade multiplies the stop-band leakage
    if not np.isfinite(e) or e > lim:
        fails.append((T, k, c, n, name, e))
print("worst", worst, "failures", len(fails))
for f in fails[:20]:
    print("  ", f)
