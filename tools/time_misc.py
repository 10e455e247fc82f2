"""Dev tool: wall time of less-travelled entry points on the headline shape (host arrays in, so PCIe is included;
the kernel times are in the library's profile report)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd._lib import get_context  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402

ctx = get_context()
rng = np.random.default_rng(0)
n, n_ch = 2**18, 16
x = rng.standard_normal((n, 1)) * 0.3
y = rng.standard_normal((n, n_ch)) * 0.3


def run(name, fn, reps=3):
    fn()
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_report()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps * 1e3
    prof = ctx.profile_report()
    ctx.profile_enable(False)
    ks = ", ".join(f"{k} {v[0] / max(1, v[1]):.3f} ms" for k, v in sorted(prof.items(), key=lambda t: -t[1][0])[:4]) if isinstance(prof, dict) else str(prof)[:200]
    print(f"{name:40s} {dt:8.2f} ms wall | {ks}", flush=True)


for W in (1024, 4096):
    run(f"tf mean W={W}", lambda: backend.welch_transfer_function(y, x, 48000, W, "H1", precision="f32"))
    run(f"tf median W={W}", lambda: backend.welch_transfer_function(y, x, 48000, W, "H1", average="median", precision="f32"))
    run(f"psd median W={W}", lambda: backend._welch(y, None, 48000, Window.Hann, W, 50, True, "median", SpectrumScaling.FFTBackward))
    run(f"csm median W={W}", lambda: backend._csm_welch(y, 48000, W, Window.Hann, 50, True, "median", SpectrumScaling.FFTBackward))
