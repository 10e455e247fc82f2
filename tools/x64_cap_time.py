"""Dev tool: what would a larger byte cap of the float64 route cost?  Paired-input transfer functions of ~60 frames of
8192 / 16384 samples (the shapes of tests/sweeps/fuzz_parity.py that read 1.0-1.7e-6 in the coherence on the fp32 kernels:
320-530 MB of frame spectra, over the 256 MB cap), end to end through backend.welch_transfer_function with float64 host
arrays, float64 kernels against fp32 kernels."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd._lib import get_context  # noqa: E402

ctx = get_context()
rng = np.random.default_rng(0)
for W, n, C, ov in ((16384, 495295, 20, 50.0), (16384, 743700, 33, 25.0), (8192, 361976, 33, 0.0), (16384, 2**20, 33, 50.0),
                    (8192, 2**19, 64, 50.0), (4096, 2**18, 64, 50.0)):
    x = rng.standard_normal((n, C))
    y = rng.standard_normal((n, C))
    hop = W - int(ov / 100 * W)
    frames = int(np.ceil(n / hop))
    mb = 2 * C * frames * (W // 2 + 1) * 16 / 2**20
    for prec in ("f64", "f32"):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            tf, coh = backend.welch_transfer_function(y, x, 48000, W, "H1", overlap_percent=ov, precision=prec)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        ctx.profile_enable(True)
        backend.welch_transfer_function(y, x, 48000, W, "H1", overlap_percent=ov, precision=prec)
        ctx.sync()
        rep = ctx.profile_report()
        ctx.profile_enable(False)
        print(f"W {W:6d} {C:3d}+{C:3d} ch {frames:4d} frames {mb:7.0f} MB of frame spectra  {prec}: {best:7.1f} ms end to end, "
              f"kernels {sum(v[0] for v in rep.values()):7.2f} ms", flush=True)

# the same question for the Welch spectra themselves and for the cross-spectral matrix (backend._welch / _csm_welch):
# "auto" with the cap lifted against "f32"
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402

backend._X64_SHORT_BYTES = 1 << 40
# (100 frames of 4096 samples x 64 channels is the advisor's example of a shape the fp32 register kernels handle in well
# under a millisecond: what does the CALL cost on either route?)
for W, n, C in ((8192, 2**19, 64), (4096, 2**18, 64), (16384, 2**19, 33), (1024, 2**16, 64), (4096, 100 * 2048, 64), (1024, 100 * 512, 64),
                (8192, 100 * 4096, 64)):
    y = rng.standard_normal((n, C))
    x = rng.standard_normal((n, C))
    frames = int(np.ceil(n / (W // 2)))
    for what in ("psd", "csd", "csm"):
        mb = (2 if what == "csd" else 1) * C * frames * (W // 2 + 1) * 16 / 2**20
        for prec in ("auto", "f32"):
            backend.SPEC_PRECISION = prec
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                if what == "csm":
                    backend._csm_welch(y, 48000, W, Window.Hann, 50, True, "mean", SpectrumScaling.FFTBackward)
                else:
                    backend._welch(y, x if what == "csd" else None, 48000, Window.Hann, W, 50, True, "mean", SpectrumScaling.FFTBackward)
                best = min(best, (time.perf_counter() - t0) * 1e3)
            ctx.routes()
            ctx.profile_enable(True)
            if what == "csm":
                backend._csm_welch(y, 48000, W, Window.Hann, 50, True, "mean", SpectrumScaling.FFTBackward)
            else:
                backend._welch(y, x if what == "csd" else None, 48000, Window.Hann, W, 50, True, "mean", SpectrumScaling.FFTBackward)
            ctx.sync()
            rep = ctx.profile_report()
            ctx.profile_enable(False)
            print(f"{what} W {W:6d} {C:3d} ch {frames:4d} frames {mb:7.0f} MB of frame spectra  {prec:4s}: {best:7.1f} ms end to end, "
                  f"kernels {sum(v[0] for v in rep.values()):7.2f} ms  routes {sorted(ctx.routes())[:3]}", flush=True)
