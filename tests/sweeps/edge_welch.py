"""Edge shapes of the Welch kernels: one ... five frames, one to three channels, paired inputs, cross spectra,
cross-spectral matrices -- the worst error PER OUTPUT KIND (psd, csd, csm, tf, coh), once on the fp32 kernels
alone (backend.SPEC_PRECISION = "f32", welch_transfer_function(precision="f32")) and once the way the API
routes them ("auto": short estimates take the float64 kernels).

    python tests/sweeps/edge_welch.py            (on the GPU box; exit status 1 if the API route misses 1e-6 for
                                                  psd / csd / csm / tf / coh, or the fp32 kernels leave their own
                                                  documented bands: psd 1e-6, csd / csm 5e-6, tf / coh 2e-5)
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
KINDS = ("psd", "csd", "csm", "csm_median", "tf", "coh")
# (a median picks single-frame values: the fp32 matrix of one to five frames is only good to ~1e-5; round 5 gave the API's route
# a float64 median matrix kernel)
LIMIT = {"f32": dict(psd=1e-6, csd=5e-6, csm=5e-6, csm_median=5e-5, tf=2e-5, coh=2e-5), "auto": {k: 1e-6 for k in KINDS}}


def rel(a, b, lo):
    a, b = np.asarray(a)[lo:], np.asarray(b)[lo:]
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def sweep(mode: str):
    backend.SPEC_PRECISION = mode
    rng = np.random.default_rng(5)
    worst = {k: (0.0, None) for k in KINDS}
    bad = 0

    def note(kind, err, case):
        nonlocal bad
        if not np.isfinite(err) or err > worst[kind][0]:
            worst[kind] = (err, case)
        if not np.isfinite(err) or err > LIMIT[mode][kind]:
            bad += 1
            print(f"  BAD [{mode}] {kind} {err:.2e} {case}")

    for W in (32, 64, 128, 256, 1024, 2048, 4096, 8192, 16384):
        for n in (W, W + 1, W + W // 2, 2 * W, 2 * W + 5, 3 * W - 1, 5 * W + 17):
            for C in (1, 2, 3):
                for det in (False, True):
                    x = rng.standard_normal((n, C)) * 0.3
                    y = np.stack([np.convolve(x[:, i], rng.standard_normal(5))[:n] for i in range(C)], axis=1)
                    y += 0.01 * rng.standard_normal((n, C))
                    lo = 1 if det else 0
                    for sc in (SpectrumScaling.FFTBackward, SpectrumScaling.PowerSpectralDensity):
                        case = (W, n, C, det, sc.name)
                        a = backend._welch(y, None, 48000, Window.Hann, W, 50, det, "mean", sc)
                        note("psd", rel(a, orc.welch(y, None, 48000, "hann", W, 50, det, "mean", sc.name), lo), case)
                        k = backend._welch(x, y, 48000, Window.Hann, W, 50, det, "mean", sc)
                        note("csd", rel(k, orc.welch(x, y, 48000, "hann", W, 50, det, "mean", sc.name), lo), case)
                        if C > 1:
                            _, m = backend._csm_welch(y, 48000, W, Window.Hann, 50, det, "mean", sc)
                            note("csm", rel(m, orc.csm_welch(y, 48000, W, "hann", 50, det, "mean", sc.name)[1], lo), case)
                            if W <= 1024:  # (the oracle's pair loop of median Welch calls)
                                _, m = backend._csm_welch(y, 48000, W, Window.Hann, 50, det, "median", sc)
                                note("csm_median", rel(m, orc.csm_welch(y, 48000, W, "hann", 50, det, "median", sc.name)[1], lo), case)
                    for xin in (x, x[:, :1]):
                        yy = y if xin.shape[1] == C else np.stack(
                            [np.convolve(x[:, 0], rng.standard_normal(5))[:n] for _ in range(C)], axis=1)
                        tf, coh = backend.welch_transfer_function(yy, xin, 48000, W, "H1", detrend=det, precision=mode)
                        rt, rc = orc.compute_transfer_function(yy, xin, 48000, W, "H1", detrend=det)
                        note("tf", rel(tf, rt, lo), (W, n, C, det, xin.shape[1]))
                        note("coh", rel(coh, rc, lo), (W, n, C, det, xin.shape[1]))
    print(f"[{mode}] worst per output kind:")
    for k in KINDS:
        print(f"  {k:4s} {worst[k][0]:.2e}  at (W, n, C, detrend, scaling | input channels) = {worst[k][1]}")
    return bad


if __name__ == "__main__":
    total = sweep("f32") + sweep("auto")
    print("cases outside their band:", total)
    sys.exit(1 if total else 0)
