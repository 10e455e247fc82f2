"""Dev tool: randomized delay-and-sum maps, chroma features, zero-padded / power-scaled STFTs,
median CSMs and transfer functions with one input channel per output channel against the oracle."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dsptoolbox_amd as dsp  # noqa: E402
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, fails = {}, []
fs = 48000
scalings = list(SpectrumScaling)
for it in range(n_cases):
    kind = str(rng.choice(["das", "chroma", "stft_pad", "csm_median", "tf_paired"]))
    info = None
    try:
        if kind == "das":
            F, Cn, G = int(rng.integers(1, 40)), int(rng.choice([2, 3, 8, 31, 32, 33, 64, 70])), int(rng.integers(1, 900))
            info = (F, Cn, G)
            a = rng.standard_normal((F, Cn, Cn + 5)) + 1j * rng.standard_normal((F, Cn, Cn + 5))
            csm = a @ np.conj(np.swapaxes(a, 1, 2)) / (Cn + 5)
            mic = rng.uniform(-0.5, 0.5, (Cn, 3))
            grid = np.c_[rng.uniform(-1, 1, (G, 2)), np.full(G, 1.5)]
            r = np.linalg.norm(mic[:, None, :] - grid[None, :, :], axis=-1)
            f = np.linspace(800.0, 2500.0, F) if F > 1 else np.array([1000.0])
            h = np.exp(-2j * np.pi * f[:, None, None] / 343.0 * r[None]) / r[None] / Cn
            rm = bool(rng.integers(0, 2))
            m = dsp.beamforming.delay_and_sum_map(f, csm, h, rm)
            e = orc.rel_max(np.atleast_1d(m), np.atleast_1d(orc.das_map(f, csm, h, rm)))
            lim = 1e-6
        elif kind == "chroma":
            W = int(rng.choice([1024, 2048, 4096]))
            n = int(rng.integers(6 * W, 20 * W))
            c = int(rng.integers(1, 3))
            info = (W, n, c)
            t = np.arange(n) / 22050
            y = 0.05 * rng.standard_normal((n, c)) + 0.4 * np.sin(2 * np.pi * rng.uniform(200, 900) * t)[:, None]
            s = dsp.Signal(None, y, 22050)
            s.set_spectrogram_parameters(window_length_samples=W)
            comp = float(rng.choice([0.5, 2.0, 10.0]))
            tt, chroma, pitch = dsp.transforms.chroma_stft(s, compression=comp)
            rt, rf, rs = orc.stft(y, 22050, W, "hann", 50.0, None, False, True, "FFTBackward")
            rc, rp = orc.chroma_stft(rs, rf, 440, comp)
            e = max(orc.rel_max(chroma, rc), orc.rel_max(pitch, rp))
            lim = 1e-6
        elif kind == "stft_pad":
            W = int(rng.choice([128, 256, 512, 1024, 2048]))
            nfft = W * int(rng.choice([1, 2, 4]))
            n = int(rng.integers(2 * W, 30 * W))
            c = int(rng.integers(1, 10))
            sc = scalings[int(rng.integers(0, len(scalings)))]
            det, pad = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            ov = float(rng.choice([0, 25, 50, 75]))
            info = (W, nfft, n, c, sc.name, det, pad, ov)
            y = rng.standard_normal((n, c)) * 0.3 + 0.05
            t_, f_, st = backend._stft(y, fs, W, Window.Hann, ov, nfft, det, pad, sc)
            rt, rf, rs = orc.stft(y, fs, W, "hann", ov, nfft, det, pad, sc.name)
            assert st.shape == rs.shape and st.dtype == rs.dtype, (st.shape, rs.shape, st.dtype, rs.dtype)
            e = orc.rel_max(st[1:] if det else st, rs[1:] if det else rs)
            lim = 1e-6
        elif kind == "csm_median":
            W = int(rng.choice([64, 256, 1024]))
            c = int(rng.integers(2, 6))
            n = int(rng.integers(30 * W, 120 * W))
            info = (W, c, n)
            common = rng.standard_normal(n)
            y = np.stack([0.3 * rng.standard_normal(n) + 0.5 * np.roll(common, 3 * j) for j in range(c)], axis=1)
            f, csm = backend._csm_welch(y, fs, W, Window.Hann, 50, True, "median", SpectrumScaling.FFTBackward)
            rf, rcsm = orc.csm_welch(y, fs, W, "hann", 50, True, "median", "FFTBackward")
            e = orc.rel_l2(csm, rcsm)
            lim = 1e-6
        else:
            W = int(rng.choice([256, 1024, 4096]))
            c = int(rng.integers(2, 5))
            n = int(rng.integers(50 * W, 120 * W))
            info = (W, c, n)
            x = rng.standard_normal((n, c)) * 0.4
            h = rng.standard_normal((8, c)) * 0.5 + 1.0
            y = np.stack([np.convolve(x[:, j], h[:, j])[:n] for j in range(c)], axis=1) + 0.01 * rng.standard_normal((n, c))
            mode = str(rng.choice(["H1", "H3"]))
            tf, coh = backend.welch_transfer_function(y, x, fs, W, mode)
            rt, rc = orc.compute_transfer_function(y, x, fs, W, mode)
            e_tf, e_coh = orc.rel_max(tf[1:], rt[1:]), orc.rel_max(coh[1:], rc[1:])
            e = max(e_tf, e_coh)
            lim = 1e-6
            if e > lim:  # where: an estimate is ill-conditioned in float64 too where the coherence is small
                d = np.abs(tf[1:] - rt[1:]) if e_tf >= e_coh else np.abs(coh[1:] - rc[1:])
                b, ch = np.unravel_index(np.argmax(d), d.shape)
                info = info + (mode, f"e_tf={e_tf:.2e} e_coh={e_coh:.2e} worst bin {b + 1} ch {ch} "
                                     f"coh_ref there={rc[b + 1, ch]:.4f} |tf_ref| there={abs(rt[b + 1, ch]):.3e} "
                                     f"max|tf_ref|={np.abs(rt[1:]).max():.3e} min coh_ref={rc[1:].min():.2e}")
    except Exception as ex:  # noqa: BLE001
        fails.append((kind, info, repr(ex)[:200]))
        continue
    worst[kind] = max(worst.get(kind, 0.0), e)
    if not np.isfinite(e) or e > lim:
        fails.append((kind, info, e))
print("worst", worst, "failures", len(fails))
for f in fails[:20]:
    print("  ", f)
