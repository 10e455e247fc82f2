"""Randomized parity sweep of round 4's long-window routes against the oracle (run by hand on a GPU box):
  * Welch with windows of 16384 ... 262144 samples (kernels_welch_long.hpp): shared / paired transfer functions, cross and
    auto spectra, 3 ... 40 frames: the fp32 kernels forced AND the API's default arithmetic ("auto": such estimates are
    short and take the float64 route, k_frames_cls / k_split of kernels_welch_f64.hpp), the latter held to 1e-6;
  * STFT with frames of 32768 / 65536 samples and transform lengths up to 262144 (kernels_stft_long.hpp);
  * FIR banks whose signal is shorter than the filter (direct float64 sum);
  * inverse STFT with frames of 8192 ... 65536 samples, transform lengths up to 131072, every overlap (kernels_istft_long.hpp).
usage: python tests/sweeps/fuzz_long_windows.py [n_cases] [seed]
Limits: auto spectra, STFT, FIR 1e-6; cross spectra / transfer functions / coherence of these few-frame fp32 estimates are
REPORTED per kind (worst value) and flagged above 2e-5 -- a defect in the class bookkeeping shows as 1e-2 ... 1, rounding as 1e-6."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd._lib import get_context  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

scalings = list(SpectrumScaling)


def relmax(a, b, skip_dc=False):
    a, b = np.asarray(a), np.asarray(b)
    if skip_dc:
        a, b = a[1:], b[1:]
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    backend.SPEC_PRECISION = "f32"
    ctx = get_context()
    worst, fails, routes = {}, [], {}
    for it in range(n_cases):
        kind = str(rng.choice(["tf", "tf_paired", "csd", "psd", "stft", "stft", "fir", "fir", "istft", "istft"]))
        det = bool(rng.integers(0, 2))
        sc = scalings[int(rng.integers(0, len(scalings)))]
        ov = float(rng.choice([0, 25, 50, 50, 75]))
        ctx.routes()
        limit, desc = 1e-6, None
        try:
            if kind in ("tf", "tf_paired", "csd", "psd"):
                W = int(rng.choice([16384, 32768, 65536, 131072, 262144]))
                hop = W - int(ov / 100 * W)
                frames = int(rng.integers(3, 41 if W <= 65536 else 13))
                n = frames * hop + int(rng.integers(-hop + 1, hop))
                n_ch = int(rng.choice([1, 2, 3, 5, 8]))
                desc = (kind, W, n, n_ch, ov, det, sc.name)
                h = rng.standard_normal((32, n_ch)) * np.exp(-np.arange(32) / 6.0)[:, None]
                n_in = 1 if kind in ("tf", "psd") else n_ch
                xs = rng.standard_normal((n, n_in)) * 0.3 + 0.05
                ys = np.stack([np.convolve(xs[:, c % n_in], h[:, c])[:n] for c in range(n_ch)], axis=1)
                ys += 0.05 * rng.standard_normal((n, n_ch))
                def auto(name, got, ref):  # the API's default arithmetic: these estimates are short -> float64 route
                    ea = relmax(got, ref, det)
                    worst[name + " (auto)"] = max(worst.get(name + " (auto)", 0.0), ea)
                    if not ea <= 1e-6:
                        fails.append(desc + ("auto", name, ea))

                if kind == "psd":
                    a = backend._welch(ys, None, 48000, Window.Hann, W, ov, det, "mean", sc)
                    r = orc.welch(ys, None, 48000, "hann", W, ov, det, "mean", sc.name)
                    e = relmax(a, r, det)
                    backend.SPEC_PRECISION = "auto"
                    auto("psd", backend._welch(ys, None, 48000, Window.Hann, W, ov, det, "mean", sc), r)
                    backend.SPEC_PRECISION = "f32"
                elif kind == "csd":
                    a = backend._welch(xs, ys, 48000, Window.Hann, W, ov, det, "mean", sc)
                    r = orc.welch(xs, ys, 48000, "hann", W, ov, det, "mean", sc.name)
                    e, limit = relmax(a, r, det), 2e-5
                    backend.SPEC_PRECISION = "auto"
                    a64 = backend._welch(xs, ys, 48000, Window.Hann, W, ov, det, "mean", sc)
                    auto("csd", a64, r)
                    backend.SPEC_PRECISION = "f32"
                    if os.environ.get("FUZZ_DEBUG") and e > 1e-3:
                        lo = 1 if det else 0
                        d = np.abs(np.asarray(a) - r)[lo:]
                        b_, c_ = np.unravel_index(np.argmax(d), d.shape)
                        print("   DEBUG worst bin", b_ + lo, "ch", c_, "got", a[b_ + lo, c_], "ref", r[b_ + lo, c_], "max|ref|", np.max(np.abs(r[lo:])),
                              "per-channel err", [float(np.max(d[:, k]) / np.max(np.abs(r[lo:]))) for k in range(n_ch)],
                              "bins > 1e-3:", int(np.sum(d > 1e-3 * np.max(np.abs(r[lo:])))), "of", d.size)
                        big = backend._X64_SHORT_BYTES
                        backend._X64_SHORT_BYTES = 1 << 40
                        backend.SPEC_PRECISION = "auto"
                        a64b = backend._welch(xs, ys, 48000, Window.Hann, W, ov, det, "mean", sc)
                        backend.SPEC_PRECISION = "f32"
                        backend._X64_SHORT_BYTES = big
                        print("   DEBUG float64 route vs oracle", relmax(a64b, r, det), "fp32 vs float64 route", relmax(a, a64b, det))
                else:
                    mode = str(rng.choice(["H1", "H2", "H3"]))
                    desc += (mode,)
                    tf, coh = backend.welch_transfer_function(ys, xs, 48000, W, mode, overlap_percent=ov, detrend=det,
                                                              scaling=sc, precision="f32")
                    if kind == "tf":
                        rt, rc = orc.compute_transfer_function_batched(ys, xs, 48000, W, mode, overlap_percent=ov,
                                                                       detrend=det, scaling=sc.name)
                    else:
                        rt, rc = orc.compute_transfer_function(ys, xs, 48000, W, mode, overlap_percent=ov, detrend=det,
                                                               scaling=sc.name)
                    tfa, coha = backend.welch_transfer_function(ys, xs, 48000, W, mode, overlap_percent=ov, detrend=det,
                                                                scaling=sc, precision="auto")
                    if mode == "H2":
                        tf = np.where(rc > 0.1, tf, rt)
                        tfa = np.where(rc > 0.1, tfa, rt)
                    auto(kind, tfa, rt)
                    auto(kind + " coh", coha, rc)
                    e_tf, e_coh = relmax(tf, rt, det), relmax(coh, rc, det)
                    worst[kind + " coh"] = max(worst.get(kind + " coh", 0.0), e_coh)
                    e, limit = max(e_tf, e_coh), 2e-5
            elif kind == "stft":
                W = int(rng.choice([32768, 65536]))
                nfft = [None, None, 2 * W, 4 * W][int(rng.integers(0, 4))]
                if nfft is not None and nfft > 262144:
                    nfft = 262144
                ovs = int(ov / 100 * W + 0.5)
                hop = W - ovs
                frames = int(rng.integers(1, 10))
                n = max(16, frames * hop + int(rng.integers(-hop + 1, hop)))
                n_ch = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 20]))
                pad = bool(rng.integers(0, 2))
                desc = (kind, W, nfft, n, n_ch, ov, det, pad, sc.name)
                x = rng.standard_normal((n, n_ch)) * 0.3 + 0.05
                t, f, st = backend._stft(x, 48000, W, Window.Hann, ov, nfft, det, pad, sc)
                rt_, rf_, rs = orc.stft(x, 48000, W, "hann", ov, nfft, det, pad, sc.name)
                assert st.shape == rs.shape
                e = relmax(st, rs)
            elif kind == "istft":  # random spectrograms (not the transform of a signal: every frame half matters)
                import dsptoolbox_amd as dsp
                W = int(rng.choice([8192, 16384, 32768, 65536]))
                nfft = [None, None, None, 2 * W][int(rng.integers(0, 4))]
                n_frames = int(rng.integers(1, 14))
                n_ch = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 17]))
                pad = bool(rng.integers(0, 2))
                if pad and n_frames < 2 and ov > 0:
                    n_frames = 2
                if pad and ov == 0:  # the reference's td[overlap:-overlap] is EMPTY for overlap 0 (and a Signal of it (C, 0))
                    pad = False
                nb = (nfft or W) // 2 + 1
                desc = (kind, W, nfft, n_frames, n_ch, ov, pad)
                sp = rng.standard_normal((nb, n_frames, n_ch)) + 1j * rng.standard_normal((nb, n_frames, n_ch))
                sp[0].imag = 0
                sp[-1].imag = 0
                got = dsp.transforms.istft(sp, sampling_rate_hz=48000, window_length_samples=W, window_type=Window.Hann,
                                           overlap_percent=ov, fft_length_samples=nfft, padding=pad,
                                           scaling=SpectrumScaling.FFTBackward)
                ref = orc.istft(sp, 48000, W, "hann", ov, nfft, pad, "FFTBackward")
                assert got.time_data.shape == ref.shape, (got.time_data.shape, ref.shape)
                e = relmax(got.time_data, ref)
            else:
                n_taps = int(rng.integers(2, 9000))
                n = int(rng.integers(1, n_taps))
                n_ch = int(rng.integers(1, 5))
                n_filt = int(rng.integers(1, 4))
                mode, name = [(backend.DS_FB_PARALLEL, "Parallel"), (backend.DS_FB_SUMMED, "Summed"),
                              (backend.DS_FB_SEQUENTIAL, "Sequential")][int(rng.integers(0, 3))]
                desc = (kind, n_taps, n, n_ch, n_filt, name)
                x = rng.standard_normal((n, n_ch)) * 0.2
                taps = [rng.standard_normal(n_taps) * np.hanning(n_taps + 2)[1:-1] / np.sqrt(n_taps) for _ in range(n_filt)]
                y = backend.fir_filter_bank(x, taps, mode)
                r = orc.filterbank_fir(taps, x, name)
                r = np.transpose(r, (2, 0, 1)) if name == "Parallel" else r
                e = relmax(y, r)
        except Exception as ex:  # noqa: BLE001
            fails.append(desc + (repr(ex)[:200],) if desc else (kind, repr(ex)[:200]))
            continue
        for r_ in ctx.routes():
            routes[r_] = routes.get(r_, 0) + 1
        worst[kind] = max(worst.get(kind, 0.0), e)
        if not np.isfinite(e) or e > limit:
            fails.append(desc + (e,))
        print(f"#{it} {desc} {e:.2e}", flush=True)
    print("worst", {k: f"{v:.2e}" for k, v in worst.items()})
    print("routes", routes)
    print("failures", len(fails))
    for f in fails[:30]:
        print("  ", f)


if __name__ == "__main__":
    main()
