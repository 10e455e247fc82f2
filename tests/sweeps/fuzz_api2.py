"""Dev tool: randomized inverse STFTs (the device STFT of a random signal, then the device inverse,
against the oracle's inverse of the same spectrogram), zero-phase / stateful FIR filtering through
Filter.filter_signal, RIR convolution and log-mel spectrograms against the oracle."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dsptoolbox_amd as dsp  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

warnings.simplefilter("ignore")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, fails = {}, []
fs = 48000
# (density scalings divide by the sampling rate, which istft(original_signal=...) does not know: the
# reference raises the same TypeError there, transforms/transforms.py:556-559)
amp_scalings = [s for s in SpectrumScaling if s.is_amplitude_scaling() and "Density" not in s.name]
for it in range(n_cases):
    kind = str(rng.choice(["istft", "zero_phase", "state", "rir", "mel"]))
    info = None
    try:
        if kind == "istft":
            W = int(rng.choice([64, 256, 512, 1024, 2048, 4096]))
            ov = float(rng.choice([50, 75, 25]))  # (0 %: the squared-window envelope hits its 1e-4 clip)
            pad = bool(rng.integers(0, 2))
            sc = amp_scalings[int(rng.integers(0, len(amp_scalings)))]
            n = int(rng.integers(3 * W, 40 * W))
            c = int(rng.integers(1, 4))
            info = (W, ov, pad, sc.name, n, c)
            y = rng.standard_normal((n, c)) * 0.3
            s = dsp.Signal(None, y, fs)
            s.set_spectrogram_parameters(window_length_samples=W, overlap_percent=ov, padding=pad, scaling=sc,
                                         detrend=False)
            t, f, st = s.get_spectrogram()
            rec = dsp.transforms.istft(st, original_signal=s)
            ref = orc.istft(st, fs, W, "hann", ov, W, pad, sc.name, original_length=n)
            e = orc.rel_max(rec.time_data, ref)
            lim = 1e-6
        elif kind in ("zero_phase", "state"):
            T = int(rng.choice([rng.integers(2, 60), rng.integers(60, 600), rng.integers(600, 3000)]))
            n = int(rng.integers(4 * T, 4 * T + 60000))
            c = int(rng.integers(1, 4))
            info = (T, n, c)
            b = rng.standard_normal(T) * np.exp(-np.arange(T) / max(2.0, T / 5.0)) * 0.3
            y = rng.standard_normal((n, c)) * 0.3
            flt = dsp.Filter.from_ba(b, [1.0], fs)
            if kind == "zero_phase":
                out = flt.filter_signal(dsp.Signal(None, y, fs), zero_phase=True)
                ref = orc.filtfilt_fir(b, y)
            else:
                flt.initialize_zi(c)
                out = flt.filter_signal(dsp.Signal(None, y, fs), activate_zi=True)
                zi = np.stack([orc.lfilter_zi_fir(b)] * c, axis=1)
                ref, _ = orc.lfilter_fir(b, y, zi)
            e = orc.rel_max(out.time_data, ref)
            lim = 2e-6
        elif kind == "rir":
            T = int(rng.integers(100, 30000))
            n = int(rng.integers(1000, 100000))
            c = int(rng.integers(1, 3))
            info = (T, n, c)
            h = rng.standard_normal(T) * np.exp(-np.arange(T) / (T / 6.0)) * 0.2
            y = rng.standard_normal((n, c)) * 0.3
            keep_len, keep_peak = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            out = dsp.room_acoustics.convolve_rir_on_signal(dsp.Signal(None, y, fs), dsp.ImpulseResponse(None, h, fs),
                                                            keep_peak_level=keep_peak, keep_length=keep_len)
            ref = orc.convolve_rir_on_signal(y, h, keep_peak, keep_len)
            e = orc.rel_max(out.time_data, ref)
            lim = 1e-6
        else:
            W = int(rng.choice([256, 512, 1024, 2048]))
            n = int(rng.integers(4 * W, 30 * W))
            c = int(rng.integers(1, 3))
            info = (W, n, c)
            y = rng.standard_normal((n, c)) * 0.3 + 0.3 * np.sin(2 * np.pi * 440 * np.arange(n) / fs)[:, None]
            s = dsp.Signal(None, y, fs)
            s.set_spectrogram_parameters(window_length_samples=W)
            t, fm, lm = dsp.transforms.log_mel_spectrogram(s, n_bands=24, generate_plot=False)
            rt, rf, rs = orc.stft(y, fs, W, "hann", 50.0, None, False, True, "FFTBackward")  # defaults: no detrend
            _, ref = orc.log_mel_spectrogram(rs, rf, None, 24)
            # frames that are mostly zero padding share a complex transform with a full-level
            # neighbour and keep ~1e-7 of its amplitude: compare the frames within
            # 30 dB of the loudest one
            level = ref.max(axis=0, keepdims=True)
            with np.errstate(invalid="ignore"):
                ok = (ref > -1000.0) & (level > np.nanmax(level) - 30.0)
            if not ok.any():  # 24 bands over too few bins give NaN filters in the reference too
                continue
            e = orc.rel_max(lm[ok], ref[ok])
            lim = 1e-6
    except Exception as ex:  # noqa: BLE001
        fails.append((kind, info, repr(ex)[:200]))
        continue
    worst[kind] = max(worst.get(kind, 0.0), e)
    if not np.isfinite(e) or e > lim:
        fails.append((kind, info, e))
print("worst", worst, "failures", len(fails))
for f in fails[:20]:
    print("  ", f)
