"""Generate golden vectors by RUNNING THE REFERENCE (dsptoolbox 0.8).

Runs only in the build container, where /root/reference exists.  The reference
never travels: only the seeded inputs and the outputs it produced are written,
as small .npz fixtures under tests/golden/.  Re-run with

    python oracle/gen_golden.py            (all fixtures; --only <name> ...: some; --out DIR: elsewhere)

Recorded skew: the reference pins numpy~=2.4 / scipy~=1.17 and python>=3.11;
this container has python 3.10, numpy 2.2.6, scipy 1.15.3.  The import recipe
is SURVEY.md section 8(c): alias typing.Self, register placeholder modules for the
absent audio-IO / plotting packages (never touched by the hot path).
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    import typing
    from unittest.mock import MagicMock

    import typing_extensions

    sys.dont_write_bytecode = True
    typing.Self = typing_extensions.Self
    for m in ("soundfile", "sounddevice", "seaborn"):
        sys.modules[m] = MagicMock()
    import matplotlib

    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import dsptoolbox as dsp

    return dsp


def save(name, meta, arrays):
    os.makedirs(OUT, exist_ok=True)
    meta = dict(meta)
    meta["generator"] = "oracle/gen_golden.py"
    meta["reference"] = "dsptoolbox 0.8 source @ /root/reference"
    import scipy

    meta["numpy"] = np.__version__
    meta["scipy"] = scipy.__version__
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(meta['cases'])} cases")


def gen_stft_anylen(dsp):
    """Signal.get_spectrogram with fft_length_samples that are not powers of two, or shorter than
    the window (standard/_spectral_methods.py:268: np.fft.rfft(..., n=fft_length_samples) crops or
    zero-pads each frame; the edge-bin division by sqrt 2 touches the last bin only for even n)."""
    from dsptoolbox.standard.enums import SpectrumScaling as S, Window
    fs = 48000
    rng = np.random.default_rng(41)
    xs = rng.standard_normal((1500, 2)) * 0.4 + 0.1
    cases, arrs = [], {"x": xs}
    combos = [
        (256, 50, 255, True, False, S.FFTBackward),              # W - 1: crop, odd
        (256, 50, 384, True, True, S.FFTBackward),               # 3 * 2^7: zero-pad
        (256, 50, 1000, False, False, S.AmplitudeSpectrum),      # 2^3 5^3
        (512, 50, 1023, True, False, S.PowerSpectralDensity),    # odd: no Nyquist bin
        (128, 0, 96, True, True, S.PowerSpectrum),               # 3 * 2^5 < W: crop
        (1024, 50, 100, False, False, S.FFTOrthogonal),          # far below W
        (64, 33, 600, True, False, S.AmplitudeSpectralDensity),
        (256, 50, 257, True, False, S.FFTForward),               # prime
    ]
    for i, (W, ov, nfft, det, pad, sc) in enumerate(combos):
        s = dsp.Signal(None, xs.copy(), fs)
        s.set_spectrogram_parameters(window_length_samples=W, window_type=Window.Hann, overlap_percent=ov,
                                     fft_length_samples=nfft, detrend=det, padding=pad, scaling=sc)
        t, f, st = s.get_spectrogram()
        cases.append(dict(W=W, overlap=ov, fft_length=nfft, detrend=det, padding=pad, scaling=sc.name))
        arrs[f"t_{i}"], arrs[f"f_{i}"], arrs[f"stft_{i}"] = t, f, st
    save("stft_anylen", dict(cases=cases, fs=fs), arrs)


def gen_stft_manych(dsp):
    """Signal.get_spectrogram of a 20-channel signal (more channels than one workgroup of the
    wave-level STFT kernels takes: a full tile of 16 and a ragged one of 4) at windows 256 ... 2048;
    every 4th bin and the edge bins, stored as complex64."""
    from dsptoolbox.standard.enums import SpectrumScaling as S, Window
    fs = 48000
    rng = np.random.default_rng(43)
    xs = (rng.standard_normal((7000, 20)) * 0.4 + 0.1).astype(np.float32)
    cases, arrs = [], {"x": xs}
    for i, (W, ov, det, pad, sc) in enumerate(((1024, 50, True, False, S.FFTBackward),
                                               (512, 75, False, True, S.AmplitudeSpectrum),
                                               (256, 50, True, True, S.PowerSpectralDensity),
                                               (2048, 50, False, False, S.FFTBackward))):
        s = dsp.Signal(None, xs.astype(np.float64), fs)
        s.set_spectrogram_parameters(window_length_samples=W, window_type=Window.Hann, overlap_percent=ov,
                                     detrend=det, padding=pad, scaling=sc)
        t, f, st = s.get_spectrogram()
        nb = W // 2 + 1
        bins = np.unique(np.r_[0:3, 0:nb:4, nb - 3:nb])
        cases.append(dict(W=W, overlap=ov, detrend=det, padding=pad, scaling=sc.name, shape=list(st.shape)))
        arrs[f"bins_{i}"], arrs[f"stft_{i}"] = bins, st[bins].astype(np.complex64)
    save("stft_manych", dict(cases=cases, fs=fs), arrs)


def gen_stft_long(dsp):
    """Signal.get_spectrogram with windows of 4096, 8192 and 16384 samples (the frame kernels of
    dsptoolbox_amd/csrc/kernels_stft4096.hpp: four teams of two neighbouring channels per workgroup; 8192 / 16384 by
    decimation in frequency, residues 1 and 3 of the 16384-point frames on neighbouring teams) of a 10-channel
    signal: one workgroup with 8 channels, one with 2 (idle teams).  Every 11th bin (all residues modulo 2 and 4)
    and the edge bins, stored as complex64."""
    from dsptoolbox.standard.enums import SpectrumScaling as S, Window
    fs = 48000
    rng = np.random.default_rng(47)
    xs = (rng.standard_normal((24000, 10)) * 0.4 + 0.1).astype(np.float32)
    cases, arrs = [], {"x": xs}
    for i, (W, ov, det, pad, sc) in enumerate(((4096, 50, True, False, S.FFTBackward),
                                               (4096, 25, False, True, S.PowerSpectralDensity),
                                               (8192, 50, False, True, S.AmplitudeSpectrum),
                                               (8192, 25, True, False, S.FFTOrthogonal),
                                               (16384, 50, True, True, S.FFTBackward),
                                               (16384, 0, False, False, S.PowerSpectrum))):
        s = dsp.Signal(None, xs.astype(np.float64), fs)
        s.set_spectrogram_parameters(window_length_samples=W, window_type=Window.Hann, overlap_percent=ov,
                                     detrend=det, padding=pad, scaling=sc)
        t, f, st = s.get_spectrogram()
        nb = W // 2 + 1
        bins = np.unique(np.r_[0:4, 0:nb:11, nb - 4:nb])
        cases.append(dict(W=W, overlap=ov, detrend=det, padding=pad, scaling=sc.name, shape=list(st.shape)))
        arrs[f"bins_{i}"], arrs[f"stft_{i}"] = bins, st[bins].astype(np.complex64)
    save("stft_long", dict(cases=cases, fs=fs), arrs)


def gen_fir_state(dsp):
    """Filter state (zi), zero-phase and long FIR filters: Filter / FilterBank.filter_signal with
    activate_zi / zero_phase (classes/filter.py:648-743, filter_helpers.py:288-382, 454-503)."""
    from dsptoolbox.standard.enums import FilterBankMode, FilterPassType
    fs = 48000
    rng = np.random.default_rng(19)
    cases, arrs = [], {}
    x = rng.standard_normal((6000, 2)) * 0.1 + 0.05
    arrs["x"] = x
    # block-wise filtering with state: three consecutive blocks through one filter object
    for order in (64, 700):
        flt = dsp.Filter.fir_filter(order, 3000.0, FilterPassType.Lowpass, fs)
        arrs[f"b_{order}"] = flt.ba[0]
        flt.initialize_zi(2)
        arrs[f"zi0_{order}"] = np.asarray(flt.zi)
        outs = []
        for k, (a, b) in enumerate(((0, 2500), (2500, 3000), (3000, 6000))):
            o = flt.filter_signal(dsp.Signal(None, x[a:b].copy(), fs), activate_zi=True)
            outs.append(o.time_data)
            arrs[f"zi_{order}_{k}"] = np.asarray(flt.zi)
        arrs[f"y_zi_{order}"] = np.concatenate(outs, axis=0)
        cases.append(dict(kind="zi_blocks", order=order, blocks=[[0, 2500], [2500, 3000], [3000, 6000]]))
        # only channel 1 with state
        flt.initialize_zi(2)
        o = flt.filter_signal(dsp.Signal(None, x[:2000].copy(), fs), channels=[1], activate_zi=True)
        arrs[f"y_zi_ch1_{order}"] = o.time_data
        arrs[f"zi_ch1_{order}"] = np.asarray(flt.zi)
        cases.append(dict(kind="zi_channel", order=order, channels=[1], n=2000))
        # zero phase
        o = flt.filter_signal(dsp.Signal(None, x.copy(), fs), zero_phase=True)
        arrs[f"y_zp_{order}"] = o.time_data
        cases.append(dict(kind="zero_phase", order=order))
    # filter bank, zero phase and state
    flts = [dsp.Filter.fir_filter(200, [lo, hi], FilterPassType.Bandpass, fs)
            for (lo, hi) in ((100.0, 3000.0), (800.0, 12000.0))]
    arrs["bank_taps"] = np.stack([f.ba[0] for f in flts])
    fb = dsp.FilterBank(flts)
    for mode in FilterBankMode:
        o = fb.filter_signal(dsp.Signal(None, x.copy(), fs), mode, zero_phase=True)
        arrs[f"y_bank_zp_{mode.name}"] = (np.asarray(o.get_all_time_data()[0]) if mode == FilterBankMode.Parallel
                                          else o.time_data)
        cases.append(dict(kind="bank_zero_phase", mode=mode.name))
    fb.initialize_zi(2)
    o = fb.filter_signal(dsp.Signal(None, x[:3000].copy(), fs), FilterBankMode.Parallel, activate_zi=True)
    o2 = fb.filter_signal(dsp.Signal(None, x[3000:].copy(), fs), FilterBankMode.Parallel, activate_zi=True)
    arrs["y_bank_zi"] = np.concatenate([np.asarray(o.get_all_time_data()[0]),
                                        np.asarray(o2.get_all_time_data()[0])], axis=0)
    cases.append(dict(kind="bank_zi", mode="Parallel", blocks=[[0, 3000], [3000, 6000]]))
    # long filter: more taps than the largest LDS-resident block can take (> 8193)
    xl = rng.standard_normal((70000, 3)) * 0.1
    bl = rng.standard_normal(20001) * np.exp(-np.arange(20001) / 3000.0) * 0.05
    fl = dsp.Filter.from_ba(bl, [1.0], fs)
    o = fl.filter_signal(dsp.Signal(None, xl.copy(), fs))
    arrs["x_long"], arrs["b_long"], arrs["y_long"] = xl.astype(np.float32), bl, o.time_data.astype(np.float32)
    cases.append(dict(kind="long", taps=20001, note="x_long / y_long stored as float32"))
    save("fir_state", dict(cases=cases, fs=fs), arrs)


def gen_istft(dsp, name="istft", variants=None, seed=21):
    """transforms.istft (transforms/transforms.py:444-586) on spectrograms of the reference's own
    Signal.get_spectrogram, with the original signal and with an explicit parameter dict."""
    from dsptoolbox.standard.enums import SpectrumScaling, Window
    fs = 48000
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((6000, 2)) * 0.2
    cases, arrs = [], {"x": x}
    variants = variants or [
        dict(W=256, ov=50.0, nfft=None, pad=True, sc="FFTBackward", win="Hann"),
        dict(W=256, ov=75.0, nfft=512, pad=True, sc="FFTBackward", win="Hann"),
        dict(W=512, ov=50.0, nfft=None, pad=False, sc="FFTBackward", win="Hann"),
        dict(W=256, ov=50.0, nfft=None, pad=True, sc="AmplitudeSpectrum", win="Hamming"),
        dict(W=128, ov=60.0, nfft=None, pad=True, sc="FFTOrthogonal", win="Hann"),
    ]
    for i, v in enumerate(variants):
        s = dsp.Signal(None, x.copy(), fs)
        s.set_spectrogram_parameters(window_length_samples=v["W"], window_type=Window[v["win"]],
                                     overlap_percent=v["ov"], fft_length_samples=v["nfft"],
                                     detrend=False, padding=v["pad"], scaling=SpectrumScaling[v["sc"]])
        t, f, sp = s.get_spectrogram()
        arrs[f"stft_{i}"] = sp
        rec = dsp.transforms.istft(sp, original_signal=s)
        arrs[f"rec_sig_{i}"] = rec.time_data
        if v["nfft"] is not None or not SpectrumScaling[v["sc"]].has_physical_units():
            # (with fft_length None the parameter-dict branch only works for scalings without
            # physical units: get_scaling_factor needs a length)
            rec2 = dsp.transforms.istft(sp, parameters=dict(s._spectrogram_parameters), sampling_rate_hz=fs)
            arrs[f"rec_par_{i}"] = rec2.time_data
        v = dict(v)
        v["has_par"] = f"rec_par_{i}" in arrs
        cases.append(v)
    save(name, dict(cases=cases, fs=fs), arrs)


def gen_istft_anylen(dsp):
    """The same with fft_length_samples that are not powers of two: transforms.py:548-577 inverts with
    np.fft.irfft(stft, n=fft_length_samples) of any length (zero-padded frames: nfft >= window)."""
    gen_istft(dsp, "istft_anylen", [
        dict(W=256, ov=50.0, nfft=384, pad=True, sc="FFTBackward", win="Hann"),       # 3 * 2^7
        dict(W=256, ov=75.0, nfft=1000, pad=True, sc="FFTBackward", win="Hann"),      # 2^3 5^3
        dict(W=512, ov=50.0, nfft=600, pad=False, sc="AmplitudeSpectrum", win="Hann"),
        dict(W=128, ov=50.0, nfft=255, pad=True, sc="FFTOrthogonal", win="Hamming"),  # odd: no Nyquist bin
    ], seed=22)


def gen_rir(dsp):
    """room_acoustics.convolve_rir_on_signal (room_acoustics/room_acoustics.py:216-266)."""
    fs = 48000
    rng = np.random.default_rng(23)
    x = rng.standard_normal((40000, 2)) * 0.2
    cases, arrs = [], {"x": x.astype(np.float32)}
    for i, (T, kp, kl) in enumerate(((1500, True, True), (12000, True, False), (12000, False, True))):
        h = rng.standard_normal(T) * np.exp(-np.arange(T) / (T / 6.0)) * 0.1
        sig = dsp.Signal(None, x.astype(np.float32).astype(np.float64), fs)
        rir = dsp.ImpulseResponse(None, h.copy(), fs, constrain_amplitude=False)
        o = dsp.room_acoustics.convolve_rir_on_signal(sig, rir, keep_peak_level=kp, keep_length=kl)
        arrs[f"h_{i}"] = np.asarray(rir.time_data[:, 0])
        arrs[f"y_{i}"] = o.time_data.astype(np.float32)
        cases.append(dict(taps=T, keep_peak_level=kp, keep_length=kl, note="x / y stored as float32"))
    save("rir", dict(cases=cases, fs=fs), arrs)


def gen_das(dsp):
    """BeamformerDASFrequency.get_beamformer_map (beamforming/beamforming.py:799-880): the
    quadratic forms h^H CSM h per grid point and bin, diagonal removal, clipping and Simpson
    integration.  Stored: the selected bins, the raw CSM slice, the steering vectors the
    reference built, and its final maps."""
    from dsptoolbox.helpers.other import (_get_fractional_octave_bandwidth,
                                          find_nearest_points_index_in_vector)
    fs = 10_000
    rng = np.random.default_rng(29)
    n_mics = 12
    pts = dict(x=rng.uniform(-0.3, 0.3, n_mics), y=rng.uniform(-0.3, 0.3, n_mics), z=np.zeros(n_mics))
    ma = dsp.beamforming.MicArray(pts)
    src = dsp.Signal(None, rng.standard_normal(20_000) * 0.3, fs)
    s = dsp.beamforming.MonopoleSource(src, [0.05, 0.3, 0.5]).get_signals_on_array(ma)
    s.set_spectrum_parameters(window_length_samples=512)
    g = dsp.beamforming.Regular2DGrid(np.arange(-0.3, 0.3, 0.05), np.arange(-0.5, 0.5, 0.05), ["x", "y"],
                                      value3=0.5)
    cases, arrs = [], {}
    f_all, csm_all = s.get_csm()
    for i, (form, fc, frac, rm) in enumerate((("TrueLocation", 2000.0, 3, True), ("Classic", 1500.0, 0, True),
                                              ("TruePower", 2500.0, 4, False), ("Inverse", 1200.0, 6, True))):
        st = dsp.beamforming.SteeringVector(formulation=dsp.beamforming.SteeringVectorType[form])
        bf = dsp.beamforming.BeamformerDASFrequency(s, ma, g, st)
        m = bf.get_beamformer_map(fc, frac, remove_csm_diagonal=rm)
        ids = find_nearest_points_index_in_vector(_get_fractional_octave_bandwidth(fc, frac), f_all)
        id1, id2 = int(ids[0]), int(ids[1])
        if id1 == id2:
            id2 += 1
        f = f_all[id1:id2]
        h = st.get_vector(f * np.pi * 2 / bf.c, grid=g, mic=ma)
        arrs[f"f_{i}"] = f
        arrs[f"csm_{i}"] = csm_all[id1:id2]
        arrs[f"h_{i}"] = h
        arrs[f"map_{i}"] = m
        cases.append(dict(formulation=form, center_hz=fc, octave_fraction=frac, remove_csm_diagonal=rm,
                          bins=[id1, id2], grid_shape=list(m.shape)))
    save("das", dict(cases=cases, fs=fs, n_mics=n_mics), arrs)


def gen_das_signal(dsp):
    """BeamformerDASFrequency end to end FROM THE MICROPHONE SIGNALS (beamforming.py:799-880): the signals
    (float32 values), the steering vectors the reference built for the selected bins, the final maps --
    for the device chain Signal.get_csm(on_device=True) -> diagonal treatment -> map."""
    from dsptoolbox.helpers.other import (_get_fractional_octave_bandwidth,
                                          find_nearest_points_index_in_vector)
    fs = 10_000
    rng = np.random.default_rng(31)
    n_mics = 12
    pts = dict(x=rng.uniform(-0.3, 0.3, n_mics), y=rng.uniform(-0.3, 0.3, n_mics), z=np.zeros(n_mics))
    ma = dsp.beamforming.MicArray(pts)
    src = dsp.Signal(None, rng.standard_normal(20_000) * 0.3, fs)
    s0 = dsp.beamforming.MonopoleSource(src, [0.05, 0.3, 0.5]).get_signals_on_array(ma)
    td = s0.time_data.astype(np.float32)
    s = dsp.Signal(None, td.astype(np.float64), fs)
    s.set_spectrum_parameters(window_length_samples=512)
    g = dsp.beamforming.Regular2DGrid(np.arange(-0.3, 0.3, 0.05), np.arange(-0.5, 0.5, 0.05), ["x", "y"], value3=0.5)
    cases, arrs = [], {"time_data": td}
    f_all, _ = s.get_csm()
    for i, (form, fc, frac, rm) in enumerate((("TrueLocation", 2000.0, 3, True), ("Classic", 1500.0, 0, False))):
        st = dsp.beamforming.SteeringVector(formulation=dsp.beamforming.SteeringVectorType[form])
        bf = dsp.beamforming.BeamformerDASFrequency(s, ma, g, st)
        m = bf.get_beamformer_map(fc, frac, remove_csm_diagonal=rm)
        ids = find_nearest_points_index_in_vector(_get_fractional_octave_bandwidth(fc, frac), f_all)
        id1, id2 = int(ids[0]), int(ids[1])
        if id1 == id2:
            id2 += 1
        f = f_all[id1:id2]
        arrs[f"h_{i}"] = st.get_vector(f * np.pi * 2 / bf.c, grid=g, mic=ma).astype(np.complex64)
        arrs[f"map_{i}"] = m
        cases.append(dict(formulation=form, center_hz=fc, octave_fraction=frac, remove_csm_diagonal=rm,
                          bins=[id1, id2], grid_shape=list(m.shape), n_points=int(g.number_of_points)))
    save("das_signal", dict(cases=cases, fs=fs, n_mics=n_mics, window=512), arrs)


def gen_mel(dsp):
    """transforms.log_mel_spectrogram / mfcc / mel_filterbank (transforms/transforms.py:113-441),
    generate_plot=False."""
    from dsptoolbox.standard.enums import SpectrumScaling
    fs = 16000
    rng = np.random.default_rng(33)
    n = 12000
    t = np.arange(n) / fs
    x = np.stack([0.3 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 440 * t),
                  0.1 * rng.standard_normal(n) * (1 + np.sin(2 * np.pi * 3 * t))], axis=1)
    cases, arrs = [], {"x": x}
    for i, (W, nfft, sc, rng_hz, nb) in enumerate(((512, None, "FFTBackward", None, 40),
                                                   (256, None, "AmplitudeSpectrum", [100.0, 6000.0], 24),
                                                   (512, None, "PowerSpectralDensity", None, 32))):
        s = dsp.Signal(None, x.copy(), fs)
        s.set_spectrogram_parameters(window_length_samples=W, fft_length_samples=nfft,
                                     scaling=SpectrumScaling[sc])
        t_, f_mel, lm = dsp.transforms.log_mel_spectrogram(s, range_hz=rng_hz, n_bands=nb, generate_plot=False)
        arrs[f"t_{i}"], arrs[f"fmel_{i}"], arrs[f"logmel_{i}"] = t_, f_mel, lm
        t2, f_mel2, mf = dsp.transforms.mfcc(s, generate_plot=False)
        arrs[f"fmel2_{i}"], arrs[f"mfcc_{i}"] = f_mel2, mf
        _, f_hz, _ = s.get_spectrogram()
        mfilt, _ = dsp.transforms.mel_filterbank(f_hz, rng_hz, nb, normalize=True)
        arrs[f"mfilt_{i}"] = mfilt
        cases.append(dict(W=W, nfft=nfft, scaling=sc, range_hz=rng_hz, n_bands=nb))
    save("mel", dict(cases=cases, fs=fs), arrs)


def gen_chroma(dsp):
    """transforms.chroma_stft (transforms/transforms.py:589-684), no plot."""
    from dsptoolbox.standard.enums import SpectrumScaling
    fs = 22050
    rng = np.random.default_rng(44)
    n = 15000
    t = np.arange(n) / fs
    x = np.stack([0.05 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 440 * t) + 0.3 * np.sin(2 * np.pi * 660 * t),
                  0.2 * rng.standard_normal(n) + 0.4 * np.sin(2 * np.pi * 261.63 * t * (1 + 0.2 * t))], axis=1)
    cases, arrs = [], {"x": x}
    for i, (W, ov, pad, sc, tune, comp) in enumerate(((1024, 50, True, "FFTBackward", 440, 0.5),
                                                      (2048, 75, False, "AmplitudeSpectrum", 442.0, 2.0),
                                                      (1024, 50, True, "PowerSpectralDensity", 440, 10.0))):
        s = dsp.Signal(None, x.copy(), fs)
        s.set_spectrogram_parameters(window_length_samples=W, overlap_percent=ov, padding=pad,
                                     scaling=SpectrumScaling[sc])
        t_, chroma, pitch = dsp.transforms.chroma_stft(s, tuning_a_hz=tune, compression=comp)
        arrs[f"t_{i}"], arrs[f"chroma_{i}"], arrs[f"pitch_{i}"] = t_, chroma, pitch
        cases.append(dict(W=W, ov=ov, pad=pad, scaling=sc, tuning=tune, compression=comp))
    save("chroma", dict(cases=cases, fs=fs), arrs)


def gen_fir_stream(dsp):
    """Block-streaming FIR classes (classes/fir_filter_realtime.py:75-335) driven as the
    reference's tests do (tests/test_classes.py:1527-1580): seeded noise through a decaying
    random impulse response, block by block."""
    rng = np.random.default_rng(55)
    fs = 48000
    cases, arrs = [], {}
    for i, (T, bs, n, n_ch) in enumerate(((700, 512, 5000, 1), (3840, 256, 6000, 1), (1500, 512, 4096, 2),
                                           (512, 512, 2048, 3), (1536, 512, 5000, 1))):
        fir = rng.standard_normal((T, n_ch)) * np.exp(-np.arange(T) / (T / 6))[:, None]
        n_blocks = n // bs + 1
        x = np.zeros((n_blocks * bs, n_ch))
        x[:n] = rng.standard_normal((n, n_ch)) * 0.3
        arrs[f"fir_{i}"], arrs[f"x_{i}"] = fir, x
        outs = {}
        for name, cls in (("ols", dsp.filterbanks.FIRFilterOverlapSave),
                          ("upart", dsp.filterbanks.FIRUniformPartitioned)):
            f = cls(fir[:, 0].copy())
            f.prepare(bs, n_ch)
            acc = np.zeros_like(x)
            for b in range(n_blocks):
                sl = slice(b * bs, (b + 1) * bs)
                for ch in range(n_ch):
                    acc[sl, ch] = f.process_block(x[sl, ch], ch)
            outs[name] = acc
        f = dsp.filterbanks.FIRUniformPartitionedMultichannel(fir.copy())
        f.prepare(bs)
        acc = np.zeros_like(x)
        for b in range(n_blocks):
            sl = slice(b * bs, (b + 1) * bs)
            acc[sl, :] = f.process_block(x[sl, :])
        outs["multi"] = acc
        for k, v in outs.items():
            arrs[f"{k}_{i}"] = v
        cases.append(dict(T=T, blocksize=bs, n=n, n_ch=n_ch))
    save("fir_stream", dict(cases=cases, fs=fs), arrs)


def gen_deconv_nonfast(dsp):
    """spectral_deconvolve for signal lengths that are NOT fast FFT lengths: get_spectrum pads to
    next_fast_len(N) but _spectral_deconvolve inverts with np.fft.irfft(..., n=N)
    (transfer_functions/_transfer_functions.py:37-41), i.e. numpy crops the spectrum to N//2 + 1
    bins before an N-point inverse."""
    fs = 48000
    rng = np.random.default_rng(66)
    cases, arrs = [], {}
    i = 0
    for n, c, reg, pad, keep in ((5918, 2, True, False, False), (5918, 2, True, True, True), (26075, 1, True, True, False),
                                 (5940, 3, False, False, False), (12345, 2, True, False, False)):
        t = np.arange(n) / fs
        if reg:
            x = (0.5 * np.sin(2 * np.pi * (20 * t + (8000 - 20) / (2 * t[-1]) * t * t)))[:, None]
        else:
            x = rng.standard_normal((n, 1)) * 0.3
        h = rng.standard_normal((200, c)) * np.exp(-np.arange(200) / 30.0)[:, None]
        y = np.stack([np.convolve(x[:, 0], h[:, j])[:n] for j in range(c)], axis=1) + 1e-3 * rng.standard_normal((n, c))
        ir = dsp.transfer_functions.spectral_deconvolve(dsp.Signal(None, y.copy(), fs), dsp.Signal(None, x.copy(), fs),
                                                        apply_regularization=reg, padding=pad,
                                                        keep_original_length=keep)
        arrs[f"x_{i}"], arrs[f"y_{i}"], arrs[f"ir_{i}"] = x, y, ir.time_data
        cases.append(dict(n=n, n_ch=c, regularized=reg, padding=pad, keep_original_length=keep))
        i += 1
    save("deconv_nonfast", dict(cases=cases, fs=fs), arrs)


def gen_welch_long(dsp):
    """Long windows (2048 ... 16384 samples) through the reference's _welch (auto spectra, cross spectra
    of channel pairs) and compute_transfer_function with one input channel per output channel and
    with one for all -- the shapes the register kernels of round 2 cover.  Inputs are stored as
    float32 values (cast to float64 before the reference runs) to keep the fixture small."""
    from dsptoolbox.standard._spectral_methods import _welch
    from dsptoolbox.standard.enums import SpectrumScaling as S, Window
    from dsptoolbox.transfer_functions.enums import TransferFunctionType
    fs = 48000
    rng = np.random.default_rng(2048)
    n = 16384 * 3 + 100
    x = (rng.standard_normal((n, 3)) * 0.3).astype(np.float32).astype(np.float64)
    h = rng.standard_normal((3, 16)) * np.exp(-np.arange(16) / 5.0)
    y = np.stack([np.convolve(x[:, c], h[c])[:n] for c in range(3)], axis=1) + 0.05 * rng.standard_normal((n, 3))
    y1 = np.stack([np.convolve(x[:, 0], h[c])[:n] for c in range(3)], axis=1) + 0.05 * rng.standard_normal((n, 3))
    y, y1 = y.astype(np.float32).astype(np.float64), y1.astype(np.float32).astype(np.float64)
    cases, arrs = [], {"x": x.astype(np.float32), "y_multi": y.astype(np.float32), "y_single": y1.astype(np.float32)}
    i = 0

    def some(a, W):  # the edge bins and every 5th one (5 is coprime to the 2 / 4 sub-spectrum classes)
        nb = W // 2 + 1
        idx = np.unique(np.r_[0:8, 0:nb:5, nb - 8:nb])
        return idx, np.asarray(a)[idx]
    for W, ov, det, sc in ((2048, 50, True, S.FFTBackward), (8192, 50, False, S.PowerSpectralDensity),
                           (8192, 75, True, S.AmplitudeSpectrum), (16384, 50, True, S.FFTBackward),
                           (16384, 25, False, S.PowerSpectrum)):
        auto = _welch(y.copy(), None, fs, Window.Hann, W, ov, det, "mean", sc)
        cross = _welch(x.copy(), y.copy(), fs, Window.Hann, W, ov, det, "mean", sc)
        arrs[f"bins_{i}"], arrs[f"auto_{i}"] = some(auto, W)
        arrs[f"cross_{i}"] = some(cross, W)[1]
        case = dict(W=W, overlap=ov, detrend=det, scaling=sc.name, tf=[])
        for mode in TransferFunctionType:
            for single in (True, False):
                inp = dsp.Signal(None, x[:, :1].copy() if single else x.copy(), fs)
                out = dsp.Signal(None, (y1 if single else y).copy(), fs)
                inp.set_spectrum_parameters(window_length_samples=W, window_type=Window.Hann, overlap_percent=ov,
                                            detrend=det, average="mean", scaling=sc)
                sp = dsp.transfer_functions.compute_transfer_function(out, inp, W, mode)
                key = f"{i}_{mode.name}_{'single' if single else 'multi'}"
                arrs["tf_" + key] = some(sp.spectral_data, W)[1].astype(np.complex64)
                arrs["coh_" + key] = some(sp.coherence, W)[1].astype(np.float32)
                case["tf"].append(key)
        cases.append(case)
        i += 1
    save("welch_long", dict(cases=cases, fs=fs, note="outputs at bins_<i> only; tf / coh stored as complex64 / float32 (compare at 1e-6)"), arrs)


def gen_welch4096(dsp):
    """compute_transfer_function with 4096-sample windows at 50 % overlap on noise inputs, 66 frames:
    the shape of the headline kernels (one input channel for all outputs -> welch4096::k_h1f / k_y3;
    one per output -> k_x3 + k_px_sum + k_y3) against the reference itself.
    3 output channels + the input(s); outputs at every 5th bin and the edges, float64."""
    from dsptoolbox.standard.enums import SpectrumScaling as S, Window
    from dsptoolbox.transfer_functions.enums import TransferFunctionType
    fs = 48000
    rng = np.random.default_rng(4096)
    n = 2048 * 66 - 333
    # inputs live on a 2^-13 grid (stored as int16, like the reference's own PCM examples): the
    # reference runs on exactly the values the fixture holds
    def grid(a):
        return np.clip(np.round(a * 8192.0), -32768, 32767).astype(np.int16)
    xq = grid(rng.standard_normal((n, 3)) * 0.3 + 0.02)
    x = xq.astype(np.float64) / 8192.0
    h = rng.standard_normal((3, 24)) * np.exp(-np.arange(24) / 6.0)
    h *= 0.5 / np.max(np.abs(h), axis=1, keepdims=True)
    yq = grid(np.stack([np.convolve(x[:, c], h[c])[:n] for c in range(3)], axis=1) + 0.05 * rng.standard_normal((n, 3)))
    y1q = grid(np.stack([np.convolve(x[:, 0], h[c])[:n] for c in range(3)], axis=1) + 0.05 * rng.standard_normal((n, 3)))
    y, y1 = yq.astype(np.float64) / 8192.0, y1q.astype(np.float64) / 8192.0
    W, nb = 4096, 2049
    bins = np.unique(np.r_[0:8, 0:nb:5, nb - 8:nb])
    cases, arrs = [], {"x_q13": xq, "y_multi_q13": yq, "y_single_q13": y1q, "bins": bins}
    for i, (det, sc) in enumerate(((True, S.FFTBackward), (False, S.PowerSpectralDensity), (True, S.AmplitudeSpectrum))):
        case = dict(W=W, overlap=50, detrend=det, scaling=sc.name, tf=[])
        for mode in TransferFunctionType:
            for single in (True, False):
                inp = dsp.Signal(None, x[:, :1].copy() if single else x.copy(), fs)
                out = dsp.Signal(None, (y1 if single else y).copy(), fs)
                inp.set_spectrum_parameters(window_length_samples=W, window_type=Window.Hann, overlap_percent=50,
                                            detrend=det, average="mean", scaling=sc)
                sp = dsp.transfer_functions.compute_transfer_function(out, inp, W, mode)
                key = f"{i}_{mode.name}_{'single' if single else 'multi'}"
                arrs["tf_" + key] = np.asarray(sp.spectral_data)[bins]
                arrs["coh_" + key] = np.asarray(sp.coherence)[bins]
                case["tf"].append(key)
        cases.append(case)
    save("welch4096", dict(cases=cases, fs=fs, frames=66, note="inputs = int16 / 8192; outputs at `bins` only, float64 / complex128"), arrs)


def gen_deconv_scaled(dsp):
    """spectral_deconvolve on signals whose spectrum parameters carry a scaling other than the plain
    transform: only the method is forced to FFT (transfer_functions.py:142-143), the scaling still applies
    in get_spectrum (classes/signal.py:899-938).  Norms, amplitude and power scalings, with and without
    regularisation / padding, one denominator for all channels and one per channel."""
    from dsptoolbox.standard.enums import SpectrumScaling as S
    fs = 48000
    rng = np.random.default_rng(77)
    n = 6000
    t = np.arange(n) / fs
    x = (0.5 * np.sin(2 * np.pi * (30 * t + (9000 - 30) / (2 * t[-1]) * t * t)))[:, None]
    h = rng.standard_normal((64, 2)) * np.exp(-np.arange(64) / 12.0)[:, None]
    y = np.stack([np.convolve(x[:, 0], h[:, j])[:n] for j in range(2)], axis=1) + 1e-3 * rng.standard_normal((n, 2))
    xn = rng.standard_normal((n, 1)) * 0.3   # white denominator for the unregularised cases
    yn = np.stack([np.convolve(xn[:, 0], h[:, j])[:n] for j in range(2)], axis=1)
    cases, arrs = [], {"x": x, "y": y, "xn": xn, "yn": yn}
    combos = [
        (S.FFTForward, S.FFTForward, True, False, False, False),
        (S.FFTOrthogonal, S.FFTBackward, True, True, True, False),
        (S.AmplitudeSpectrum, S.AmplitudeSpectrum, True, False, False, True),
        (S.AmplitudeSpectralDensity, S.FFTForward, False, False, False, False),
        (S.PowerSpectralDensity, S.PowerSpectralDensity, True, False, False, False),
        (S.PowerSpectrum, S.AmplitudeSpectrum, True, True, False, True),
    ]
    for i, (sy, sx, reg, pad, keep, per_ch) in enumerate(combos):
        xi, yi = (x, y) if reg else (xn, yn)
        xin = np.repeat(xi, 2, axis=1) * np.array([1.0, 0.8]) if per_ch else xi
        out = dsp.Signal(None, yi.copy(), fs)
        inp = dsp.Signal(None, xin.copy(), fs)
        out.set_spectrum_parameters(method=dsp.SpectrumMethod.FFT, scaling=sy)
        inp.set_spectrum_parameters(method=dsp.SpectrumMethod.FFT, scaling=sx)
        ir = dsp.transfer_functions.spectral_deconvolve(out, inp, apply_regularization=reg, padding=pad,
                                                        keep_original_length=keep)
        arrs[f"ir_{i}"] = ir.time_data
        cases.append(dict(scaling_y=sy.name, scaling_x=sx.name, regularized=reg, padding=pad, keep=keep,
                          per_channel=per_ch))
    save("deconv_scaled", dict(cases=cases, fs=fs), arrs)


def gen_fir_complex(dsp):
    """Filter.filter_signal with COMPLEX taps (a one-sided band pass: real prototype times exp(j w n)): the
    reference keeps the imaginary part of the output in Signal.time_data_imaginary
    (classes/filter_helpers.py:364-371).  Plain call, a channel subset, and with filter state."""
    import warnings
    fs = 48000
    rng = np.random.default_rng(364)
    n = 9000
    x = rng.standard_normal((n, 3)) * 0.3
    cases, arrs = [], {"x": x}
    from scipy.signal import firwin
    for i, (T, channels, zi) in enumerate(((301, None, False), (2500, [0, 2], False), (129, None, True))):
        proto = firwin(T, 3000.0, fs=fs)
        b = proto * np.exp(1j * 2 * np.pi * 6000.0 / fs * np.arange(T))
        f = dsp.Filter.from_ba(b, [1.0], fs)
        s = dsp.Signal(None, x.copy(), fs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if zi:
                f.initialize_zi(3)
            out = f.filter_signal(s, channels=channels, activate_zi=zi)
        arrs[f"b_{i}"] = b
        arrs[f"re_{i}"] = out.time_data
        arrs[f"im_{i}"] = out.time_data_imaginary
        cases.append(dict(taps=T, channels=channels, zi=zi))
    save("fir_complex", dict(cases=cases, fs=fs), arrs)


def gen_csm_coherent(dsp):
    """Cross-spectral matrices of coherent channels (one source through responses of either sign): at DC
    and Nyquist the cross spectra are real and some are negative, where the amplitude scalings take the
    square root on its branch cut -- the reference's convention for those elements (and even channel
    counts, which the csm fixture of round 1 does not have)."""
    from dsptoolbox.standard.enums import SpectrumScaling as S, SpectrumMethod
    fs = 48000
    rng = np.random.default_rng(77)
    n = 7665
    src = rng.standard_normal(n) * 0.3 + 0.05
    h = rng.standard_normal((32, 8)) * np.exp(-np.arange(32) / 6.0)[:, None]
    x = np.stack([np.convolve(src, h[:, c])[:n] for c in range(8)], axis=1) + 0.05 * rng.standard_normal((n, 8))
    x = x.astype(np.float32).astype(np.float64)
    cases, arrs = [], {"x": x.astype(np.float32)}
    for i, (C, W, ov, det, sc) in enumerate(((8, 256, 50, False, S.FFTBackward), (8, 256, 50, True, S.AmplitudeSpectrum),
                                             (6, 512, 75, False, S.AmplitudeSpectralDensity),
                                             (8, 256, 50, False, S.PowerSpectralDensity))):
        s_ = dsp.Signal(None, x[:, :C].copy(), fs)
        s_.set_spectrum_parameters(method=SpectrumMethod.WelchPeriodogram, window_length_samples=W,
                                   overlap_percent=ov, detrend=det, scaling=sc)
        f, csm = s_.get_csm()
        cases.append(dict(n_ch=C, W=W, overlap=ov, detrend=det, scaling=sc.name))
        arrs[f"csm_{i}"] = csm
    # 64 channels = the 8 stored ones through a stored 8 x 64 mixing matrix (the input itself would be
    # 2 MB); every 8th bin and the edge bins, as complex64
    mix = rng.standard_normal((8, 64)).astype(np.float32).astype(np.float64)
    x64 = x @ mix
    s_ = dsp.Signal(None, x64.copy(), fs)
    s_.set_spectrum_parameters(method=SpectrumMethod.WelchPeriodogram, window_length_samples=256,
                               overlap_percent=50, detrend=False, scaling=S.FFTBackward)
    f, csm = s_.get_csm()
    bins = np.unique(np.r_[0:3, 0:129:8, 126:129])
    arrs["mix"], arrs["bins64"], arrs["csm64"] = mix.astype(np.float32), bins, csm[bins].astype(np.complex64)
    cases.append(dict(n_ch=64, W=256, overlap=50, detrend=False, scaling="FFTBackward", data="x @ mix, bins64 only"))
    save("csm_coherent", dict(cases=cases, fs=fs), arrs)


def gen_api_rest(dsp):
    """The FIR-side API around the hot path: Filter.get_ir / get_transfer_function
    (classes/filter.py:818-900), FilterBank.get_ir / get_transfer_function / filter_multiband_signal /
    swap_filters (classes/filterbank.py:479-655), Signal.add_channel (classes/signal.py:776-852),
    MultiBandSignal.get_all_bands / swap_bands (classes/multibandsignal.py:378-410, 463-518)."""
    import warnings
    from dsptoolbox.standard.enums import FilterBankMode, FilterPassType
    warnings.simplefilter("ignore")
    fs = 48000
    rng = np.random.default_rng(77)
    cases, arrs = [], {}
    f1 = dsp.Filter.fir_filter(300, [200.0, 5000.0], FilterPassType.Bandpass, fs)
    f2 = dsp.Filter.fir_filter(150, 3000.0, FilterPassType.Lowpass, fs)
    f3 = dsp.Filter.fir_filter(64, 1000.0, FilterPassType.Highpass, fs)
    arrs["b1"], arrs["b2"], arrs["b3"] = f1.ba[0], f2.ba[0], f3.ba[0]
    # -- Filter.get_ir: longer than the taps, shorter (warning, adapted), zero phase
    for i, (L, zp) in enumerate([(1024, False), (100, False), (2048, True)]):
        ir = f1.get_ir(L, zero_phase=zp)
        assert type(ir).__name__ == "ImpulseResponse"
        arrs[f"ir_{i}"] = ir.time_data
        cases.append(dict(kind="filter_get_ir", length=L, zero_phase=zp, key=f"ir_{i}"))
    # -- Filter.get_transfer_function: a uniform rfft grid and an arbitrary (log-spaced) vector
    fv_u = np.fft.rfftfreq(4096, 1 / fs)
    fv_l = np.logspace(np.log10(20.0), np.log10(23999.0), 500)
    arrs["fv_uniform"], arrs["fv_log"] = fv_u, fv_l
    arrs["h_uniform"] = f1.get_transfer_function(fv_u)
    arrs["h_log"] = f1.get_transfer_function(fv_l)
    cases.append(dict(kind="filter_get_tf", keys=["h_uniform", "h_log"]))
    # -- FilterBank
    fb = dsp.FilterBank([f1, f2, f3])
    for mode in FilterBankMode:
        o = fb.get_ir(1000, mode)
        if mode == FilterBankMode.Parallel:
            assert type(o).__name__ == "MultiBandSignal"
            arrs[f"bank_ir_{mode.name}"] = np.asarray(o.get_all_time_data()[0])
        else:
            arrs[f"bank_ir_{mode.name}"] = o.time_data
        arrs[f"bank_h_{mode.name}"] = fb.get_transfer_function(fv_l, mode)
        cases.append(dict(kind="bank_get_ir_tf", mode=mode.name, length=1000, out_type=type(o).__name__))
    o = fb.get_ir(100, FilterBankMode.Summed)  # shorter than the longest filter: order + 100
    arrs["bank_ir_short"] = o.time_data
    o = fb.get_ir(2000, FilterBankMode.Parallel, zero_phase=True)
    arrs["bank_ir_zero_phase"] = np.asarray(o.get_all_time_data()[0])
    cases.append(dict(kind="bank_get_ir_edge", keys=["bank_ir_short", "bank_ir_zero_phase"]))
    # -- filter_multiband_signal: a 3-band, 2-channel multiband signal through the bank (band n by filter n)
    bands = [rng.standard_normal((3000, 2)) * 0.2 for _ in range(3)]
    for n, bnd in enumerate(bands):
        arrs[f"mb_in_{n}"] = bnd
    mb = dsp.MultiBandSignal([dsp.Signal(None, bnd.copy(), fs) for bnd in bands])
    out = fb.filter_multiband_signal(mb)
    arrs["mb_out"] = np.asarray(out.get_all_time_data()[0])
    out_zp = fb.filter_multiband_signal(mb, zero_phase=True)
    arrs["mb_out_zero_phase"] = np.asarray(out_zp.get_all_time_data()[0])
    cases.append(dict(kind="filter_multiband_signal", shape=list(arrs["mb_out"].shape)))
    # -- MultiBandSignal.get_all_bands / swap_bands, FilterBank.swap_filters
    arrs["all_bands_ch1"] = out.get_all_bands(channel=1).time_data
    out.swap_bands([2, 0, 1])
    arrs["swapped_all_bands_ch0"] = out.get_all_bands(channel=0).time_data
    fb2 = dsp.FilterBank([f1, f2, f3])
    fb2.swap_filters([1, 2, 0])
    arrs["swapped_first_taps"] = fb2.filters[0].ba[0]
    cases.append(dict(kind="bands_and_filters_reordered", new_band_order=[2, 0, 1], new_filter_order=[1, 2, 0]))
    # -- Signal.add_channel: same length, shorter (padded), longer (trimmed), flat vector
    base = rng.standard_normal((1000, 2)) * 0.3
    arrs["sig_base"] = base
    for i, extra in enumerate([rng.standard_normal((1000, 1)) * 0.3, rng.standard_normal((700, 2)) * 0.3,
                               rng.standard_normal((1300, 1)) * 0.3, rng.standard_normal(1000) * 0.3]):
        sgl = dsp.Signal(None, base.copy(), fs)
        sgl.add_channel(None, extra.copy(), fs)
        arrs[f"add_in_{i}"] = extra
        arrs[f"add_out_{i}"] = sgl.time_data
        cases.append(dict(kind="add_channel", key=f"add_out_{i}", channels=int(sgl.number_of_channels)))
    save("api_rest", dict(cases=cases, fs=fs), arrs)


def gen_api_holes(dsp):
    """Round 5 (missing 4 and 5): Spectrum.sum_channels(power_sum) (classes/spectrum.py:435-459),
    MultiBandSignal.is_complex_signal (classes/multibandsignal.py:262-274), ImpulseResponse.set_window
    (classes/impulse_response.py:139-152), and a SHORT cross-spectral matrix of more than 64 channels
    (_spectral_methods.py:285-371 through Signal.get_csm: 70 channels, 22 frames), mean and median averaging."""
    import warnings
    from dsptoolbox.standard.enums import SpectrumScaling, SpectrumMethod
    warnings.simplefilter("ignore")
    fs = 48000
    rng = np.random.default_rng(505)
    cases, arrs = [], {}
    f = np.fft.rfftfreq(256, 1 / fs)
    cplx = rng.standard_normal((129, 3)) + 1j * rng.standard_normal((129, 3))
    mag = np.abs(rng.standard_normal((129, 4)))
    arrs["freqs"], arrs["spec_complex"], arrs["spec_magnitude"] = f, cplx, mag
    for name, data in (("complex", cplx), ("magnitude", mag)):
        sp = dsp.Spectrum(f, data.copy())
        for ps in (True, False):
            out = sp.sum_channels(power_sum=ps)
            assert type(out).__name__ == "Spectrum" and out.number_of_channels == 1
            arrs[f"sum_{name}_{int(ps)}"] = out.spectral_data
            cases.append(dict(kind="spectrum_sum_channels", data=name, power_sum=ps, key=f"sum_{name}_{int(ps)}"))
        arrs[f"sum_{name}_default"] = sp.sum_channels().spectral_data  # the default is the power sum
    # MultiBandSignal.is_complex_signal
    real_bands = [dsp.Signal(None, rng.standard_normal((400, 2)) * 0.1, fs) for _ in range(2)]
    cb = [rng.standard_normal((400, 2)) * 0.1 + 1j * rng.standard_normal((400, 2)) * 0.1 for _ in range(2)]
    cplx_bands = [dsp.Signal(None, b, fs) for b in cb]
    flags = dict(empty=bool(dsp.MultiBandSignal().is_complex_signal),
                 real=bool(dsp.MultiBandSignal(real_bands).is_complex_signal),
                 complex=bool(dsp.MultiBandSignal(cplx_bands).is_complex_signal))
    arrs["complex_band_0"] = cb[0]
    cases.append(dict(kind="multiband_is_complex", flags=flags))
    # ImpulseResponse.set_window: kept as given, returns the object, refuses another shape
    td = rng.standard_normal((300, 2)) * 0.1
    ir = dsp.ImpulseResponse(None, td.copy(), fs)
    w = np.abs(rng.standard_normal((300, 2)))
    back = ir.set_window(w)
    refused = False
    try:
        ir.set_window(w[:100])
    except AssertionError:
        refused = True
    arrs["ir_td"], arrs["ir_window"], arrs["ir_window_kept"] = td, w, ir.window
    cases.append(dict(kind="ir_set_window", returns_self=bool(back is ir), refuses_other_shape=refused))
    # a short cross-spectral matrix of 70 channels (float64 route of the product: <= 128 frames)
    n_ch, n, W = 70, 1400, 128
    x = 0.1 * rng.standard_normal((n, n_ch)) + 0.2 * rng.standard_normal(n)[:, None]
    sig = dsp.Signal(None, x.copy(), fs)
    for i, sc in enumerate((SpectrumScaling.FFTBackward, SpectrumScaling.AmplitudeSpectrum)):
        sig.set_spectrum_parameters(method=SpectrumMethod.WelchPeriodogram, window_length_samples=W, scaling=sc)
        fv, m = sig.get_csm()
        arrs[f"csm70_{i}"] = m[::8]  # every 8th bin (9 of 65): 0.7 MB instead of 5
        cases.append(dict(kind="csm_short_many_channels", channels=n_ch, W=W, scaling=sc.name, key=f"csm70_{i}",
                          bin_step=8, frames=int(np.ceil(n / (W // 2)))))
    # ... and with average="median" (every element the median of its pair's frames): 22 frames (the two middle
    # ranks are averaged) and 21 frames (the first 1344 samples)
    for i, (sc, n_used) in enumerate(((SpectrumScaling.FFTBackward, n), (SpectrumScaling.AmplitudeSpectrum, 1344))):
        sig = dsp.Signal(None, x[:n_used].copy(), fs)
        sig.set_spectrum_parameters(method=SpectrumMethod.WelchPeriodogram, window_length_samples=W, scaling=sc,
                                    average="median")
        fv, m = sig.get_csm()
        arrs[f"csm70_median_{i}"] = m[::16]  # every 16th bin (5 of 65)
        cases.append(dict(kind="csm_short_median", channels=n_ch, W=W, scaling=sc.name, key=f"csm70_median_{i}",
                          bin_step=16, samples=n_used, frames=int(np.ceil(n_used / (W // 2)))))
    arrs["csm70_x"] = x
    save("api_holes", dict(cases=cases, fs=fs), arrs)


def gen_core(dsp):
    """framing, welch, transfer_function, stft, csm, spectrum_fft, deconvolve, fir and chirp_pair."""
    from dsptoolbox.standard._spectral_methods import _welch
    from dsptoolbox.standard._framed_signal_representation import _get_framed_signal
    from dsptoolbox.helpers.other import _compute_number_frames
    from dsptoolbox.standard.enums import (SpectrumScaling, SpectrumMethod, Window,
                                           FilterBankMode, FilterPassType)
    from dsptoolbox.transfer_functions.enums import TransferFunctionType

    S = SpectrumScaling
    fs = 48000
    import warnings

    warnings.simplefilter("ignore")

    # ------------------------------------------------------------ framing
    cases, arrs = [], {}
    rng = np.random.default_rng(11)
    for i, (N, W, hop, keep) in enumerate([(1000, 64, 32, True), (1024, 64, 32, True),
                                           (1000, 64, 48, True), (1000, 64, 32, False),
                                           (130, 128, 64, True), (777, 256, 64, False)]):
        x = rng.standard_normal((N, 2))
        fr = _get_framed_signal(x.copy(), W, hop, keep)
        nf, pad = _compute_number_frames(W, hop, N, keep)
        cases.append(dict(N=N, W=W, hop=hop, keep=keep, n_frames=int(nf), pad=int(pad)))
        arrs[f"in_{i}"] = x
        arrs[f"out_{i}"] = fr
    save("framing", dict(cases=cases), arrs)

    # ------------------------------------------------------------ welch
    rng = np.random.default_rng(12)
    xw = rng.standard_normal((16384, 3)) * 0.3
    xw[:, 1] += 0.25  # DC offset exercises detrend
    xw[:, 2] = np.convolve(xw[:, 0], rng.standard_normal(32) * 0.3)[:16384] + 0.01 * xw[:, 2]
    xr = xw[:10000]  # ragged: N % hop != 0
    cases, arrs = [], {"x": xw}
    combos = []
    for sc in S:
        combos.append((1024, 50, True, "mean", sc, "hann", "full"))
    combos += [
        (256, 0, True, "mean", S.FFTBackward, "hann", "full"),
        (256, 75, False, "mean", S.PowerSpectralDensity, "hann", "full"),
        (256, 50, False, "mean", S.FFTBackward, "hamming", "full"),
        (1024, 33, True, "mean", S.AmplitudeSpectrum, "blackman", "ragged"),
        (1024, 50, True, "mean", S.FFTBackward, "hann", "ragged"),
        (512, 50, False, "mean", S.PowerSpectrum, "boxcar", "ragged"),
        (1024, 50, True, "median", S.FFTBackward, "hann", "full"),
        (256, 50, False, "median", S.PowerSpectralDensity, "hann", "ragged"),
        (4096, 50, True, "mean", S.FFTBackward, "hann", "full"),
        (8, 50, True, "mean", S.FFTBackward, "hann", "ragged"),
    ]
    wmap = {"hann": Window.Hann, "hamming": Window.Hamming, "blackman": Window.Blackman,
            "boxcar": Window.Boxcar}
    for i, (W, ov, det, avg, sc, win, which) in enumerate(combos):
        data = xw if which == "full" else xr
        auto = _welch(data.copy(), None, fs, wmap[win], W, ov, det, avg, sc)
        cross = _welch(data[:, 0].copy(), data[:, 2].copy(), fs, wmap[win], W, ov, det, avg, sc)
        cases.append(dict(W=W, overlap=ov, detrend=det, average=avg, scaling=sc.name,
                          window=win, data=which))
        arrs[f"auto_{i}"] = auto
        arrs[f"cross_{i}"] = cross
    save("welch", dict(cases=cases, fs=fs, ragged_len=10000,
                       cross="x[:,0] vs x[:,2]"), arrs)

    # ------------------------------------------------------------ transfer functions
    rng = np.random.default_rng(13)
    N = 8000
    xin = rng.standard_normal((N, 3)) * 0.2
    h = rng.standard_normal((3, 24)) * np.exp(-np.arange(24) / 6.0)
    yout = np.stack([np.convolve(xin[:, c], h[c])[:N] for c in range(3)], axis=1)
    yout += 0.02 * rng.standard_normal((N, 3))
    y1 = np.stack([np.convolve(xin[:, 0], h[c])[:N] for c in range(3)], axis=1)
    y1 += 0.02 * rng.standard_normal((N, 3))
    cases, arrs = [], {"x": xin, "y_multi": yout, "y_single": y1}
    i = 0
    for mode in TransferFunctionType:
        for sc in (S.FFTBackward, S.PowerSpectralDensity, S.AmplitudeSpectrum, S.PowerSpectrum):
            for single in (True, False):
                for det, ov, W in ((True, 50, 512), (False, 75, 256)):
                    inp = dsp.Signal(None, xin[:, :1].copy() if single else xin.copy(), fs)
                    out = dsp.Signal(None, (y1 if single else yout).copy(), fs)
                    inp.set_spectrum_parameters(window_length_samples=1024,
                                                window_type=Window.Hann, overlap_percent=ov,
                                                detrend=det, average="mean", scaling=sc)
                    sp = dsp.transfer_functions.compute_transfer_function(out, inp, W, mode)
                    cases.append(dict(mode=mode.name, scaling=sc.name, single_input=single,
                                      detrend=det, overlap=ov, W=W))
                    arrs[f"f_{i}"] = np.asarray(sp.frequency_vector_hz)
                    arrs[f"tf_{i}"] = np.asarray(sp.spectral_data)
                    arrs[f"coh_{i}"] = np.asarray(sp.coherence)
                    i += 1
    save("transfer_function", dict(cases=cases, fs=fs), arrs)

    # ------------------------------------------------------------ stft
    rng = np.random.default_rng(14)
    xs = rng.standard_normal((3000, 2)) * 0.4 + 0.1
    cases, arrs = [], {"x": xs}
    combos = [(256, 50, None, False, True, sc) for sc in S]
    combos += [
        (256, 50, None, False, False, S.FFTBackward),
        (256, 75, 512, False, True, S.FFTBackward),
        (256, 33, 128, True, True, S.FFTBackward),
        (128, 0, None, True, False, S.AmplitudeSpectralDensity),
        (1024, 50, 2048, False, True, S.PowerSpectrum),
        (16, 50, None, False, True, S.FFTBackward),
    ]
    for i, (W, ov, nfft, det, pad, sc) in enumerate(combos):
        s = dsp.Signal(None, xs.copy(), fs)
        s.set_spectrogram_parameters(window_length_samples=W, window_type=Window.Hann,
                                     overlap_percent=ov, fft_length_samples=nfft,
                                     detrend=det, padding=pad, scaling=sc)
        t, f, st = s.get_spectrogram()
        cases.append(dict(W=W, overlap=ov, fft_length=nfft, detrend=det, padding=pad,
                          scaling=sc.name))
        arrs[f"t_{i}"] = t
        arrs[f"f_{i}"] = f
        arrs[f"stft_{i}"] = st
    save("stft", dict(cases=cases, fs=fs), arrs)

    # ------------------------------------------------------------ csm
    rng = np.random.default_rng(15)
    common = rng.standard_normal(4096)
    xc = 0.1 * rng.standard_normal((4096, 5)) + 0.2 * common[:, None]
    cases, arrs = [], {"x": xc}
    combos = [("welch", 256, 50, True, sc) for sc in S]
    combos += [("welch", 512, 75, False, S.FFTBackward), ("welch", 64, 0, False, S.PowerSpectralDensity)]
    combos += [("fft", None, None, None, sc) for sc in (S.FFTBackward, S.PowerSpectralDensity,
                                                       S.AmplitudeSpectrum)]
    for i, (meth, W, ov, det, sc) in enumerate(combos):
        s = dsp.Signal(None, xc.copy() if meth == "welch" else xc[:1000, :3].copy(), fs)
        if meth == "welch":
            s.set_spectrum_parameters(method=SpectrumMethod.WelchPeriodogram,
                                      window_length_samples=W, overlap_percent=ov, detrend=det,
                                      scaling=sc)
        else:
            s.set_spectrum_parameters(method=SpectrumMethod.FFT, scaling=sc)
        f, csm = s.get_csm()
        cases.append(dict(method=meth, W=W, overlap=ov, detrend=det, scaling=sc.name,
                          data="x" if meth == "welch" else "x[:1000,:3]"))
        arrs[f"f_{i}"] = f
        arrs[f"csm_{i}"] = csm
    save("csm", dict(cases=cases, fs=fs), arrs)

    # ------------------------------------------------------------ whole-signal spectrum
    rng = np.random.default_rng(16)
    xf = rng.standard_normal((3000, 2)) * 0.3
    cases, arrs = [], {"x": xf}
    i = 0
    for sc in S:
        for fast in (True, False):
            s = dsp.Signal(None, xf[:2999].copy() if not fast else xf.copy(), fs)
            s.set_spectrum_parameters(method=SpectrumMethod.FFT, scaling=sc,
                                      pad_to_fast_length=fast)
            f, sp = s.get_spectrum()
            cases.append(dict(scaling=sc.name, pad_to_fast_length=fast,
                              n=2999 if not fast else 3000))
            arrs[f"f_{i}"] = f
            arrs[f"sp_{i}"] = sp
            i += 1
    save("spectrum_fft", dict(cases=cases, fs=fs), arrs)

    # ------------------------------------------------------------ deconvolution
    rng = np.random.default_rng(17)
    cases, arrs = [], {}

    def sweep(n):
        t = np.arange(n) / fs
        T = n / fs
        k = np.log(20000.0 / 20.0)
        return 0.5 * np.sin(2 * np.pi * 20.0 * T / k * (np.exp(t / T * k) - 1.0))

    for tag, n in (("p2", 4096), ("np2", 6000)):
        xsw = sweep(n)
        hh = rng.standard_normal((2, 64)) * np.exp(-np.arange(64) / 10.0)
        yy = np.stack([np.convolve(xsw, hh[c])[:n] for c in range(2)], axis=1)
        yy += 1e-3 * rng.standard_normal(yy.shape)
        x2 = np.stack([xsw, 0.7 * sweep(n) + 1e-3 * rng.standard_normal(n)], axis=1)
        arrs[f"x_{tag}"] = xsw[:, None]
        arrs[f"x2_{tag}"] = x2
        arrs[f"y_{tag}"] = yy
    variants = [
        dict(data="p2", den="mono", reg=True, ss=None, thr=-30.0, pad=False, keep=False),
        dict(data="p2", den="multi", reg=True, ss=None, thr=-30.0, pad=False, keep=False),
        dict(data="p2", den="mono", reg=True, ss=[100.0, 15000.0], thr=-30.0, pad=False, keep=False),
        dict(data="p2", den="mono", reg=True, ss=[50.0, 100.0, 12000.0, 18000.0], thr=-30.0,
             pad=False, keep=False),
        dict(data="p2", den="mono", reg=False, ss=None, thr=-30.0, pad=False, keep=False),
        dict(data="p2", den="mono", reg=True, ss=None, thr=-20.0, pad=True, keep=False),
        dict(data="p2", den="mono", reg=False, ss=None, thr=-30.0, pad=True, keep=True),
        dict(data="np2", den="mono", reg=True, ss=None, thr=-30.0, pad=False, keep=False),
        dict(data="np2", den="multi", reg=False, ss=None, thr=-30.0, pad=True, keep=True),
    ]
    for i, v in enumerate(variants):
        tag = v["data"]
        inp = dsp.Signal(None, (arrs[f"x_{tag}"] if v["den"] == "mono" else arrs[f"x2_{tag}"]).copy(), fs)
        out = dsp.Signal(None, arrs[f"y_{tag}"].copy(), fs)
        ir = dsp.transfer_functions.spectral_deconvolve(
            out, inp, apply_regularization=v["reg"], start_stop_hz=v["ss"],
            threshold_db=v["thr"], padding=v["pad"], keep_original_length=v["keep"])
        assert type(ir).__name__ == "ImpulseResponse"
        cases.append(v)
        arrs[f"ir_{i}"] = ir.time_data
    save("deconvolve", dict(cases=cases, fs=fs), arrs)

    # ------------------------------------------------------------ FIR / filter bank
    rng = np.random.default_rng(18)
    cases, arrs = [], {}
    xa = rng.standard_normal((4096, 2)) * 0.1
    xb = rng.standard_normal((30000, 2)) * 0.1
    arrs["x_a"], arrs["x_b"] = xa, xb
    i = 0
    for order, cutoff, ptype in ((150, 3000.0, FilterPassType.Lowpass),
                                 (4096, [500.0, 4000.0], FilterPassType.Bandpass)):
        flt = dsp.Filter.fir_filter(order, cutoff, ptype, fs)
        b = flt.ba[0]
        for which, data in (("a", xa), ("b", xb)):
            for ch in (None, [1]):
                if which == "b" and ((order == 150) == (ch is None)):
                    continue  # keep the fixture small
                s = dsp.Signal(None, data.copy(), fs)
                o = flt.filter_signal(s, channels=ch)
                cases.append(dict(kind="filter", order=order, data=which,
                                  channels=ch, taps_key=f"b_{order}"))
                arrs[f"b_{order}"] = b
                arrs[f"y_{i}"] = o.time_data
                i += 1
    # filter bank: three FIR band filters
    bank_taps = []
    flts = []
    for (lo, hi) in ((100.0, 3000.0), (800.0, 12000.0), (300.0, 6000.0)):  # overlapping pass bands
        f_ = dsp.Filter.fir_filter(300, [lo, hi], FilterPassType.Bandpass, fs)
        flts.append(f_)
        bank_taps.append(f_.ba[0])
    arrs["bank_taps"] = np.stack(bank_taps)
    fb = dsp.FilterBank(flts)
    for mode in FilterBankMode:
        s = dsp.Signal(None, xa.copy(), fs)
        o = fb.filter_signal(s, mode)
        if mode == FilterBankMode.Parallel:
            td = o.get_all_time_data()[0] if isinstance(o.get_all_time_data(), tuple) else o.get_all_time_data()
            arrs[f"y_{i}"] = np.asarray(td)
            shape_note = "get_all_time_data()"
        else:
            arrs[f"y_{i}"] = o.time_data
            shape_note = "time_data"
        cases.append(dict(kind="bank", mode=mode.name, data="a", note=shape_note,
                          out_type=type(o).__name__))
        i += 1
    save("fir", dict(cases=cases, fs=fs), arrs)

    # ------------------------------------------------------------ config 1: the example chirps
    # BASELINE.json configs[0]: chirp_stereo.wav vs chirp.wav (16-bit PCM, 192 000 samples, 48 kHz):
    # Welch H1 (nfft 4096, Hann, 50 %) and the regularised spectral deconvolution.
    from scipy.io import wavfile
    fsw, xi = wavfile.read(os.path.join(REF, "example_data", "chirp.wav"))
    _, yi = wavfile.read(os.path.join(REF, "example_data", "chirp_stereo.wav"))
    assert xi.dtype == np.int16 and yi.dtype == np.int16
    xs_, ys_ = xi.astype(np.float64) / 32768, yi.astype(np.float64) / 32768
    inp = dsp.Signal(None, xs_.copy(), int(fsw))
    out = dsp.Signal(None, ys_.copy(), int(fsw))
    inp.set_spectrum_parameters(window_length_samples=4096, window_type=Window.Hann,
                                overlap_percent=50, detrend=True, scaling=S.FFTBackward)
    sp = dsp.transfer_functions.compute_transfer_function(out, inp, 4096, TransferFunctionType.H1)
    ir = dsp.transfer_functions.spectral_deconvolve(out, inp)
    n = ir.time_data.shape[0]
    save("chirp_pair", dict(cases=[dict(name="chirp_stereo vs chirp", n=int(n), fs=int(fsw),
                                        ir_head=48000, ir_tail=4096)], fs=int(fsw)),
         dict(x_int16=xi, y_int16=yi, tf=np.asarray(sp.spectral_data), coh=np.asarray(sp.coherence),
              ir_head=ir.time_data[:48000], ir_tail=ir.time_data[-4096:],
              ir_peak=np.array([np.max(np.abs(ir.time_data))])))


# name -> (generator, fixtures it writes).  `python oracle/gen_golden.py` runs all of them;
# `--only <name> [<name> ...]` some.  tests/test_oracle_golden.py checks that the union of the
# fixture lists is exactly tests/golden/*.npz.
GENERATORS = {
    "core": (gen_core, ["framing", "welch", "transfer_function", "stft", "csm", "spectrum_fft", "deconvolve", "fir",
                        "chirp_pair"]),
    "welch_long": (gen_welch_long, ["welch_long"]),
    "welch4096": (gen_welch4096, ["welch4096"]),
    "deconv_scaled": (gen_deconv_scaled, ["deconv_scaled"]),
    "deconv_nonfast": (gen_deconv_nonfast, ["deconv_nonfast"]),
    "fir_complex": (gen_fir_complex, ["fir_complex"]),
    "fir_state": (gen_fir_state, ["fir_state"]),
    "fir_stream": (gen_fir_stream, ["fir_stream"]),
    "das": (gen_das, ["das"]),
    "das_signal": (gen_das_signal, ["das_signal"]),
    "csm_coherent": (gen_csm_coherent, ["csm_coherent"]),
    "stft_manych": (gen_stft_manych, ["stft_manych"]),
    "stft_long": (gen_stft_long, ["stft_long"]),
    "stft_anylen": (gen_stft_anylen, ["stft_anylen"]),
    "istft": (gen_istft, ["istft"]),
    "istft_anylen": (gen_istft_anylen, ["istft_anylen"]),
    "rir": (gen_rir, ["rir"]),
    "mel": (gen_mel, ["mel"]),
    "chroma": (gen_chroma, ["chroma"]),
    "api_rest": (gen_api_rest, ["api_rest"]),
    "api_holes": (gen_api_holes, ["api_holes"]),
}


def fixtures_written():
    return sorted(n for _, names in GENERATORS.values() for n in names)


def main(argv=None):
    """All fixtures, or `--only name ...` (also the older `--only-some-name` flags); `--out DIR` writes
    somewhere else than tests/golden (to compare a fresh run with the committed files)."""
    global OUT
    import warnings
    argv = list(sys.argv[1:] if argv is None else argv)
    if "--out" in argv:
        i = argv.index("--out")
        OUT = argv[i + 1]
        del argv[i:i + 2]
    want = []
    if "--only" in argv:
        want = [a for a in argv[argv.index("--only") + 1:] if not a.startswith("--")]
    want += [a[len("--only-"):].replace("-", "_") for a in argv if a.startswith("--only-")]
    unknown = [w for w in want if w not in GENERATORS]
    if unknown:
        raise SystemExit(f"unknown generator(s) {unknown}; known: {sorted(GENERATORS)}")
    dsp = import_reference()
    warnings.simplefilter("ignore")
    for name, (fn, _) in GENERATORS.items():
        if not want or name in want:
            fn(dsp)


if __name__ == "__main__":
    main()
