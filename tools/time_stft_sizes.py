"""Dev tool: ds_stft_r2c_dev on 64 channels x 512 000 samples (device resident), 50 % overlap, across
window lengths: ms per call and bytes moved (input + (bins, frames, channels) complex64 output)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd._lib import DeviceBuffer, get_context  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402

ctx = get_context()
n, n_ch = 512000, 64
x = np.random.default_rng(0).standard_normal((n, n_ch))
sizes = sys.argv[1:] or ["256", "512", "1024", "2048", "4096"]  # "W" or "W:nfft"
for arg in sizes:
    W, nfft = (int(a) for a in arg.split(":")) if ":" in arg else (int(arg), None)
    pl = backend._stft_plan(x, 48000, W, Window.Hann, 50, nfft, True, SpectrumScaling.FFTBackward)
    d_x = DeviceBuffer.from_array(ctx, pl["xp"])
    d_w = DeviceBuffer.from_array(ctx, pl["w32"])
    nbytes_out = pl["B"] * pl["n_frames"] * pl["n_ch"] * 8
    d_s = DeviceBuffer(ctx, nbytes_out)

    def step():
        ctx.check(ctx.lib.ds_stft_r2c_dev(ctx.handle, C.c_void_p(d_x.ptr), pl["n"], pl["n_ch"], pl["n"], pl["W"],
                                          pl["hop"], pl["nfft"], pl["pad_front"], pl["n_frames"],
                                          C.c_void_p(d_w.ptr), 0, pl["scale"], pl["edge"], pl["power"],
                                          C.c_void_p(d_s.ptr)), "ds_stft_r2c_dev")
    for _ in range(3):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        step()
    ctx.sync()
    ms = (time.perf_counter() - t0) / K * 1e3
    tot = pl["xp"].nbytes + nbytes_out
    ctx.profile_enable(True)
    for _ in range(5):
        step()
    ctx.sync()
    print("   kernels (ms over 5 calls):", ctx.profile_report())
    ctx.profile_enable(False)
    print(f"W {arg:>13s}: {ms:7.3f} ms  frames {pl['n_frames']:6d}  {tot / 1e6:7.1f} MB  {tot / ms / 1e9:5.2f} TB/s", flush=True)
    for d in (d_x, d_w, d_s):
        d.free()
