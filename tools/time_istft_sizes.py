"""Dev tool: ds_istft_dev on the spectrogram of 64 channels x 512 000 samples (device resident), 50 % overlap,
across window lengths: ms per call and bytes moved ((bins, frames, channels) complex64 in, samples out)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd._lib import DeviceBuffer, get_context  # noqa: E402

ctx = get_context()
n, n_ch = 512000, 64
sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096]
rng = np.random.default_rng(0)
for W in sizes:
    step = W // 2
    n_frames = n // step + 1
    B = W // 2 + 1
    total = (n_frames - 1) * step + W
    sp = (rng.standard_normal((B, n_frames, n_ch)) + 1j * rng.standard_normal((B, n_frames, n_ch))).astype(np.complex64)
    w = np.hanning(W + 1)[:-1].astype(np.float32)
    d_s = DeviceBuffer.from_array(ctx, sp)
    d_w = DeviceBuffer.from_array(ctx, w)
    d_o = DeviceBuffer(ctx, n_ch * total * 4)

    def stepf():
        ctx.check(ctx.lib.ds_istft_dev(ctx.handle, C.c_void_p(d_s.ptr), B, n_frames, n_ch, W, W, step, 0, n_frames,
                                       C.c_void_p(d_w.ptr), C.c_float(1.0), C.c_int64(total), C.c_void_p(d_o.ptr),
                                       C.c_int64(total)), "ds_istft_dev")
    for _ in range(3):
        stepf()
    ctx.sync()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        stepf()
    ctx.sync()
    ms = (time.perf_counter() - t0) / K * 1e3
    tot = sp.nbytes + n_ch * total * 4
    ctx.profile_enable(True)
    for _ in range(5):
        stepf()
    ctx.sync()
    print("   kernels (ms over 5 calls):", ctx.profile_report())
    ctx.profile_enable(False)
    print(f"W {W:5d}: {ms:7.3f} ms  frames {n_frames:6d}  {tot / 1e6:7.1f} MB  {tot / ms / 1e9:5.2f} TB/s", flush=True)
    for d in (d_s, d_w, d_o):
        d.free()
