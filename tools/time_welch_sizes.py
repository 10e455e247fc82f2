"""Dev tool: Welch H1 (1-channel input, 64 responses x 2^20 samples, device resident) across FFT
lengths: ms per call and the input rate.  python tools/time_welch_sizes.py [nfft ...]"""
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
n, n_cy, fs = 2**20, 64, 48000
rng = np.random.default_rng(0)
x = rng.standard_normal((n, 1)).astype(np.float32)
y = rng.standard_normal((n, n_cy)).astype(np.float32)
d_y = DeviceBuffer.from_array(ctx, backend._planar_f32(y))
d_x = DeviceBuffer.from_array(ctx, backend._planar_f32(x))
sizes = [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096, 8192, 16384]
for W in sizes:
    window = backend._window_array(Window.Hann, W)
    hop, n_frames = backend._welch_framing(n, W, 50, window)
    amp, ns, fac, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, fs, window)
    d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
    B = W // 2 + 1
    d_tf = DeviceBuffer(ctx, B * n_cy * 8)
    d_coh = DeviceBuffer(ctx, B * n_cy * 4)

    def step():
        ctx.check(ctx.lib.ds_welch_tf_dev(ctx.handle, C.c_void_p(d_x.ptr), 1, n, C.c_void_p(d_y.ptr), n_cy, n, n, W,
                                          hop, n_frames, C.c_void_p(d_w.ptr), 1, 0, 1, amp, ns, fac, phys,
                                          C.c_void_p(d_tf.ptr), C.c_void_p(d_coh.ptr)), "welch")
    for _ in range(300):  # past the clock ramp of an idle chip (profiles/r05_clock_ramp.txt)
        step()
    ctx.sync()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        step()
    ctx.sync()
    ms = (time.perf_counter() - t0) / K * 1e3
    ctx.profile_enable(True)
    for _ in range(5):
        step()
    ctx.sync()
    rep = ctx.profile_report()
    ctx.profile_enable(False)
    print("   kernels:", rep)
    print(f"nfft {W:6d}: {ms:8.3f} ms  {65 * n / ms / 1e6:8.1f} G samples/s  {65 * n * 4 / ms / 1e9:6.2f} TB/s", flush=True)
    for d in (d_w, d_tf, d_coh):
        d.free()
