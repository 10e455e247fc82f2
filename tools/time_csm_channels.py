"""Time the cross-spectral matrix for several channel counts (64 mics is the benchmark shape)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd import backend
from dsptoolbox_amd._lib import DeviceBuffer, get_context
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window
ctx = get_context()
n, W = 512000, 1024
window = backend._window_array(Window.Hann, W)
hop, n_frames = backend._welch_framing(n, W, 50, window)
amp, norm_scale, factor, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, 48000, window)
d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
for n_ch in (32, 62, 63, 64, 66, 96, 128):
    x = np.random.default_rng(1).standard_normal((n_ch, n)).astype(np.float32)
    d_x = DeviceBuffer.from_array(ctx, x)
    d_c = DeviceBuffer(ctx, (W // 2 + 1) * n_ch * n_ch * 8)
    def step():
        ctx.check(ctx.lib.ds_csm_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, W, hop, n_frames,
                                     C.c_void_p(d_w.ptr), 1, 0, amp, norm_scale, factor, phys, C.c_void_p(d_c.ptr)), "csm")
    for _ in range(3):
        step()
    ctx.sync()
    ctx.profile_enable(True); ctx.profile_report(); step(); rep = ctx.profile_report(); ctx.profile_enable(False)
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    ctx.sync()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{n_ch:4d} channels: {ms:.3f} ms/step  kernels {({k: round(v[0], 3) for k, v in rep.items()})}", flush=True)
    d_x.free(); d_c.free()
