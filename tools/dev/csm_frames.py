"""Dev tool (GPU): does the spectrogram survive in the 256 MB Infinity Cache between the transform and
the Gram kernel?  Times both kernels of ds_csm_dev (64 channels, 1024-sample windows) for frame counts
whose spectrograms are 33 ... 525 MB: if the Gram kernel's time per frame is flat, slabs buy nothing."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dsptoolbox_amd import backend
from dsptoolbox_amd._lib import DeviceBuffer, get_context
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window
ctx = get_context()
W, n_ch = 1024, 64
window = backend._window_array(Window.Hann, W)
amp, norm_scale, factor, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, 48000, window)
d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
for frames in (125, 250, 500, 1000, 2000):
    n = frames * 512
    hop, n_frames = backend._welch_framing(n, W, 50, window)
    x = np.random.default_rng(1).standard_normal((n_ch, n)).astype(np.float32)
    d_x = DeviceBuffer.from_array(ctx, x)
    d_c = DeviceBuffer(ctx, (W // 2 + 1) * n_ch * n_ch * 8)
    def step():
        ctx.check(ctx.lib.ds_csm_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, W, hop, n_frames,
                                     C.c_void_p(d_w.ptr), 1, 0, amp, norm_scale, factor, phys, C.c_void_p(d_c.ptr)), "csm")
    for _ in range(5):
        step()
    ctx.sync()
    acc = {}
    ctx.profile_enable(True); ctx.profile_report()
    for _ in range(10):
        step()
    rep = ctx.profile_report(); ctx.profile_enable(False)
    mb = (W // 2 + 1) * n_frames * n_ch * 8 / 1e6
    print(f"{n_frames:5d} frames, spectrogram {mb:6.1f} MB: " + ", ".join(f"{k} {v[0] / v[1] * 1e3:7.1f} us ({v[0] / v[1] * 1e3 / n_frames * 1e3:6.1f} ns/frame)" for k, v in rep.items()), flush=True)
    d_x.free(); d_c.free()
