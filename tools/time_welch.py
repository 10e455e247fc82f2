"""Dev tool: per-phase cycle counts of the radix-16 Welch kernel (library built with -DW4_TIMING=1):
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC -DW4_TIMING=1 \
        -o tools/dbg/libdbg.so dsptoolbox_amd/csrc/api.hip -ldl
  DSPTOOLBOX_AMD_LIB=tools/dbg/libdbg.so DSPTOOLBOX_AMD_WELCH_VARIANT=r16 python tools/time_welch.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd._lib import DeviceBuffer, get_context  # noqa: E402
from dsptoolbox_amd.generators import sweep_and_responses  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402

ctx = get_context()
n, n_cy, W, fs = 2**20, 64, 4096, 48000
x, y = sweep_and_responses(n, n_cy, fs)
window = backend._window_array(Window.Hann, W)
hop, n_frames = backend._welch_framing(n, W, 50, window)
amp, ns, fac, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, fs, window)
d_y = DeviceBuffer.from_array(ctx, backend._planar_f32(y))
d_x = DeviceBuffer.from_array(ctx, backend._planar_f32(x))
d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
B = W // 2 + 1
d_tf = DeviceBuffer(ctx, B * n_cy * 8)
d_coh = DeviceBuffer(ctx, B * n_cy * 4)
lib = C.CDLL(os.environ["DSPTOOLBOX_AMD_LIB"])
out = (C.c_ulonglong * 16)()


def step():
    ctx.check(ctx.lib.ds_welch_tf_dev(ctx.handle, C.c_void_p(d_x.ptr), 1, n, C.c_void_p(d_y.ptr), n_cy, n, n, W,
                                      hop, n_frames, C.c_void_p(d_w.ptr), 1, 0, 1, amp, ns, fac, phys,
                                      C.c_void_p(d_tf.ptr), C.c_void_p(d_coh.ptr)), "welch")


for _ in range(3):
    step()
ctx.sync()
lib.ds_debug_welch_timing(out)
for _ in range(10):
    step()
ctx.sync()
lib.ds_debug_welch_timing(out)
v = list(out)
iters = v[15]
names = ["window+issue loads", "dft16#1+tw1", "ex1 writes (+tw2 reads) drain", "barrier 1", "ex1 reads",
         "dft16#2+tw2", "ex2 writes drain", "barrier 2", "ex2 reads", "dft16#3", "accumulate (+xs wait)"]
tot = sum(v[:11])
print(f"iterations {iters}, cycles/iteration {tot / iters:.0f}")
for nm, c in zip(names, v[:11]):
    print(f"  {nm:34s} {c / iters:8.0f}  {100.0 * c / tot:5.1f} %")
