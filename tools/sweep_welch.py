"""Interleaved timing of the Welch-4096 kernel variants (one process, one device;
cdna_hip_programming.md rule 24).  Usage: python tools/sweep_welch.py [rounds]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd._lib import Context, DeviceBuffer  # noqa: E402
from dsptoolbox_amd.standard.enums import SpectrumScaling, Window  # noqa: E402

n, n_cy, W, FS = 2**20, 64, 4096, 48000
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = Context(0)
rng = np.random.default_rng(0)
y = rng.standard_normal((n_cy, n)).astype(np.float32) * 0.3
x = rng.standard_normal((1, n)).astype(np.float32) * 0.3
window = backend._window_array(Window.Hann, W)
hop, n_frames = backend._welch_framing(n, W, 50, window)
amp, norm_scale, factor, phys = backend._finish_params(SpectrumScaling.FFTBackward, W, FS, window)
d_y, d_x = DeviceBuffer.from_array(ctx, y), DeviceBuffer.from_array(ctx, x)
d_w = DeviceBuffer.from_array(ctx, window.astype(np.float32))
B = W // 2 + 1
d_tf, d_coh = DeviceBuffer(ctx, B * n_cy * 8), DeviceBuffer(ctx, B * n_cy * 4)


def step():
    ctx.check(ctx.lib.ds_welch_tf_dev(ctx.handle, C.c_void_p(d_x.ptr), 1, n, C.c_void_p(d_y.ptr), n_cy, n,
                                      n, W, hop, n_frames, C.c_void_p(d_w.ptr), 1, 0, 1, amp, norm_scale,
                                      factor, phys, C.c_void_p(d_tf.ptr), C.c_void_p(d_coh.ptr)), "tf")


variants = [(2, q) for q in (4, 8, 16, 24, 32)]
if os.environ.get("SWEEP_VARIANTS"):
    variants = [tuple(int(v) for v in s.split(":")) for s in os.environ["SWEEP_VARIANTS"].split(",")]
res = {v: [] for v in variants}
ctx.profile_enable(True)
for r in range(rounds + 1):
    for v in variants:
        os.environ["DSPTOOLBOX_AMD_WELCH_CHUNKS"] = str(v[1])
        for _ in range(3):
            step()
        prof = ctx.profile_report()
        if r > 0:
            res[v].append({k: ms / cnt for k, (ms, cnt) in prof.items()})
for v in variants:
    main = sorted(d["welch4096_main"] for d in res[v])
    tot = sorted(sum(d.values()) for d in res[v])
    print(f"chunks={v[1]:3d}  main med {main[len(main)//2]*1e3:7.1f} us  min {main[0]*1e3:7.1f} us   "
          f"all kernels med {tot[len(tot)//2]*1e3:7.1f} us  "
          + " ".join(f"{k}={res[v][-1][k]*1e3:.1f}" for k in res[v][-1] if k != "welch4096_main"))
