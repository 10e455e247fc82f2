"""Dev tool: where kernels_fir16s.hpp differs from the oracle (first bad samples per case), and a tap-count sweep against
the shipped routes on the bank shape (32 bands over 8 x 2^22)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dsptoolbox_amd import _lib, backend  # noqa: E402
from dsptoolbox_amd._lib import DeviceBuffer  # noqa: E402
from dsptoolbox_amd.generators import fir_bank_taps  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402


def ctx_for(env):
    for k in ("DSPTOOLBOX_AMD_FIR_16S", "DSPTOOLBOX_AMD_FIR_SPLIT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    return _lib.reset_context()


rng = np.random.default_rng(3)
ctx = ctx_for({"DSPTOOLBOX_AMD_FIR_16S": "1"})
for n_taps, n, n_ch, n_filt in [(3000, 70000, 1, 1), (2999, 70000, 1, 1), (3001, 70000, 1, 1), (5000, 16384, 1, 1), (4999, 30000, 1, 1),
                                (4097, 100, 2, 2), (4097, 5000, 1, 1)]:
    x = rng.standard_normal((n, n_ch)) * 0.1
    taps = [rng.standard_normal(n_taps) * np.hanning(n_taps) / np.sqrt(n_taps) for _ in range(n_filt)]
    y = backend.fir_filter_bank(x, taps, backend.DS_FB_PARALLEL)
    ref = orc.lfilter_fir(taps[0], x)
    d = np.abs(y[0] - ref)[:, 0]
    bad = np.nonzero(d > 1e-5 * np.max(np.abs(ref)))[0]
    L = 16384 - (n_taps - 1)
    print(f"taps {n_taps} n {n} L {L}: rel-max {d.max() / np.max(np.abs(ref)):.2e}; bad samples {len(bad)}" +
          (f": first {bad[:6]}, last {bad[-3:]}, positions mod L {sorted(set((bad % L).tolist()))[:8]} ..." if len(bad) else ""), flush=True)
    ctx0 = ctx_for({})
    y0 = backend.fir_filter_bank(x, taps, backend.DS_FB_PARALLEL)
    print(f"    shipped route: rel-max {np.max(np.abs(y0[0] - ref)) / np.max(np.abs(ref)):.2e}", flush=True)
    ctx = ctx_for({"DSPTOOLBOX_AMD_FIR_16S": "1"})

n, n_ch, K = 2**22, 8, 32
x = np.random.default_rng(3).standard_normal((n, n_ch)) * 0.1
xp = backend._planar_f32(x)
for T in (513, 1025, 2049, 3073, 4097, 6145, 8193):
    taps = fir_bank_taps(K, T, 48000).astype(np.float32)
    for env in ({}, {"DSPTOOLBOX_AMD_FIR_16S": "1"}, {"DSPTOOLBOX_AMD_FIR_16S": "1", "DSPTOOLBOX_AMD_FIR_SPLIT": "8"}):
        ctx = ctx_for(env)
        d_x, d_t = DeviceBuffer.from_array(ctx, xp), DeviceBuffer.from_array(ctx, taps)
        d_y = DeviceBuffer(ctx, K * n_ch * n * 4)

        def step():
            ctx.check(ctx.lib.ds_fir_ola_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, C.c_void_p(d_t.ptr), K, T,
                                             backend.DS_FB_PARALLEL, C.c_void_p(d_y.ptr), n), "ds_fir_ola_dev")
        for _ in range(3):
            step()
        ctx.sync()
        ctx.routes()
        ctx.timer_start()
        for _ in range(10):
            step()
        ms = ctx.timer_stop() / 10
        print(f"taps {T:5d} {str(env or 'shipped'):75s}: step {ms:.3f} ms   {sorted(ctx.routes())}", flush=True)
        for d in (d_x, d_t, d_y):
            d.free()
