"""Dev tool: kernels_fir16s.hpp (DSPTOOLBOX_AMD_FIR_16S=1) against the oracle on edge shapes, then the benchmark
shape (32 x 4097 taps over 8 x 2^22) timed against the shipped route, alternating contexts."""
import ctypes as C
import os
import sys
import time

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
worst = 0.0
for n_taps, n, n_ch, n_filt in [(4097, 40000, 2, 3), (4097, 12288 * 3, 3, 2), (4097, 12288 * 2 + 1, 1, 1), (2049, 50001, 3, 2),
                                (8193, 40000, 1, 2), (3000, 70000, 2, 5), (5000, 16384, 2, 1), (2, 20000, 2, 1), (100, 300, 1, 2),
                                (4097, 100, 2, 2), (6001, 8192 * 5 + 7, 5, 3)]:
    x = rng.standard_normal((n, n_ch)) * 0.1
    taps = [rng.standard_normal(n_taps) * np.hanning(n_taps) / np.sqrt(n_taps) for _ in range(n_filt)]
    ctx.routes()
    y = backend.fir_filter_bank(x, taps, backend.DS_FB_PARALLEL)
    routes = ctx.routes()
    err = max(orc.rel_max(y[k], orc.lfilter_fir(taps[k], x)) for k in range(n_filt))
    worst = max(worst, err)
    print(f"taps {n_taps:5d} n {n:6d} ch {n_ch} filters {n_filt}: rel-max {err:.2e}  routes {sorted(routes)}", flush=True)
print("worst", worst)

n, n_ch, K, T = 2**22, 8, 32, 4097
x = np.random.default_rng(3).standard_normal((n, n_ch)) * 0.1
taps = fir_bank_taps(K, T, 48000).astype(np.float32)
xp = backend._planar_f32(x)
ref_out = None
for rep in range(2):
    for env in ({}, {"DSPTOOLBOX_AMD_FIR_16S": "1"}, {"DSPTOOLBOX_AMD_FIR_16S": "1", "DSPTOOLBOX_AMD_FIR_SPLIT": "2"},
                {"DSPTOOLBOX_AMD_FIR_16S": "1", "DSPTOOLBOX_AMD_FIR_SPLIT": "8"},
                {"DSPTOOLBOX_AMD_FIR_16S": "1", "DSPTOOLBOX_AMD_FIR_SPLIT": "32"}):
        ctx = ctx_for(env)
        d_x, d_t = DeviceBuffer.from_array(ctx, xp), DeviceBuffer.from_array(ctx, taps)
        d_y = DeviceBuffer(ctx, K * n_ch * n * 4)

        def step():
            ctx.check(ctx.lib.ds_fir_ola_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, C.c_void_p(d_t.ptr), K, T,
                                             backend.DS_FB_PARALLEL, C.c_void_p(d_y.ptr), n), "ds_fir_ola_dev")
        for _ in range(3):
            step()
        ctx.sync()
        ctx.profile_enable(True)
        ctx.profile_report()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        ctx.sync()
        wall = (time.perf_counter() - t0) / 10 * 1e3
        prof = ctx.profile_report()
        ctx.profile_enable(False)
        got = np.empty((n_ch, n), dtype=np.float32)
        ctx.download(d_y.ptr + 4 * 17 * n_ch * n, got)  # band 17
        if ref_out is None:
            ref_out = orc.lfilter_fir(taps[17].astype(np.float64), x).T
        err = orc.rel_max(got, ref_out)
        print(f"{env or 'shipped'}: step {wall:.3f} ms  kernels " + ", ".join(f"{k} {v[0] / v[1]:.3f} ms" for k, v in sorted(prof.items())) +
              f"  band 17 rel-max {err:.2e}", flush=True)
        for d in (d_x, d_t, d_y):
            d.free()
