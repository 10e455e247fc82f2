"""Dev tool: ds_fir_ola_dev, 32 bands over 8 channels x 2^22 samples (device resident), across tap
counts: ms per call and the output rate (4.29 GB of fp32 out)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd._lib import DeviceBuffer, get_context  # noqa: E402

ctx = get_context()
n, n_ch, n_filt = 2**22, 8, 32
rng = np.random.default_rng(0)
d_x = DeviceBuffer.from_array(ctx, (rng.standard_normal((n_ch, n)) * 0.1).astype(np.float32))
d_y = DeviceBuffer(ctx, n_filt * n_ch * n * 4)
for T in [int(a) for a in sys.argv[1:]] or [129, 513, 1025, 2049, 4097, 8193]:
    d_t = DeviceBuffer.from_array(ctx, (rng.standard_normal((n_filt, T)) * 0.01).astype(np.float32))

    def step():
        ctx.check(ctx.lib.ds_fir_ola_dev(ctx.handle, C.c_void_p(d_x.ptr), n_ch, n, n, C.c_void_p(d_t.ptr), n_filt, T, 1,
                                         C.c_void_p(d_y.ptr), n), "ds_fir_ola_dev")
    for _ in range(2):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        step()
    ctx.sync()
    ms = (time.perf_counter() - t0) / K * 1e3
    print(f"taps {T:5d}: {ms:8.3f} ms   {n_filt * n_ch * n * 4 / ms / 1e9:5.2f} TB/s of output", flush=True)
    d_t.free()
