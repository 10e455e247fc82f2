"""Dev tool: ds_deconv_dev (stereo items against one shared inverse spectrum, device resident) across transform
lengths, the batch sized to ~64 MB of samples: ms per call and bytes moved (samples in + impulse responses out)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsptoolbox_amd._lib import DeviceBuffer, get_context  # noqa: E402

ctx = get_context()
rng = np.random.default_rng(0)
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 65536, 262144]
for n in sizes:
    n_ch = 2
    items = max(1, (64 << 20) // (n * n_ch * 4))
    y = rng.standard_normal((items, n_ch, n)).astype(np.float32) * 0.1
    r = (rng.standard_normal(n // 2 + 1) + 1j * rng.standard_normal(n // 2 + 1)).astype(np.complex64)
    d_y, d_r = DeviceBuffer.from_array(ctx, y), DeviceBuffer.from_array(ctx, r)
    d_o = DeviceBuffer(ctx, y.nbytes)

    def step():
        ctx.check(ctx.lib.ds_deconv_dev(ctx.handle, C.c_void_p(d_y.ptr), items, n_ch, n, n, n, C.c_void_p(d_r.ptr), 0, n, n,
                                        C.c_void_p(d_o.ptr)), "ds_deconv_dev")
    for _ in range(3):
        step()
    ctx.sync()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        step()
    ctx.sync()
    ms = (time.perf_counter() - t0) / K * 1e3
    tot = 2 * y.nbytes
    print(f"n {n:7d} x {items:6d} items: {ms:7.3f} ms  {tot / 1e6:7.1f} MB  {tot / ms / 1e9:5.2f} TB/s", flush=True)
    for d in (d_y, d_r, d_o):
        d.free()
