"""Dev tool: the batched 8192-point deconvolution (1024 stereo items, BASELINE config 5) with half of the workgroups
starting late (DSPTOOLBOX_AMD_DECONV_STAGGER="bit,ticks": workgroups whose index has that bit set wait ticks x 10 ns).
Kernel time by the dispatch's own timestamps (ds_profile_*), alternating with the unstaggered kernel."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dsptoolbox_amd import _lib  # noqa: E402
from dsptoolbox_amd._lib import DeviceBuffer  # noqa: E402

n, items, n_ch = 8192, 1024, 2
rng = np.random.default_rng(5000)
y = rng.standard_normal((items, n_ch, n)).astype(np.float32) * 0.1
r = (rng.standard_normal(n // 2 + 1) + 1j * rng.standard_normal(n // 2 + 1)).astype(np.complex64)


def run(setting):
    if setting is None:
        os.environ.pop("DSPTOOLBOX_AMD_DECONV_STAGGER", None)
    else:
        os.environ["DSPTOOLBOX_AMD_DECONV_STAGGER"] = setting
    ctx = _lib.reset_context()
    d_y, d_r = DeviceBuffer.from_array(ctx, y), DeviceBuffer.from_array(ctx, r)
    d_o = DeviceBuffer(ctx, y.nbytes)

    def step():
        ctx.check(ctx.lib.ds_deconv_dev(ctx.handle, C.c_void_p(d_y.ptr), items, n_ch, n, n, n, C.c_void_p(d_r.ptr), 0, n, n,
                                        C.c_void_p(d_o.ptr)), "ds_deconv_dev")
    for _ in range(20):
        step()
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_report()
    for _ in range(100):
        step()
    rep = ctx.profile_report()
    ctx.profile_enable(False)
    out = d_o.to_array((items, n_ch, n), np.float32)
    for d in (d_y, d_r, d_o):
        d.free()
    ms, cnt = rep["deconv"]
    return 1e3 * ms / cnt, out


base_us, ref = run(None)
print(f"no stagger: {base_us:.1f} us")
settings = [f"{b},{t}" for b in (0, 3, 5, 8, 9) for t in (300, 600, 900, 1200)]
for s in settings + ["0,0"]:
    us, out = run(s)
    b_us, _ = run(None)
    print(f"stagger {s:8s}: {us:5.1f} us   (unstaggered right after: {b_us:5.1f} us)   identical output: {np.array_equal(out, ref)}", flush=True)
