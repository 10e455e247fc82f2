"""Parity of the Welch-4096 path vs the float64 oracle as a function of the chunk count
(length of the fp32 accumulation chains).  Headline shape, first NCH channels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsptoolbox_amd import backend  # noqa: E402
from dsptoolbox_amd.generators import sweep_and_responses  # noqa: E402
from oracle import dsp_oracle as orc  # noqa: E402

NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x, y = sweep_and_responses(2**20, NCH, 48000)
f = np.fft.rfftfreq(4096, 1 / 48000)
m = (f >= 30) & (f <= 19000)
for det in (True, False):
    rt, rc = orc.compute_transfer_function_batched(y, x, 48000, 4096, "H1", detrend=det, workers=-1)
    for chunks in (4, 8, 16, 32, 64, 128):
        os.environ["DSPTOOLBOX_AMD_WELCH_CHUNKS"] = str(chunks)
        os.environ["DSPTOOLBOX_AMD_WELCH_OCC"] = "2"
        tf, coh = backend.welch_transfer_function(y, x, 48000, 4096, "H1", detrend=det)
        print(f"detrend={det} chunks={chunks:4d}  tf rel-max {orc.rel_max(tf[m], rt[m]):.3e}  "
              f"coh rel-max {orc.rel_max(coh[m], rc[m]):.3e}  tf rel-l2 {orc.rel_l2(tf[m], rt[m]):.3e}")
os.environ["DSPTOOLBOX_AMD_NO_WELCH4096"] = "1"
