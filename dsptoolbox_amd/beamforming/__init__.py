"""beamforming: the delay-and-sum map on the cross-spectral matrix
(dsptoolbox/beamforming/beamforming.py:799-880).

The reference's grid / microphone-array / steering-vector classes stay the reference's; what is
replaced is the hot double loop over grid points and frequency bins (:853-858).  A reference
maintainer calls `delay_and_sum_map(f, csm, h, remove_csm_diagonal)` with the selected bins
`f[id1:id2]`, the CSM slice `csm[id1:id2]` (before the diagonal treatment) and the steering
vectors `h = st_vec.get_vector(...)` and gets the integrated map vector back."""

from __future__ import annotations

import numpy as np
from scipy.integrate import simpson

from .. import backend

__all__ = ["delay_and_sum_map", "quadratic_form_map"]


def quadratic_form_map(csm, h) -> np.ndarray:
    """map[g, f] = Re(h[f, :, g]^H csm[f] h[f, :, g]) on the device (fp32 MFMA).
    csm (F, C, C), h (F, C, G) complex -> (G, F) float64."""
    return backend._das_map(csm, h)


def delay_and_sum_map(f, csm, h, remove_csm_diagonal: bool = True) -> np.ndarray:
    """Frequency-domain delay-and-sum map integrated over the bins `f` (:838-876)."""
    f = np.asarray(f, dtype=np.float64)
    csm = np.array(csm, dtype=np.complex128)  # the reference scales / zeroes its CSM in place: copy
    n_ch = csm.shape[1]
    if remove_csm_diagonal:
        csm *= n_ch / (n_ch - 1)  # account for energy loss
        for i in range(len(f)):
            np.fill_diagonal(csm[i, :, :], 0)
    m = quadratic_form_map(csm, h)
    if remove_csm_diagonal:
        m[m < 0] = 0  # unphysical values for the removed diagonal
    if len(f) > 1:
        return simpson(m, dx=f[1] - f[0], axis=1)
    return m.squeeze()
