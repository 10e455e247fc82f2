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

__all__ = ["delay_and_sum_map", "quadratic_form_map", "BeamformerDASFrequency"]


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


class BeamformerDASFrequency:
    """Frequency-domain delay-and-sum beamformer with the reference's interface
    (beamforming/beamforming.py:760-880): built from a multi-channel Signal, a microphone array, a grid and
    a steering vector -- the reference's own geometry objects, or anything that offers
    `steering_vector.get_vector(wave_numbers, grid=, mic=) -> (bins, channels, grid points)`,
    `grid.number_of_points` and `grid.reconstruct_map_shape(map)`.  The cross-spectral matrix is computed
    on the device and STAYS there (Signal.get_csm(on_device=True)); the diagonal treatment and the
    grid x bin quadratic forms run on it in place; only the steering vectors travel up and the map down."""

    beamformer_type = "Delay-and-sum (Frequency)"

    def __init__(self, multi_channel_signal, mic_array, grid, steering_vector, c: float = 343):
        assert multi_channel_signal.number_of_channels > 1, "Signal must be multichannel"
        assert c > 0, "Speed of sound should be bigger than 0"
        assert hasattr(steering_vector, "get_vector"), "steering_vector should offer get_vector()"
        assert hasattr(grid, "number_of_points") and hasattr(grid, "reconstruct_map_shape"), "grid should be a Grid object"
        self.signal, self.mics, self.grid, self.st_vec, self.c = multi_channel_signal, mic_array, grid, steering_vector, c

    def set_csm_parameters(self, **kwargs):
        """Spectrum parameters of the multi-channel signal's CSM (Signal.set_spectrum_parameters)."""
        self.signal.set_spectrum_parameters(**kwargs)

    def get_beamformer_map(self, center_frequency_hz: float, octave_fraction: int = 3,
                           remove_csm_diagonal: bool = True) -> np.ndarray:
        from ..transfer_functions import find_nearest_points_index_in_vector
        self.center_frequency_hz, self.octave_fraction = center_frequency_hz, octave_fraction
        # helpers/other.py:156-178
        self.f_range_hz = (np.array([center_frequency_hz, center_frequency_hz]) if octave_fraction == 0 else
                           np.array([center_frequency_hz * 2 ** (-1 / octave_fraction / 2),
                                     center_frequency_hz * 2 ** (1 / octave_fraction / 2)]))
        f, csm = self.signal.get_csm(on_device=True)
        try:
            ids = find_nearest_points_index_in_vector(self.f_range_hz, f)
            id1, id2 = int(ids[0]), int(ids[1])
            if id1 == id2:
                id2 += 1
            f = f[id1:id2]
            wave_numbers = f * np.pi * 2 / self.c
            h = self.st_vec.get_vector(wave_numbers, grid=self.grid, mic=self.mics)
            self.f_range_hz = np.array([f[0], f[-1]])
            m = backend._das_map_device(csm, id1, id2, h, remove_csm_diagonal)
        finally:
            csm.free()
        if remove_csm_diagonal:
            m[m < 0] = 0  # unphysical values for the removed diagonal
        m = simpson(m, dx=f[1] - f[0], axis=1) if id2 - id1 > 1 else m.squeeze()
        self.map = self.grid.reconstruct_map_shape(m)
        return self.map.copy()
