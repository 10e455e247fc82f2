"""dsptoolbox_amd -- MI355X-native implementation of dsptoolbox's batched
spectral hot path (Welch/H1-H3, STFT, CSM, spectral deconvolution, FIR filter
banks) behind the reference's Signal / Filter / FilterBank API."""

from . import beamforming, filterbanks, room_acoustics, transfer_functions, transforms
from .classes import Filter, FilterBank, ImpulseResponse, MultiBandSignal, Signal, Spectrum
from .standard.enums import (FilterBankMode, FilterCoefficientsType, FilterPassType,
                             SpectrumMethod, SpectrumScaling, SpectrumType, Window)
from .transfer_functions.enums import TransferFunctionType

__version__ = "0.1.0"
__all__ = ["Signal", "ImpulseResponse", "Spectrum", "Filter", "FilterBank", "MultiBandSignal",
           "SpectrumMethod", "SpectrumScaling", "SpectrumType", "Window", "FilterBankMode",
           "FilterPassType", "FilterCoefficientsType", "TransferFunctionType",
           "transfer_functions", "transforms", "room_acoustics", "beamforming", "filterbanks"]
