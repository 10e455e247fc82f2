from .filter import Filter
from .filterbank import FilterBank
from .impulse_response import ImpulseResponse
from .multibandsignal import MultiBandSignal
from .signal import Signal
from .spectrum import Spectrum

__all__ = ["Signal", "ImpulseResponse", "Spectrum", "Filter", "FilterBank", "MultiBandSignal"]
