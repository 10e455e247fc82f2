from enum import Enum, auto


class TransferFunctionType(Enum):
    """H1 = Gxy/Gxx (noise at the output), H2 = Gyy/Gyx (noise at the input),
    H3 = Gxy/|Gxy| * sqrt(Gyy/Gxx) (noise in both)."""

    H1 = auto()
    H2 = auto()
    H3 = auto()
