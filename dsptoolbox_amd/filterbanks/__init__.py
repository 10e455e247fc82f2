"""filterbanks: the block-streaming FIR classes of the reference's filterbanks module
(dsptoolbox/filterbanks/__init__.py:78-83) on the device FIR kernels."""

from ..classes.fir_filter_realtime import (FIRFilterOverlapSave, FIRUniformPartitioned,
                                           FIRUniformPartitionedMultichannel)

__all__ = ["FIRFilterOverlapSave", "FIRUniformPartitioned", "FIRUniformPartitionedMultichannel"]
