"""mmannot_amd — MI355X-native overlap-and-count engine, drop-in for the hot path of mmannot v1.1.

The package holds only what that path needs: csrc/ (HIP kernels + C ABI + host finalisation),
and this thin host-side mirror of the reference's interface.
"""
from .api import (ALN_DTYPE, Config, ConfigElement, Counter, IntervalIndex, TableCount, format_stats, overlap_mode_of,  # noqa: F401
                  pack_alignments)
from ._lib import MMError, lib  # noqa: F401

__version__ = "0.1"
