"""mmannot_amd — MI355X-native overlap-and-count engine, drop-in for the hot path of mmannot v1.1.

The package holds only what that path needs: csrc/ (HIP kernels + C ABI + host finalisation),
and this thin host-side mirror of the reference's interface.
"""
from .api import (ALN12_DTYPE, ALN_DTYPE, NH_BIG_DTYPE, Config, ConfigElement, Counter, IntervalIndex, TableCount, aln_nh, aln_strand,  # noqa: F401
                  aln_tid, format_stats, nh_side_list, overlap_mode_of, pack12, pack_alignments, row_chunk)
from ._lib import MMError, lib  # noqa: F401

__version__ = "0.1"
