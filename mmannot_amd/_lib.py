"""ctypes binding of include/mmannot_amd.h.

torch is imported first on purpose: PyTorch-ROCm ships its own libamdhip64.so (SONAME
libamdhip64.so.7); loading it before libmmannot_amd.so makes both share ONE HIP runtime in the
process, so device pointers and streams created by torch are valid inside the library.

There is no fallback: if the shared library is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

try:  # plumbing only (device memory, streams, torch.distributed)
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch-less use of the C ABI is allowed
    torch = None

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libmmannot_amd.so")

MM_N_STATS = 7
MM_BATCH_ON_DEVICE, MM_BATCH_SINGLE_HIT = 1, 2
MM_ALN_NH_BITS = 11
MM_ALN_NH_ESCAPE = (1 << MM_ALN_NH_BITS) - 1
STAT_NAMES = ("nHits", "nReads", "nUnique", "nAmbiguous", "nMultiple", "nUnassigned", "nRescued")


class MMError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mmannot_amd error {code}: {msg}")
        self.code = code


class Elements(C.Structure):
    _fields_ = [("n_elements", C.c_int32), ("line_of", C.c_void_p), ("elem_strand", C.c_void_p),
                ("is_up", C.c_void_p), ("is_down", C.c_void_p)]


class Options(C.Structure):
    _fields_ = [("overlap_mode", C.c_int32), ("overlap_value", C.c_float), ("strandedness", C.c_int32),
                ("strategy", C.c_int32), ("keep_hit_masks", C.c_int32), ("defer_ratio", C.c_int32)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first", C.c_uint64), ("read_len", C.c_int32), ("nh_max", C.c_int32),
                ("n_chromosomes", C.c_int32), ("chr_len", C.c_void_p), ("scatter_log2", C.c_uint32), ("pair_insert", C.c_uint32)]


_lib = None

# every symbol include/mmannot_amd.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = {
    "mm_last_error": (C.c_char_p, []),
    "mm_version": (C.c_char_p, []),
    "mm_device_count": (C.c_int, []),
    "mm_index_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(Elements), C.c_int32, C.c_size_t, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "mm_index_destroy": (None, [C.c_void_p]),
    "mm_index_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mm_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(Options), C.c_void_p]),
    "mm_ctx_destroy": (None, [C.c_void_p]),
    "mm_reset": (C.c_int, [C.c_void_p]),
    "mm_reserve_qids": (C.c_int, [C.c_void_p, C.c_uint64]),
    "mm_annotate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32]),
    "mm_pack12": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mm_annotate_batch12": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32]),
    "mm_host_alloc": (C.c_void_p, [C.c_size_t]),
    "mm_host_free": (None, [C.c_void_p]),
    "mm_sync": (C.c_int, [C.c_void_p]),
    "mm_finalize": (C.c_int, [C.c_void_p]),
    "mm_get_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mm_get_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mm_get_hit_masks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mm_get_hit_intervals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "mm_get_walk_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mm_get_probe_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "mm_probe_path": (C.c_int, [C.c_void_p]),
    "mm_set_row_sums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mm_ratio_resolve": (C.c_int, [C.c_void_p]),
    "mm_ratio_resolve_n": (C.c_int, [C.c_void_p, C.c_size_t]),
    "mm_ratio_pending": (C.c_size_t, [C.c_void_p]),
    "mm_table_export_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm_table_export_counts_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mm_table_merge_counts_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "mm_table_import_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]),
    "mm_ratio_replay": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32]),
    "mm_ratio_release": (C.c_int, [C.c_void_p]),
    "mm_ratio_reserve": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]),
    "mm_ratio_handoff": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "mm_ratio_resolve_chunk": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "mm_row_chunk": (C.c_uint32, [C.c_uint64, C.c_uint32]),
    "mm_partition": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    "mm_partition_threads": (C.c_int, [C.c_int]),
    "mm_merge": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "mm_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mm_get_phase_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "mm_round_to_uint": (C.c_uint32, [C.c_double]),
    "mm_sort_rows": (None, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mm_annotation_load": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32]),
    "mm_annotation_free": (None, [C.c_void_p]),
    "mm_annotation_error": (C.c_char_p, []),
    "mm_annotation_log": (C.c_char_p, [C.c_void_p]),
    "mm_annotation_n_elements": (C.c_int32, [C.c_void_p]),
    "mm_annotation_n_chromosomes": (C.c_int32, [C.c_void_p]),
    "mm_annotation_n_intervals": (C.c_size_t, [C.c_void_p]),
    "mm_annotation_element_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "mm_annotation_chromosome": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "mm_annotation_chromosome_has_intervals": (C.c_int, [C.c_void_p, C.c_int32]),
    "mm_annotation_interval_id": (C.c_char_p, [C.c_void_p, C.c_size_t]),
    "mm_annotation_elements": (C.c_int, [C.c_void_p, C.POINTER(Elements)]),
    "mm_annotation_intervals": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 5),
    "mm_index_from_annotation": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "mm_reader_open": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mm_reader_error": (C.c_char_p, [C.c_void_p]),
    "mm_reader_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                 C.POINTER(C.c_uint32)]),
    "mm_reader_qid_bound": (C.c_uint64, [C.c_void_p]),
    "mm_reader_qid_lo": (C.c_uint64, [C.c_void_p]),
    "mm_inflate_block": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "mm_reader_n_reads": (C.c_uint64, [C.c_void_p]),
    "mm_reader_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "mm_reader_take_log": (C.c_char_p, [C.c_void_p]),
    "mm_reader_close": (None, [C.c_void_p]),
    "mm_synth_fill": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(SynthParams), C.c_void_p]),
}


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("MMANNOT_AMD_LIB", LIB_PATH)   # another build of the same library (kernel experiments)
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python -m mmannot_amd.build` "
                              "(the HIP library is the only implementation of the hot path; there is no fallback)")
        l = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            if not hasattr(l, name):
                continue  # optional groups are bound lazily below
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise MMError(rc, lib().mm_last_error().decode(errors="replace"))
