"""Host-side mirror of the reference's interface for the overlap-and-count path.

Same names and argument meaning as the classes of mmannot.cpp so that the parity tests read
like the reference's call sites:

    Config         mmannot.cpp:219-471   (the flattened Order: elements, names, strands)
    IntervalIndex  IntervalList          mmannot.cpp:1084-1336 (device-resident index; scan() lives in the kernel)
    Counter        Counter               mmannot.cpp:1653-1852 (clear / read -> add_batch+flush / getCounts / dump)
    TableCount     TableCount            mmannot.cpp:1854-1901 (addCounter / dump)

Everything heavy happens behind the C ABI (include/mmannot_amd.h) on the GPU; this module holds
no compute and has no fallback path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import Elements, Options, check, lib

STRAND_ALL, STRAND_F, STRAND_R = 0, 1, 2
OVL_INCLUSION, OVL_PC, OVL_NT = 0, 1, 2
STRANDEDNESS = {"U": 0, "F": 1, "R": 2}
STRATEGY = {"default": 0, "unique": 1, "random": 2, "ratio": 3}

ALN_DTYPE = np.dtype([("start", "<i4"), ("end", "<i4"), ("tsn", "<u4"), ("qid", "<u4")])     # mm_aln, 16 bytes
NH_BIG_DTYPE = np.dtype([("index", "<u4"), ("nh", "<u4")])                                     # mm_nh_big
NH_BITS, NH_ESCAPE = _lib.MM_ALN_NH_BITS, _lib.MM_ALN_NH_ESCAPE


def pack_alignments(tid, start, end, strand, qid, nh=None) -> np.ndarray:
    """SoA -> the 16-byte mm_aln tuples (include/mmannot_amd.h): tsn = strand | NH << 1 | (tid + 1) << 12.
    tid < 0 = chromosome unknown to the annotation; nh None = every NH is 1; an NH beyond the 11-bit field becomes the escape
    value (its true value then travels in the side list nh_side_list builds)."""
    n = len(start)
    a = np.empty(n, dtype=ALN_DTYPE)
    a["start"] = start
    a["end"] = end
    nhf = np.ones(n, dtype=np.uint32) if nh is None else np.minimum(np.asarray(nh, dtype=np.uint64), NH_ESCAPE).astype(np.uint32)
    a["tsn"] = ((np.asarray(tid, dtype=np.int64) + 1).astype(np.uint32) << np.uint32(NH_BITS + 1)) | (nhf << np.uint32(1)) | \
        (np.asarray(strand, dtype=np.uint32) & np.uint32(1))
    a["qid"] = np.asarray(qid, dtype=np.uint32)
    return a


def nh_side_list(nh) -> np.ndarray:
    """mm_nh_big entries (ascending tuple index) for the NH values that do not fit the tuple's NH field."""
    nh = np.asarray(nh, dtype=np.uint64)
    idx = np.nonzero(nh >= NH_ESCAPE)[0]
    big = np.empty(len(idx), dtype=NH_BIG_DTYPE)
    big["index"] = idx
    big["nh"] = nh[idx]
    return big


ALN12_DTYPE = np.dtype([("w", "<u4", (3,))])                                                  # mm_aln12, 12 bytes


def pack12(aln: np.ndarray, big: Optional[np.ndarray] = None, qid_base: int = 0):
    """16-byte tuples (+ their NH side list) -> the 12-byte link format and its side list (NH >= 255); raises MMError(MM_ERR_RANGE)
    when a tuple does not fit (the batch then travels as 16-byte tuples)."""
    aln = np.ascontiguousarray(aln)
    assert aln.dtype == ALN_DTYPE
    nb = 0 if big is None else len(big)
    out = np.empty(len(aln), dtype=ALN12_DTYPE)
    ob = np.empty(max(64, len(aln)), dtype=NH_BIG_DTYPE)
    n_out = C.c_size_t(0)
    rc = lib().mm_pack12(aln.ctypes.data, len(aln), big.ctypes.data if nb else None, nb, int(qid_base), out.ctypes.data, ob.ctypes.data, len(ob), C.byref(n_out))
    if rc != 0:
        raise _lib.MMError(rc, "mm_pack12: a tuple does not fit the 12-byte format" if rc == 3 else "mm_pack12: bad argument")
    return out, ob[:n_out.value].copy()


def aln_tid(aln) -> np.ndarray:
    return (aln["tsn"] >> np.uint32(NH_BITS + 1)).astype(np.int64) - 1


def aln_strand(aln) -> np.ndarray:
    return (aln["tsn"] & np.uint32(1)).astype(np.uint8)


def aln_nh(aln, big=None) -> np.ndarray:
    """record.nHits of every tuple (NH field, side list applied)."""
    nh = ((aln["tsn"] >> np.uint32(1)) & np.uint32(NH_ESCAPE)).astype(np.uint32)
    if big is not None and len(big):
        nh[big["index"]] = big["nh"]
    return nh


def row_chunk(masks: np.ndarray, n_chunks: int) -> np.ndarray:
    """mm_row_chunk for an array of element sets: which chunk of a pipelined -y ratio chain a row belongs to."""
    m = np.asarray(masks, dtype=np.uint64)
    if n_chunks <= 1:
        return np.zeros(len(m), dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = ((m * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)).astype(np.uint32)
    return h % np.uint32(n_chunks)


def overlap_mode_of(l: float) -> int:
    """-l: <0 inclusion, <1 fraction of the read, else nucleotides (mmannot.cpp:1972-1977)."""
    l = np.float32(l)
    if l < 0.0:
        return OVL_INCLUSION
    if l < 1.0:
        return OVL_PC
    return OVL_NT


@dataclass
class ConfigElement:
    source: str
    type: str
    strand: int
    line: int


class Config:
    """The part of Config (mmannot.cpp:219-471) the hot path uses: the flattened Order."""

    def __init__(self, elements: Sequence[ConfigElement]):
        self.elements = list(elements)
        if not (1 <= len(self.elements) <= 64):
            raise ValueError("1..64 Order elements are supported on the device (element sets travel as 64-bit masks)")

    @staticmethod
    def from_json(js) -> "Config":
        return Config([ConfigElement(d["source"], d["type"], int(d["strand"]), int(d["line"])) for d in js])

    def getNElements(self) -> int:
        return len(self.elements)

    def getName(self, i: int) -> str:  # mmannot.cpp:445-462
        e = self.elements[i]
        s = e.source
        if e.type:
            s += ":" + e.type
        if e.strand == STRAND_F:
            s += " (+)"
        elif e.strand == STRAND_R:
            s += " (-)"
        return s

    def isUpstream(self, i: int) -> bool:  # mmannot.cpp:463-466
        return self.elements[i].type == "upstream"

    def isDownstream(self, i: int) -> bool:  # mmannot.cpp:467-470
        return self.elements[i].type == "downstream"

    def _c_elements(self):
        E = len(self.elements)
        self._line = np.array([e.line for e in self.elements], dtype=np.int32)
        self._strand = np.array([e.strand for e in self.elements], dtype=np.uint8)
        self._up = np.array([self.isUpstream(i) for i in range(E)], dtype=np.uint8)
        self._down = np.array([self.isDownstream(i) for i in range(E)], dtype=np.uint8)
        return Elements(E, self._line.ctypes.data, self._strand.ctypes.data, self._up.ctypes.data, self._down.ctypes.data)


class IntervalIndex:
    """IntervalList's sorted interval vector + bins (mmannot.cpp:1086-1090, 1267-1284) resident in HBM."""

    def __init__(self, config: Config, chr, start, end, type, strand, n_chromosomes: int, device: int = 0, bin_shift: int = 0):
        self.config = config
        self.n_chromosomes = int(n_chromosomes)
        chr_ = np.ascontiguousarray(chr, dtype=np.int32)
        st = np.ascontiguousarray(start, dtype=np.int64)
        en = np.ascontiguousarray(end, dtype=np.int64)
        if len(st) and (st.max() >= 2**31 - 16 or en.max() >= 2**31 - 16 or st.min() < 0):
            raise ValueError("coordinates must be in [0, 2^31)")
        st32 = st.astype(np.int32)
        en32 = en.astype(np.int32)
        ty = np.ascontiguousarray(type, dtype=np.uint8)
        sd = np.ascontiguousarray(strand, dtype=np.uint8)
        self.n_intervals = len(st32)
        h = C.c_void_p()
        el = config._c_elements()
        check(lib().mm_index_create(C.byref(h), C.byref(el), self.n_chromosomes, self.n_intervals, chr_.ctypes.data,
                                    st32.ctypes.data, en32.ctypes.data, ty.ctypes.data, sd.ctypes.data, device, bin_shift))
        self._h = h
        self.device = device

    def info(self) -> Dict[str, int]:
        a = (C.c_uint64 * 8)()
        check(lib().mm_index_info(self._h, a))
        return dict(interval_bytes=a[0], bin_bytes=a[1], active_bytes=a[2], bin_shift=a[3], slot_bytes=a[4], flat_bytes=a[5], slot_shift=a[6], slot_bitmap_bytes=a[7])

    def close(self):
        if getattr(self, "_h", None):
            lib().mm_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Counter:
    """Counter (mmannot.cpp:1653-1852) on one GPU."""

    def __init__(self, index: IntervalIndex, *, overlap: float = -1.0, strandedness: str = "F", strategy: str = "default",
                 keep_hit_masks: bool = False, stream: Optional[int] = None, defer_ratio: bool = False):
        self.index = index
        self.strategy = strategy
        opt = Options(overlap_mode_of(overlap), float(np.float32(overlap)), STRANDEDNESS[strandedness], STRATEGY[strategy],
                      1 if keep_hit_masks else 0, 1 if defer_ratio else 0)
        h = C.c_void_p()
        check(lib().mm_ctx_create(C.byref(h), index._h, C.byref(opt), C.c_void_p(stream) if stream else None))
        self._h = h
        self.stream = stream or 0   # the HIP stream handle the context was given (0: a stream of its own)
        self._keep = []  # host buffers that must outlive the async copy

    def clear(self):  # Counter::clear mmannot.cpp:1742-1747
        check(lib().mm_reset(self._h))

    def reserve(self, qid_bound: int):
        """Size the per-read-name state ahead of time (optional)."""
        check(lib().mm_reserve_qids(self._h, int(qid_bound)))

    def add_batch(self, aln: np.ndarray, nh: Optional[np.ndarray], qid_bound: int, single_hit: Optional[bool] = None, qid_lo: int = 0):
        """Host tuples (mm_aln structured array) -> one batch of the loop of Counter::read (mmannot.cpp:1772-1781).
        nh given: the NH of every tuple (packed into a copy of the tuples, values beyond the field into the side list);
        nh None: the tuples' own NH fields count."""
        aln = np.ascontiguousarray(aln)
        assert aln.dtype == ALN_DTYPE
        big = np.empty(0, dtype=NH_BIG_DTYPE)
        if nh is not None:
            nh = np.asarray(nh, dtype=np.uint64)
            aln = aln.copy()
            aln["tsn"] = (aln["tsn"] & ~np.uint32(NH_ESCAPE << 1)) | (np.minimum(nh, NH_ESCAPE).astype(np.uint32) << np.uint32(1))
            big = nh_side_list(nh)
        if single_hit is None:
            single_hit = len(aln) > 0 and int(((aln["tsn"] >> np.uint32(1)) & np.uint32(NH_ESCAPE)).max()) <= 1
        self._keep.append(aln)
        check(lib().mm_annotate_batch(self._h, aln.ctypes.data, len(aln), big.ctypes.data if len(big) else None, len(big), int(qid_lo), int(qid_bound),
                                      _lib.MM_BATCH_SINGLE_HIT if single_hit else 0))

    def add_batch_device(self, aln_ptr: int, n: int, qid_bound: int, single_hit: bool = False, big: Optional[np.ndarray] = None, qid_lo: int = 0):
        """Tuples already resident in HBM (e.g. a torch tensor: pass .data_ptr())."""
        flags = _lib.MM_BATCH_ON_DEVICE | (_lib.MM_BATCH_SINGLE_HIT if single_hit else 0)
        nb = 0 if big is None else len(big)
        check(lib().mm_annotate_batch(self._h, C.c_void_p(aln_ptr), int(n), big.ctypes.data if nb else None, nb, int(qid_lo), int(qid_bound), flags))

    def add_batch12(self, pk: np.ndarray, big: Optional[np.ndarray], qid_base: int, qid_bound: int, single_hit: bool = False, qid_lo: int = 0,
                    on_device_ptr: Optional[int] = None):
        """12-byte tuples (pack12): host array, or - on_device_ptr - already resident in HBM."""
        nb = 0 if big is None else len(big)
        flags = (_lib.MM_BATCH_SINGLE_HIT if single_hit else 0) | (_lib.MM_BATCH_ON_DEVICE if on_device_ptr else 0)
        if on_device_ptr is None:
            pk = np.ascontiguousarray(pk)
            assert pk.dtype == ALN12_DTYPE
            self._keep.append(pk)
        ptr = on_device_ptr if on_device_ptr else pk.ctypes.data
        check(lib().mm_annotate_batch12(self._h, C.c_void_p(ptr), len(pk), int(qid_base), big.ctypes.data if nb else None, nb, int(qid_lo), int(qid_bound), flags))

    def add_batch_host12(self, pk_ptr: int, n: int, qid_base: int, qid_bound: int, single_hit: bool = False, qid_lo: int = 0):
        """12-byte tuples in (pinned) host memory the caller keeps alive."""
        check(lib().mm_annotate_batch12(self._h, C.c_void_p(pk_ptr), int(n), int(qid_base), None, 0, int(qid_lo), int(qid_bound),
                                        _lib.MM_BATCH_SINGLE_HIT if single_hit else 0))

    def add_batch_host(self, aln_ptr: int, n: int, qid_bound: int, single_hit: bool = False, qid_lo: int = 0):
        """Packed tuples in (pinned) host memory the caller keeps alive (see the lifetime rule of mm_annotate_batch)."""
        check(lib().mm_annotate_batch(self._h, C.c_void_p(aln_ptr), int(n), None, 0, int(qid_lo), int(qid_bound),
                                      _lib.MM_BATCH_SINGLE_HIT if single_hit else 0))

    def sync(self):
        check(lib().mm_sync(self._h))
        self._keep.clear()

    def flush(self):
        """End of file: pending multi-hit reads are counted (mmannot.cpp:1783-1792)."""
        check(lib().mm_finalize(self._h))
        self._keep.clear()

    def stats(self) -> Dict[str, int]:
        a = (C.c_uint64 * _lib.MM_N_STATS)()
        check(lib().mm_get_stats(self._h, a))
        return {nm: int(a[i]) for i, nm in enumerate(_lib.STAT_NAMES)}

    def getCounts(self):
        """Counter::getCounts mmannot.cpp:1803: (masks, sums) unsorted."""
        n = C.c_size_t(0)
        masks = np.zeros(self.TABLE_SLOTS, dtype=np.uint64)      # (one call: the table cannot hold more rows than it has slots)
        sums = np.zeros(self.TABLE_SLOTS, dtype=np.float64)
        check(lib().mm_get_rows(self._h, masks.ctypes.data, sums.ctypes.data, len(masks), C.byref(n)))
        return masks[:n.value].copy(), sums[:n.value].copy()

    def set_row_sums(self, masks: np.ndarray, sums: np.ndarray):
        """-y ratio chaining: start this counter's rows from the sums of the rank holding the preceding block of the file."""
        masks = np.ascontiguousarray(masks, dtype=np.uint64); sums = np.ascontiguousarray(sums, dtype=np.float64)
        check(lib().mm_set_row_sums(self._h, masks.ctypes.data, sums.ctypes.data, len(masks)))

    def resolve_ratio(self):
        """Replay the batches parked by defer_ratio, in order."""
        check(lib().mm_ratio_resolve(self._h))

    def resolve_ratio_chunk(self, chunk: int, n_chunks: int):
        """Replay the parked batches for the rows of one chunk only (see row_chunk); the last chunk releases them."""
        check(lib().mm_ratio_resolve_chunk(self._h, int(chunk), int(n_chunks)))

    # ---- the same chain with the table staying on the device (asynchronous on the context's stream; see include/mmannot_amd.h)
    TABLE_SLOTS = 65536

    def export_table_device(self, keys_ptr: int, sums_ptr: int):
        check(lib().mm_table_export_device(self._h, C.c_void_p(keys_ptr), C.c_void_p(sums_ptr)))

    def export_counts_device(self, keys_ptr: int, counts_ptr: int, stats_ptr: int):
        """-y default / unique: sets, integer counts and the seven counters, device to device on the context's stream (for dist.merge_counts_device)."""
        check(lib().mm_table_export_counts_device(self._h, C.c_void_p(keys_ptr), C.c_void_p(counts_ptr), C.c_void_p(stats_ptr)))

    def merge_counts_device(self, tables_ptr: int, n_tables: int, stride: int, own: int):
        """n_tables exported tables (stride 64-bit words apart) added into this context's table and counters, all but table `own`."""
        check(lib().mm_table_merge_counts_device(self._h, C.c_void_p(tables_ptr), int(n_tables), int(stride), int(own)))

    def import_table_device(self, keys_ptr: int, sums_ptr: int, n: int = 65536, chunk: int = 0, n_chunks: int = 1):
        check(lib().mm_table_import_device(self._h, C.c_void_p(keys_ptr), C.c_void_p(sums_ptr), int(n), int(chunk), int(n_chunks)))

    def ratio_replay(self, first: int = 0, count: int = 1 << 62, chunk: int = 0, n_chunks: int = 1):
        check(lib().mm_ratio_replay(self._h, int(first), int(count), int(chunk), int(n_chunks)))

    def ratio_release(self):
        check(lib().mm_ratio_release(self._h))

    def ratio_reserve(self, n_runs: int, batch_capacity: int):
        check(lib().mm_ratio_reserve(self._h, int(n_runs), int(batch_capacity)))

    def ratio_handoff_to(self, other: "Counter", chunk: int = 0, n_chunks: int = 1):
        check(lib().mm_ratio_handoff(self._h, other._h, int(chunk), int(n_chunks)))

    def ratio_pending(self) -> int:
        return int(lib().mm_ratio_pending(self._h))

    def hit_masks(self, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=np.uint64)
        check(lib().mm_get_hit_masks(self._h, out.ctypes.data, n))
        return out

    def walk_stats(self):
        a = (C.c_uint64 * 2)()
        check(lib().mm_get_walk_stats(self._h, a))
        return int(a[0])

    def probe_stats(self) -> Dict[str, int]:
        """tuples the pipelined probe handed on to the general walk since clear(), and the size of its slot table"""
        a = (C.c_uint64 * 4)()
        check(lib().mm_get_probe_stats(self._h, a))
        return dict(walked=int(a[0]), grouped_batches=int(a[1]), passed_on=int(a[2]), fast_slot_bytes=int(a[3]))

    def probe_path(self) -> str:
        """which probe this counter runs where both apply: 'measuring', 'pipelined' (k_probe_fast) or 'walk' (k_probe)"""
        return {0: "measuring", 1: "pipelined", 2: "walk"}.get(int(lib().mm_probe_path(self._h)), "?")

    def grouped_batches(self) -> int:
        """Batches whose multi-hit records were already grouped by read name (no name sort needed)."""
        a = (C.c_uint64 * 2)()
        check(lib().mm_get_walk_stats(self._h, a))
        return int(a[1])

    PHASES = ("h2d", "probe", "compact", "sort", "merge", "flush")

    def set_profiling(self, on):
        """False/0 off, True/1 every phase, 2 the probe phase only."""
        check(lib().mm_set_profiling(self._h, int(on)))

    def phase_ms(self):
        ms = (C.c_double * 6)()
        nl = (C.c_uint64 * 6)()
        check(lib().mm_get_phase_ms(self._h, ms, nl))
        return {p: (float(ms[i]), int(nl[i])) for i, p in enumerate(self.PHASES)}

    def dump(self, file_name: str) -> str:
        """Counter::dump mmannot.cpp:1806-1818 (the text the reference writes to stderr)."""
        return format_stats(self.stats(), file_name)

    def close(self):
        if getattr(self, "_h", None):
            lib().mm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TableCount:
    """TableCount (mmannot.cpp:1854-1901): one column per input file."""

    def __init__(self, config: Config, n_inputs: int = 1):
        self.config = config
        self.n_inputs = n_inputs
        self.n_columns = 0
        self.rows: Dict[int, List[int]] = {}

    def addCounter(self, counter: Counter):  # mmannot.cpp:1861-1876
        masks, sums = counter.getCounts()
        self.add_rows(masks, sums)

    def add_rows(self, masks, sums):
        for m, s in zip(masks, sums):
            col = self.rows.setdefault(int(m), [0] * self.n_inputs)
            col[self.n_columns] = int(lib().mm_round_to_uint(float(s)))
        self.n_columns += 1

    def sorted_masks(self) -> np.ndarray:
        m = np.array(sorted(self.rows.keys()), dtype=np.uint64)
        if len(m):
            lib().mm_sort_rows(m.ctypes.data, None, len(m))
        return m

    def dump(self, samples: Sequence[str]) -> str:  # mmannot.cpp:1877-1900
        out = ["Type" + "".join("\t" + s for s in samples)]
        for m in self.sorted_masks():
            name = "--".join(self.config.getName(i) for i in range(64) if (int(m) >> i) & 1)
            out.append(name + "".join("\t" + str(v) for v in self.rows[int(m)]))
        return "\n".join(out) + "\n"


def format_stats(stats: Dict[str, int], file_name: str) -> str:
    """Counter::dump + printStats (mmannot.cpp:139-143, 1806-1818) under the comma_numpunct locale (:111-115)."""
    out = [f"Results for {file_name}:"]
    if stats["nHits"] == 0:
        out.append("\tNo hit.")
        return "\n".join(out) + "\n"

    def ps(n: int, s: str, tot: int) -> str:
        size = int(np.log10(float(tot))) + 1 if tot > 0 else 0
        size += int(size / 3.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            pc = np.float32(n) / np.float32(tot) * np.float32(100)
        return "\t" + s + f"{n:,}".rjust(size) + " (" + ("%.1f" % float(pc)).rjust(5) + "%)"

    out.append("\t# reads:                       " + f"{stats['nReads']:,}")
    out.append(ps(stats["nUnique"], "# uniquely mapped reads:       ", stats["nReads"]))
    out.append(ps(stats["nRescued"], "# multi-mapping rescued reads: ", stats["nReads"]))
    out.append("\t# hits:                        " + f"{stats['nHits']:,}")
    out.append(ps(stats["nAmbiguous"], "# ambiguous hits:              ", stats["nHits"]))
    out.append(ps(stats["nUnassigned"], "# unassigned hits:             ", stats["nHits"]))
    return "\n".join(out) + "\n"


class Annotation:
    """Config + IntervalList construction (mmannot.cpp:237-382, 1094-1290) done by the host library:
    configuration file + GTF -> sorted typed intervals (cold path, once per run)."""

    def __init__(self, config_path: str, gtf_path: str, upstream: int = 1000, downstream: int = 1000):
        h = C.c_void_p()
        rc = lib().mm_annotation_load(C.byref(h), config_path.encode(), gtf_path.encode(), upstream, downstream)
        if rc != 0:
            raise _lib.MMError(rc, lib().mm_annotation_error().decode(errors="replace"))
        self._h = h
        l = lib()
        self.log = l.mm_annotation_log(h).decode(errors="replace")
        E = l.mm_annotation_n_elements(h)
        self.names = [l.mm_annotation_element_name(h, e).decode() for e in range(E)]
        nc = l.mm_annotation_n_chromosomes(h)
        self.chromosomes = [l.mm_annotation_chromosome(h, c).decode() for c in range(nc)]
        self.chr_has_intervals = [bool(l.mm_annotation_chromosome_has_intervals(h, c)) for c in range(nc)]
        n = l.mm_annotation_n_intervals(h)
        self.n_intervals = n
        ptrs = [C.c_void_p() for _ in range(5)]
        check(l.mm_annotation_intervals(h, *[C.byref(p) for p in ptrs]))

        def view(p, ct, cnt):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(cnt,)).copy()
        self.chr = view(ptrs[0], C.c_int32, n); self.start = view(ptrs[1], C.c_int32, n); self.end = view(ptrs[2], C.c_int32, n)
        self.type = view(ptrs[3], C.c_uint8, n); self.strand = view(ptrs[4], C.c_uint8, n)
        el = Elements()
        check(l.mm_annotation_elements(h, C.byref(el)))
        self.line_of = view(C.c_void_p(el.line_of), C.c_int32, E)
        self.elem_strand = view(C.c_void_p(el.elem_strand), C.c_uint8, E)
        self.is_up = view(C.c_void_p(el.is_up), C.c_uint8, E)
        self.is_down = view(C.c_void_p(el.is_down), C.c_uint8, E)

    def interval_id(self, i: int) -> str:
        return lib().mm_annotation_interval_id(self._h, i).decode()

    def config(self) -> "Config":
        """The flattened Order as a Config (names are reproduced through source/type of the printed name)."""
        els = []
        for e, nm in enumerate(self.names):
            base = nm[:-4] if nm.endswith((" (+)", " (-)")) else nm
            src, _, typ = base.partition(":")
            els.append(ConfigElement(src, typ, int(self.elem_strand[e]), int(self.line_of[e])))
        return Config(els)

    def close(self):
        if getattr(self, "_h", None):
            lib().mm_annotation_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReadsDecoder:
    """Reader / SamReader / BamReader (mmannot.cpp:1339-1650): file -> mm_aln tuples + NH, in file order."""
    FORMATS = {None: 0, "auto": 0, "sam": 1, "bam": 2}

    def __init__(self, path: str, annotation: Optional[Annotation] = None, fmt=None, strand_compat: int = 0,
                 unique_only: bool = False, keep_names: bool = False):
        h = C.c_void_p()
        rc = lib().mm_reader_open(C.byref(h), path.encode(), self.FORMATS[fmt], annotation._h if annotation is not None else None,
                                  strand_compat, 1 if unique_only else 0, 1 if keep_names else 0)
        self._h = h
        if rc != 0:
            msg = lib().mm_reader_error(h).decode(errors="replace")
            self.close()
            raise _lib.MMError(rc, msg)

    def next(self, cap: int = 1 << 20):
        """-> (tuples, NH of every tuple); the side list of the batch stays in self.last_big"""
        aln = np.empty(cap, dtype=ALN_DTYPE)
        big = np.empty(max(64, min(cap, 1 << 16)), dtype=NH_BIG_DTYPE)
        n, nb, mx = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
        rc = lib().mm_reader_next(self._h, aln.ctypes.data, cap, C.byref(n), big.ctypes.data, len(big), C.byref(nb), C.byref(mx))
        if rc != 0:
            raise _lib.MMError(rc, lib().mm_reader_error(self._h).decode(errors="replace"))
        self.last_big = big[:nb.value].copy()
        self.last_max_nh = int(mx.value)
        a = aln[:n.value]
        return a, aln_nh(a, self.last_big)

    def read_all(self, cap: int = 1 << 20):
        al, nl = [], []
        while True:
            a, h = self.next(cap)
            if len(a) == 0:
                break
            al.append(a.copy()); nl.append(h.copy())
        if not al:
            return np.empty(0, dtype=ALN_DTYPE), np.empty(0, dtype=np.uint32)
        return np.concatenate(al), np.concatenate(nl)

    def qid_bound(self) -> int:
        return int(lib().mm_reader_qid_bound(self._h))

    def n_reads(self) -> int:
        return int(lib().mm_reader_n_reads(self._h))

    def name(self, qid: int) -> str:
        return lib().mm_reader_name(self._h, qid).decode(errors="replace")

    def take_log(self) -> str:
        return lib().mm_reader_take_log(self._h).decode(errors="replace")

    def close(self):
        if getattr(self, "_h", None):
            lib().mm_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
