"""TEST INFRASTRUCTURE ONLY — ctypes front-end of oracle/mm_oracle.c plus the small
host-side restatements the fixtures need (config descriptor, BAM text <-> tuples).

Nothing in mmannot_amd/ imports this module; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do, and only as the checker.

Reference citations are into /root/reference/mmannot.cpp (v1.1).
"""
from __future__ import annotations

import ctypes as C
import gzip
import os
import struct
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmm_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

STRAND_ALL, STRAND_F, STRAND_R = 0, 1, 2
OVL_INCLUSION, OVL_PC, OVL_NT = 0, 1, 2
STRANDED = {"U": 0, "F": 1, "R": 2}
STRATEGY = {"default": 0, "unique": 1, "random": 2, "ratio": 3}
STAT_NAMES = ("nHits", "nReads", "nUnique", "nAmbiguous", "nMultiple", "nUnassigned", "nRescued")


def build(force: bool = False) -> str:
    """gcc -O2 -shared oracle/mm_oracle.c -> oracle/libmm_oracle.so"""
    src = os.path.join(HERE, "mm_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-shared", "-fPIC", "-o", LIB_PATH, src, "-lm"])
    return LIB_PATH


# --------------------------------------------------------------------------- config descriptor
@dataclass
class Element:
    source: str
    type: str
    strand: int  # STRAND_*
    line: int

    @property
    def name(self) -> str:  # Config::getName mmannot.cpp:445-462
        s = self.source
        if self.type:
            s += ":" + self.type
        if self.strand == STRAND_F:
            s += " (+)"
        elif self.strand == STRAND_R:
            s += " (-)"
        return s


@dataclass
class ConfigDesc:
    """What the hot path needs from Config (mmannot.cpp:219-471): the flattened Order."""
    elements: List[Element] = field(default_factory=list)

    @property
    def E(self) -> int:
        return len(self.elements)

    def names(self) -> List[str]:
        return [e.name for e in self.elements]

    def arrays(self):
        line_of = np.array([e.line for e in self.elements], dtype=np.int32)
        strand = np.array([e.strand for e in self.elements], dtype=np.uint8)
        is_up = np.array([e.type == "upstream" for e in self.elements], dtype=np.uint8)  # :463-466
        is_down = np.array([e.type == "downstream" for e in self.elements], dtype=np.uint8)  # :467-470
        return line_of, strand, is_up, is_down

    def to_json(self):
        return [dict(source=e.source, type=e.type, strand=e.strand, line=e.line) for e in self.elements]

    @staticmethod
    def from_json(js) -> "ConfigDesc":
        return ConfigDesc([Element(**d) for d in js])


def parse_config_order(text: str) -> ConfigDesc:
    """Order section of Config::parse (mmannot.cpp:296-337, 370-374) — enough to derive
    the element table used by scan/getFirst/getName.  (Synonyms/Introns/Vicinity only
    matter to the cold GTF path and are restated in the product's builder tests.)"""
    elements: List[Element] = []
    in_order = False
    line_no = 0
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        if line in ("Synonyms:", "Introns:", "Vicinity:", "Order:"):
            in_order = line == "Order:"
            continue
        if not in_order:
            continue
        for fld in line.split(","):
            strand = STRAND_ALL
            pos = fld.find(" ")
            if pos != -1:  # split(field, ' ', rest, strandStr) :303 (both halves trimmed :165-166)
                rest, s = fld[:pos].strip(), fld[pos + 1:].strip()
                strand = {"+": STRAND_F, "-": STRAND_R}[s]
                fld = rest
            pos = fld.find(":")
            if pos != -1:
                src, typ = fld[:pos].strip(), fld[pos + 1:].strip()
            else:
                src, typ = fld, ""
            elements.append(Element(src, typ, strand, line_no))
        line_no += 1
    return ConfigDesc(elements)


# --------------------------------------------------------------------------- ctypes structs
class _Cfg(C.Structure):
    _fields_ = [("n_elements", C.c_int32), ("line_of", C.c_void_p), ("elem_strand", C.c_void_p),
                ("is_up", C.c_void_p), ("is_down", C.c_void_p), ("overlap_mode", C.c_int32),
                ("overlap_value", C.c_float), ("strandedness", C.c_int32), ("strategy", C.c_int32)]


class _Idx(C.Structure):
    _fields_ = [("n_intervals", C.c_int64), ("n_chromosomes", C.c_int32), ("chr", C.c_void_p),
                ("start", C.c_void_p), ("end", C.c_void_p), ("type", C.c_void_p), ("strand", C.c_void_p)]


class _Reads(C.Structure):
    _fields_ = [("n", C.c_int64), ("tid", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p),
                ("strand", C.c_void_p), ("nh", C.c_void_p), ("qid", C.c_void_p)]


class _Res(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("cap_rows", C.c_int64), ("row_mask", C.c_void_p),
                ("row_sum", C.c_void_p), ("row_count", C.c_void_p), ("stats", C.c_uint64 * 7),
                ("hit_mask", C.c_void_p), ("walked", C.c_int64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.mmo_run.restype = C.c_int
        _lib.mmo_round_to_uint.restype = C.c_uint32
        _lib.mmo_round_to_uint.argtypes = [C.c_double]
        _lib.mmo_cmp_lists.restype = C.c_int
        _lib.mmo_cmp_lists.argtypes = [C.c_uint64, C.c_uint64]
    return _lib


def overlap_mode_of(l: float) -> int:
    """mmannot.cpp:1972-1977 (l is Globals::overlap after stof)."""
    l = np.float32(l)
    if l < 0.0:
        return OVL_INCLUSION
    if l < 1.0:
        return OVL_PC
    return OVL_NT


@dataclass
class OracleResult:
    row_mask: np.ndarray
    row_sum: np.ndarray
    row_count: np.ndarray
    stats: Dict[str, int]
    hit_mask: Optional[np.ndarray]
    walked: int


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def run(cfg: ConfigDesc, index: Dict[str, np.ndarray], reads: Dict[str, np.ndarray], *,
        overlap: float = -1.0, strandedness: str = "F", strategy: str = "default",
        want_hit_mask: bool = False) -> OracleResult:
    """index: chr(int32) start(int64) end(int64) type(int32) strand(uint8) n_chromosomes
    reads: tid(int32) start(int64) end(int64) strand(uint8) nh(uint32) qid(int64)"""
    line_of, es, up, down = cfg.arrays()
    c = _Cfg(cfg.E, _p(line_of), _p(es), _p(up), _p(down), overlap_mode_of(overlap),
             float(np.float32(overlap)), STRANDED[strandedness], STRATEGY[strategy])
    ic = np.ascontiguousarray(index["chr"], dtype=np.int32)
    ist = np.ascontiguousarray(index["start"], dtype=np.int64)
    ien = np.ascontiguousarray(index["end"], dtype=np.int64)
    ity = np.ascontiguousarray(index["type"], dtype=np.int32)
    isd = np.ascontiguousarray(index["strand"], dtype=np.uint8)
    nchr = int(index.get("n_chromosomes", int(ic.max()) + 1 if len(ic) else 0))
    ix = _Idx(len(ic), nchr, _p(ic), _p(ist), _p(ien), _p(ity), _p(isd))
    n = len(reads["start"])
    rt = np.ascontiguousarray(reads["tid"], dtype=np.int32)
    rs = np.ascontiguousarray(reads["start"], dtype=np.int64)
    re_ = np.ascontiguousarray(reads["end"], dtype=np.int64)
    rsd = np.ascontiguousarray(reads["strand"], dtype=np.uint8) if reads.get("strand") is not None else None
    rnh = np.ascontiguousarray(reads["nh"], dtype=np.uint32) if reads.get("nh") is not None else None
    rq = np.ascontiguousarray(reads["qid"], dtype=np.int64) if reads.get("qid") is not None else np.arange(n, dtype=np.int64)
    r = _Reads(n, _p(rt), _p(rs), _p(re_), _p(rsd), _p(rnh), _p(rq))
    cap = 1 << 16
    row_mask = np.zeros(cap, dtype=np.uint64)
    row_sum = np.zeros(cap, dtype=np.float64)
    row_count = np.zeros(cap, dtype=np.uint32)
    hm = np.zeros(n, dtype=np.uint64) if want_hit_mask else None
    res = _Res(0, cap, _p(row_mask), _p(row_sum), _p(row_count), (C.c_uint64 * 7)(), _p(hm), 0)
    rc = lib().mmo_run(C.byref(c), C.byref(ix), C.byref(r), C.byref(res))
    if rc != 0:
        raise RuntimeError(f"mmo_run failed: {rc}")
    k = res.n_rows
    return OracleResult(row_mask[:k].copy(), row_sum[:k].copy(), row_count[:k].copy(),
                        {nm: int(res.stats[i]) for i, nm in enumerate(STAT_NAMES)}, hm, int(res.walked))


# --------------------------------------------------------------------------- table / stats text
def mask_to_list(m: int) -> List[int]:
    return [i for i in range(64) if (int(m) >> i) & 1]


def format_table(cfg: ConfigDesc, row_mask: Sequence[int], row_count: Sequence[int], sample: str) -> str:
    """TableCount::dump mmannot.cpp:1877-1900 for one input column (rows already sorted)."""
    names = cfg.names()
    out = ["Type\t" + sample]
    for m, v in zip(row_mask, row_count):
        out.append("--".join(names[i] for i in mask_to_list(m)) + "\t" + str(int(v)))
    return "\n".join(out) + "\n"


def _commas(n: int) -> str:
    return f"{n:,}"


def format_stats(stats: Dict[str, int], file_name: str) -> str:
    """Counter::dump mmannot.cpp:1806-1818 + printStats :139-143 under comma_numpunct :111-115."""
    out = [f"Results for {file_name}:"]
    if stats["nHits"] == 0:
        out.append("\tNo hit.")
        return "\n".join(out) + "\n"

    def ps(n: int, s: str, tot: int) -> str:
        size = int(np.log10(float(tot))) + 1 if tot > 0 else 0  # unsigned int size = log10(nHits)+1 (tot==0 -> -inf: UB in the reference)
        size += int(size / 3.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            pc = np.float32(n) / np.float32(tot) * np.float32(100)  # static_cast<float>(n)/nHits*100
        return "\t" + s + _commas(n).rjust(size) + " (" + ("%.1f" % float(pc)).rjust(5) + "%)"

    out.append("\t# reads:                       " + _commas(stats["nReads"]))
    out.append(ps(stats["nUnique"], "# uniquely mapped reads:       ", stats["nReads"]))
    out.append(ps(stats["nRescued"], "# multi-mapping rescued reads: ", stats["nReads"]))
    out.append("\t# hits:                        " + _commas(stats["nHits"]))
    out.append(ps(stats["nAmbiguous"], "# ambiguous hits:              ", stats["nHits"]))
    out.append(ps(stats["nUnassigned"], "# unassigned hits:             ", stats["nHits"]))
    return "\n".join(out) + "\n"


# --------------------------------------------------------------------------- reference dumps
def parse_dump(path: str):
    """Parse the MM_DUMP file of oracle/_ref/mmannot*_dump (see oracle/build_ref.sh):
    IV lines (sorted interval vector) and HIT lines (one per hit handed to addCount)."""
    iv = dict(chr=[], chr_name=[], start=[], end=[], type=[], strand=[], id=[])
    hits = dict(name=[], chr_name=[], start=[], end=[], strand=[], nh=[], mask=[])
    with open(path) as f:
        for ln in f:
            p = ln.rstrip("\n").split("\t")
            if p[0] == "IV":
                iv["chr"].append(int(p[2])); iv["chr_name"].append(p[3]); iv["start"].append(int(p[4]))
                iv["end"].append(int(p[5])); iv["type"].append(int(p[6])); iv["strand"].append(int(p[7])); iv["id"].append(p[8])
            elif p[0] == "HIT":
                hits["name"].append(p[1]); hits["chr_name"].append(p[2]); hits["start"].append(int(p[3]))
                hits["end"].append(int(p[4])); hits["strand"].append(int(p[5])); hits["nh"].append(int(p[6]))
                m = 0
                if p[7] != ".":
                    for e in p[7].split(","):
                        m |= 1 << int(e)
                hits["mask"].append(m)
    return iv, hits


def run_ref(variant: str, args: Sequence[str], dump: Optional[str] = None, cwd: Optional[str] = None,
            timeout: int = 600) -> Tuple[int, str, str]:
    """Run oracle/_ref/<variant> (the compiled reference). Returns (rc, stdout, stderr)."""
    exe = os.path.join(REF_DIR, variant)
    env = dict(os.environ)
    env.pop("MM_DUMP", None)
    if dump:
        env["MM_DUMP"] = dump
    p = subprocess.run([exe] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=cwd,
                       timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "mmannot"))


# --------------------------------------------------------------------------- BAM / SAM writers (fixtures)
CIGAR_OPS = "MIDNSHP=X"


def _bam_record(tid: int, pos0: int, flag: int, name: str, cigar: Sequence[Tuple[str, int]], seq_len: int,
                tags: bytes) -> bytes:
    nm = name.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (ln << 4) | CIGAR_OPS.index(op)) for op, ln in cigar)
    body = struct.pack("<iiBBHHHiiii", tid, pos0, len(nm), 255, 4680, len(cigar), flag, seq_len, -1, -1, 0)
    body += nm + cig + b"\x11" * ((seq_len + 1) // 2) + b"\xff" * seq_len + tags
    return struct.pack("<i", len(body)) + body


def tag_nh(nh: Optional[int], typ: str = "C") -> bytes:
    if nh is None:
        return b""
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[typ]
    return b"NH" + typ.encode() + struct.pack(fmt, nh)


def write_bam(path: str, chroms: Sequence[Tuple[str, int]], records: Sequence[bytes], level: int = 1) -> None:
    """Plain-gzip BAM (the reference reads through gzread, mmannot.cpp:1493-1498, so a single
    gzip member is as good as BGZF for it; the product's decoder accepts both)."""
    hdr = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", len(chroms))
    for nm, ln in chroms:
        b = nm.encode() + b"\0"
        hdr += struct.pack("<i", len(b)) + b + struct.pack("<i", ln)
    with gzip.open(path, "wb", compresslevel=level) as f:
        f.write(hdr)
        for r in records:
            f.write(r)


def bam_records_fixed(tid: np.ndarray, pos0: np.ndarray, flag: np.ndarray, qname_id: np.ndarray, read_len: np.ndarray,
                      nh: Optional[np.ndarray], name_width: int = 10, seq_len: int = 0) -> bytes:
    """Vectorised writer for large synthetic BAMs: every record is `<len>M`, name = zero-padded
    decimal of qname_id (fixed width), one NH:I tag (or none).  Returns the record bytes."""
    n = len(tid)
    lname = name_width + 1
    has_nh = nh is not None
    # seq_len = 0 by default (the reference skips (l_seq+1)/2 + l_seq bytes, mmannot.cpp:1558-1559) to keep files small;
    # seq_len > 0 adds random bases and qualities, the bulk of a real BAM record
    nseq = (seq_len + 1) // 2 + seq_len
    rec_len = 32 + lname + 4 + nseq + (7 if has_nh else 0)
    dt = [("block_size", "<i4"), ("tid", "<i4"), ("pos", "<i4"), ("l_read_name", "u1"), ("mapq", "u1"),
          ("bin", "<u2"), ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_tid", "<i4"),
          ("next_pos", "<i4"), ("tlen", "<i4"), ("name", f"S{lname}"), ("cigar", "<u4")]
    if nseq:
        dt += [("seq", "u1", ((seq_len + 1) // 2,)), ("qual", "u1", (seq_len,))]
    if has_nh:
        dt += [("nh_key", "S3"), ("nh_val", "<u4")]
    a = np.zeros(n, dtype=np.dtype(dt))
    assert a.dtype.itemsize == rec_len + 4
    a["block_size"] = rec_len
    a["tid"] = tid; a["pos"] = pos0; a["l_read_name"] = lname; a["mapq"] = 255; a["bin"] = 4680
    a["n_cigar"] = 1; a["flag"] = flag; a["l_seq"] = seq_len; a["next_tid"] = -1; a["next_pos"] = -1
    if nseq:
        rng = np.random.default_rng(12345)
        a["seq"] = rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], dtype=np.uint8),
                              size=(n, (seq_len + 1) // 2))
        a["qual"] = rng.integers(20, 41, size=(n, seq_len), dtype=np.uint8)
    digits = np.zeros((n, lname), dtype=np.uint8)
    q = qname_id.astype(np.int64).copy()
    for k in range(name_width - 1, -1, -1):
        digits[:, k] = 48 + (q % 10); q //= 10
    a["name"] = digits.view(f"S{lname}").reshape(n)
    a["cigar"] = (read_len.astype(np.uint32) << 4)  # op 0 = M
    if has_nh:
        a["nh_key"] = b"NHI"
        a["nh_val"] = nh
    return a.tobytes()


def write_bam_bytes(path: str, chroms: Sequence[Tuple[str, int]], body: bytes, level: int = 1) -> None:
    hdr = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", len(chroms))
    for nm, ln in chroms:
        b = nm.encode() + b"\0"
        hdr += struct.pack("<i", len(b)) + b + struct.pack("<i", ln)
    import zlib
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    with open(path, "wb") as f:
        f.write(co.compress(hdr))
        step = 1 << 26
        for o in range(0, len(body), step):
            f.write(co.compress(body[o:o + step]))
        f.write(co.flush())


def write_bgzf(path: str, payload: bytes, level: int = 1, block: int = 0xff00, extra_before: bytes = b"", extra_after: bytes = b"") -> None:
    """BGZF container (the blocked gzip real BAM files use): independent <= 64 KiB gzip members with a 'BC' extra field
    holding the compressed block size, plus the empty end-of-file block.  extra_before / extra_after: further (well-formed)
    subfields of the gzip extra field around 'BC' - the SAM specification allows them."""
    import zlib
    with open(path, "wb") as f:
        def member(chunk: bytes):
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            xlen = 6 + len(extra_before) + len(extra_after)
            bsize = len(comp) + 12 + xlen + 8 - 1       # total block size - 1
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", xlen) + extra_before + b"BC\x02\x00" + struct.pack("<H", bsize) +
                    extra_after + comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
        for o in range(0, len(payload), block):
            member(payload[o:o + block])
        member(b"")


def write_bam_bgzf(path: str, chroms: Sequence[Tuple[str, int]], body: bytes, level: int = 1) -> None:
    hdr = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", len(chroms))
    for nm, ln in chroms:
        b = nm.encode() + b"\0"
        hdr += struct.pack("<i", len(b)) + b + struct.pack("<i", ln)
    write_bgzf(path, hdr + body, level)


def names_to_qid(names: Sequence[str]) -> np.ndarray:
    d: Dict[str, int] = {}
    out = np.empty(len(names), dtype=np.int64)
    for i, nm in enumerate(names):
        out[i] = d.setdefault(nm, len(d))
    return out
