"""CPU (host logic): the reads decoder on BGZF input (parallel block inflation, mm_reads.cpp:BgzfPipe) must produce exactly the
tuples it produces for the same BAM stored as one plain gzip member (the gzread path, which test_kat / test_synth_e2e pin
against the compiled reference)."""
import gzip
import os

import numpy as np
import pytest

import golden_util as gu
import synth_ref as sr


def _bam_payload(path):
    return gzip.open(path, "rb").read()


@pytest.mark.parametrize("block", [97, 4096, 0xff00])
def test_bgzf_equals_gzip(oracle, tmp_path, block):
    import mmannot_amd.api as api
    # a KAT with XA / NM / Z tags and a synthetic stream
    kat = os.path.join(gu.GOLD, "kat", "X1_xa")
    an = api.Annotation(os.path.join(kat, "c.cfg"), os.path.join(kat, "a.gtf"))
    src = os.path.join(kat, "r.bam")
    b = tmp_path / "x.bam"
    oracle.write_bgzf(str(b), _bam_payload(src), level=6, block=block)
    a0, h0 = api.ReadsDecoder(src, an).read_all()
    a1, h1 = api.ReadsDecoder(str(b), an).read_all(cap=2)
    assert (a0 == a1).all() and (h0 == h1).all() and len(a0) == 7

    rng = np.random.default_rng(3)
    n = 50000
    chroms = [("chr1", 100000), ("chrZ", 50000)]
    body = oracle.bam_records_fixed(rng.integers(-1, 2, n), rng.integers(0, 90000, n), rng.integers(0, 32, n).astype(np.uint16) & 16,
                                    rng.integers(0, 9000, n), rng.integers(20, 150, n), rng.integers(0, 9, n).astype(np.uint32))
    p0, p1 = tmp_path / "g.bam", tmp_path / "b.bam"
    oracle.write_bam_bytes(str(p0), chroms, body, level=1)
    oracle.write_bam_bgzf(str(p1), chroms, body, level=1)
    if block != 0xff00:
        oracle.write_bgzf(str(p1), _bam_payload(str(p0)), level=1, block=block * 13)
    d0, d1 = api.ReadsDecoder(str(p0), an), api.ReadsDecoder(str(p1), an)
    x0, y0 = d0.read_all()
    x1, y1 = d1.read_all(cap=777)
    assert len(x0) == n and (x0 == x1).all() and (y0 == y1).all()
    assert d0.take_log() .replace(str(p0), "F") == d1.take_log().replace(str(p1), "F")
    assert d0.qid_bound() == d1.qid_bound() and d0.name(5) == d1.name(5)


def test_truncated_bgzf_is_an_error(oracle, tmp_path):
    import mmannot_amd.api as api
    from mmannot_amd import MMError
    kat = os.path.join(gu.GOLD, "kat", "E1_overcomplete")
    an = api.Annotation(os.path.join(kat, "c.cfg"), os.path.join(kat, "a.gtf"))
    b = tmp_path / "t.bam"
    oracle.write_bgzf(str(b), _bam_payload(os.path.join(kat, "r.bam")), level=6, block=64)
    raw = open(b, "rb").read()
    open(b, "wb").write(raw[:len(raw) // 2 + 3])
    with pytest.raises(MMError):
        api.ReadsDecoder(str(b), an).read_all()


def _decode(path, an, cap=None):
    import mmannot_amd.api as api
    d = api.ReadsDecoder(path, an)
    a, h = d.read_all() if cap is None else d.read_all(cap=cap)
    return a, h, d.take_log().replace(path, "F"), d.qid_bound(), [d.name(q) for q in range(min(d.qid_bound(), 50))]


def test_fast_record_path_equals_general_parser(oracle, tmp_path, monkeypatch):
    """mm_reads.cpp:fast_bam_record (fields read in place) against the general parser (MM_READER_GENERAL=1), which the KATs and
    the synthetic end-to-end goldens pin against the compiled reference: every known-answer BAM (odd CIGAR operations, NH in
    every tag type, XA, B/H arrays) and a large synthetic one with tags of all plain types, as BGZF and as one gzip member."""
    import mmannot_amd.api as api
    files = []
    for k in sorted(os.listdir(os.path.join(gu.GOLD, "kat"))):
        b = os.path.join(gu.GOLD, "kat", k, "r.bam")
        if os.path.exists(b):
            files.append((k, b, api.Annotation(os.path.join(gu.GOLD, "kat", k, "c.cfg"), os.path.join(gu.GOLD, "kat", k, "a.gtf"))))
    assert len(files) >= 8
    # synthetic: names of varying length, 0..3 CIGAR operations incl. I/S/D, tags A c C s S i I f Z in random order around NH / NM
    import struct
    rng = np.random.default_rng(11)
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    recs = []
    for i in range(120000):                      # long enough for the framed runs to reach the parser threads (>= 8192 records)
        ops = [("MIDS=X"[int(rng.integers(0, 6))], int(rng.integers(1, 90))) for _ in range(int(rng.integers(0, 4)))]
        tags = b""
        for _ in range(int(rng.integers(0, 5))):
            t = "AcCsSiIfZ"[int(rng.integers(0, 9))]
            key = ["NH", "NM", "XS", "ZZ", "AS"][int(rng.integers(0, 5))]
            tags += key.encode() + t.encode()
            if t == "A":
                tags += b"x"
            elif t == "Z":
                tags += b"some text\0"
            elif t == "f":
                tags += struct.pack("<f", 1.5)
            else:
                tags += struct.pack(fmt[t], int(rng.integers(0, 100)))
        recs.append(oracle._bam_record(int(rng.integers(-1, 2)), int(rng.integers(0, 90000)), int(rng.integers(0, 2)) * 16,
                                       "r%d" % int(rng.integers(0, 24000)), ops, int(rng.integers(0, 3)) * 25, tags))
        if i % 9973 == 9972:                     # now and then a record only the general parser takes (N in the CIGAR, XA)
            recs.append(oracle._bam_record(0, 500 + i, 0, "odd%d" % i, [("M", 10), ("N", 100), ("M", 10)], 0, b"NMC\x01XAZchr1,+77,20M,1;\0"))
    chroms = [("chr1", 100000), ("chrZ", 50000)]
    p0, p1 = str(tmp_path / "s.bam"), str(tmp_path / "sb.bam")
    oracle.write_bam(p0, chroms, recs)
    oracle.write_bgzf(p1, _bam_payload(p0), level=1, block=3001)
    an0 = files[0][2]
    files += [("synthetic gzip", p0, an0), ("synthetic bgzf", p1, an0)]
    for label, path, an in files:
        monkeypatch.delenv("MM_READER_GENERAL", raising=False)
        fast = _decode(path, an)
        monkeypatch.setenv("MM_READER_GENERAL", "1")
        gen = _decode(path, an)
        assert (fast[0] == gen[0]).all() and (fast[1] == gen[1]).all(), label
        assert fast[2:] == gen[2:], label


def test_header_larger_than_a_batch_and_cut_records(oracle, tmp_path):
    """The BGZF pipe frames and parses records on its threads only when the BAM header lies inside the first batch of blocks
    (512 x 64 KiB); a longer header text leaves the whole file to the sequential reader.  And with tiny blocks nearly every record
    is cut by a block boundary and many by a batch boundary (the tail moves to the front of the next batch).  Same tuples, names and
    messages as the same payload stored as one gzip member, both ways."""
    import struct
    import mmannot_amd.api as api
    kat = os.path.join(gu.GOLD, "kat", "X1_xa")
    an = api.Annotation(os.path.join(kat, "c.cfg"), os.path.join(kat, "a.gtf"))
    rng = np.random.default_rng(8)
    n = 30000
    chroms = [("chr1", 100000), ("chrZ", 50000)]
    body = oracle.bam_records_fixed(rng.integers(-1, 2, n), rng.integers(0, 90000, n), rng.integers(0, 32, n).astype(np.uint16) & 16,
                                    rng.integers(0, 7000, n), rng.integers(20, 150, n), rng.integers(0, 9, n).astype(np.uint32))
    refs = b"".join(struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln) for nm, ln in chroms)
    for text_len, block in ((0, 61), (34 << 20, 0xff00)):
        text = (b"@CO\t" + b"x" * 59 + b"\n") * (text_len // 64)
        payload = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(chroms)) + refs + body
        pg, pb = str(tmp_path / f"g{block}.bam"), str(tmp_path / f"b{block}.bam")
        with gzip.open(pg, "wb", compresslevel=1) as f:
            f.write(payload)
        oracle.write_bgzf(pb, payload, level=1, block=block)
        a = _decode(pg, an)
        b = _decode(pb, an, cap=4099)
        assert len(a[0]) == n and (a[0] == b[0]).all() and (a[1] == b[1]).all()
        assert a[2:] == b[2:]


def test_corrupted_records_framed_equals_sequential(oracle, tmp_path, monkeypatch):
    """Differential fuzz: a few bytes of the record stream are overwritten (random bytes, bit flips, implausible block sizes).  What
    the reference does with such a file is whatever its sequential parser does; the framed / parallel path must do the same - same
    tuples, names, messages, or the same error - because it hands every record it cannot vouch for to that parser and falls back
    to reading sequentially once the framing and the parser disagree."""
    import struct
    import mmannot_amd.api as api
    from mmannot_amd import MMError
    kat = os.path.join(gu.GOLD, "kat", "X1_xa")
    an = api.Annotation(os.path.join(kat, "c.cfg"), os.path.join(kat, "a.gtf"))
    rng = np.random.default_rng(123)
    n = 20000
    chroms = [("chr1", 100000), ("chrZ", 50000)]
    body = oracle.bam_records_fixed(rng.integers(-1, 2, n), rng.integers(0, 90000, n), rng.integers(0, 32, n).astype(np.uint16) & 16,
                                    rng.integers(0, 5000, n), rng.integers(20, 150, n), rng.integers(0, 9, n).astype(np.uint32))
    refs = b"".join(struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln) for nm, ln in chroms)
    hdr = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", len(chroms)) + refs

    def decode(path):
        try:
            d = api.ReadsDecoder(path, an)
            a, h = d.read_all(cap=3001)
            return ("ok", a.tobytes(), h.tobytes(), d.take_log().replace(path, "F"), d.qid_bound())
        except MMError as e:
            return ("err", str(e).replace(path, "F"))

    outcomes = set()
    for trial in range(14):
        b = bytearray(body)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(b)))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                b[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                b[pos] ^= 1 << int(rng.integers(0, 8))
            else:
                b[pos:pos + 4] = struct.pack("<i", int(rng.integers(-5, 400)))
        path = str(tmp_path / f"fuzz{trial}.bam")
        oracle.write_bgzf(path, hdr + bytes(b), level=1, block=int(rng.choice([3001, 0xff00, 977])))
        monkeypatch.delenv("MM_READER_GENERAL", raising=False)
        fast = decode(path)
        monkeypatch.setenv("MM_READER_GENERAL", "1")
        gen = decode(path)
        monkeypatch.delenv("MM_READER_GENERAL", raising=False)
        assert fast == gen, trial
        outcomes.add(fast[0])
    assert "ok" in outcomes


def _own_inflate(comp: bytes, n: int):
    import ctypes as C
    from mmannot_amd._lib import lib
    out = np.zeros(max(n, 1), dtype=np.uint8)
    buf = np.frombuffer(comp, dtype=np.uint8) if len(comp) else np.zeros(1, dtype=np.uint8)
    rc = lib().mm_inflate_block(buf.ctypes.data, len(comp), out.ctypes.data, n)
    return rc, out[:n].tobytes()


def _inflate_cases(rng, count):
    yield b""
    yield b"a"
    yield b"abc" * 1000
    yield b"\0" * 65280
    yield bytes(rng.integers(0, 256, 60000, dtype=np.uint8))
    yield (b"ACGT" * 25 + bytes(rng.integers(33, 74, 100, dtype=np.uint8))) * 300
    for k in range(count):
        n = int(rng.integers(0, 65281))
        kind = k % 5
        if kind == 0:
            d = rng.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1:
            d = rng.integers(0, int(rng.integers(1, 20)), n, dtype=np.uint8)
        elif kind == 2:
            r = int(rng.integers(1, 40))
            d = np.repeat(rng.integers(0, 256, n // r + 1, dtype=np.uint8), r)[:n]
        elif kind == 3:
            base = rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8)
            d = np.tile(base, n // len(base) + 1)[:n].copy()
            idx = rng.integers(0, max(n, 1), n // 50)
            d[idx[idx < n]] = 7
        else:
            d = rng.normal(128, int(rng.integers(1, 60)), n).astype(np.uint8)
        yield bytes(d)


def test_block_inflate_equals_zlib():
    """The reader's own DEFLATE decoder for BGZF blocks (mm_inflate.h) against zlib: stored, fixed and dynamic blocks, every level and
    strategy, literal-only and run-heavy data, several blocks per stream - every stream zlib wrote must be taken and reproduced."""
    import zlib
    rng = np.random.default_rng(11)
    n_streams = 0
    for data in _inflate_cases(rng, 60):
        for level in (0, 1, 4, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strat)
                comp = co.compress(data) + co.flush()
                rc, out = _own_inflate(comp, len(data))
                assert rc == 0 and out == data, (len(data), level, strat, rc)
                n_streams += 1
    assert n_streams > 1500


def test_block_inflate_refuses_what_zlib_refuses():
    """Damaged, truncated and mis-sized streams: the decoder must give the same verdict as zlib (and the same bytes when both
    accept) - what it refuses goes to zlib in the reader, what it accepts is final."""
    import zlib
    rng = np.random.default_rng(12)
    agree = accepted = 0
    for it, data in enumerate(_inflate_cases(rng, 400)):
        data = data[:20000]
        co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -15, 8, int(rng.integers(0, 5)))
        comp = bytearray(co.compress(data) + co.flush())
        n = len(data)
        mut = it % 5
        if mut == 1 and comp:
            comp[int(rng.integers(0, len(comp)))] ^= 1 << int(rng.integers(0, 8))
        elif mut == 2 and comp:
            comp = comp[:int(rng.integers(0, len(comp)))]
        elif mut == 3 and comp:
            for _ in range(4):
                comp[int(rng.integers(0, len(comp)))] = int(rng.integers(0, 256))
        elif mut == 4:
            n = max(0, n + int(rng.integers(-1, 2)))
        d = zlib.decompressobj(-15)
        try:
            ref = d.decompress(bytes(comp), n + 1)
            ok = d.eof and len(ref) == n
        except zlib.error:
            ok = False
        rc, out = _own_inflate(bytes(comp), n)
        assert (rc == 0) == ok, (it, mut, rc, ok)
        if ok:
            assert out == ref
            accepted += 1
        agree += 1
    assert agree > 400 and accepted > 80


def test_reader_with_zlib_equals_reader_with_own_inflate(oracle, tmp_path, monkeypatch):
    import mmannot_amd.api as api
    kat = os.path.join(gu.GOLD, "kat", "X1_xa")
    an = api.Annotation(os.path.join(kat, "c.cfg"), os.path.join(kat, "a.gtf"))
    rng = np.random.default_rng(5)
    n = 120000
    body = oracle.bam_records_fixed(rng.integers(-1, 2, n), rng.integers(0, 90000, n), rng.integers(0, 32, n).astype(np.uint16) & 16,
                                    rng.integers(0, 9000, n), rng.integers(20, 150, n), rng.integers(0, 9, n).astype(np.uint32), seq_len=60)
    p = tmp_path / "s.bam"
    oracle.write_bam_bgzf(str(p), [("chr1", 100000), ("chrZ", 50000)], body, level=6)
    x0, y0 = api.ReadsDecoder(str(p), an).read_all()
    monkeypatch.setenv("MM_READER_ZLIB", "1")
    x1, y1 = api.ReadsDecoder(str(p), an).read_all()
    assert len(x0) == n and (x0 == x1).all() and (y0 == y1).all()


def test_record_of_several_megabytes_across_a_batch_boundary(oracle, tmp_path):
    """A 9 MB record (6 M bases) that starts 5.7 MB before the end of the BGZF pipe's first batch and a 7.5 MB one later on: the
    unframed tail that moves to the next batch is larger than most batches' whole carry.  Same tuples as the plain-gzip reader,
    which never frames."""
    import mmannot_amd.api as api
    kat = os.path.join(gu.GOLD, "kat", "X1_xa")
    an = api.Annotation(os.path.join(kat, "c.cfg"), os.path.join(kat, "a.gtf"))
    rng = np.random.default_rng(9)
    chroms = [("chr1", 100000), ("chrZ", 50000)]

    def recs(n, q0, seq_len=0):
        return oracle.bam_records_fixed(rng.integers(-1, 2, n), rng.integers(0, 90000, n), rng.integers(0, 32, n).astype(np.uint16) & 16,
                                        q0 + rng.integers(0, 9000, n), rng.integers(20, 150, n), rng.integers(0, 9, n).astype(np.uint32), seq_len=seq_len)
    n1 = 470000                                   # 27.7 MB of 59-byte records: the big one starts 5.7 MB before the end of the first batch (512 blocks of 65280 B)
    body = recs(n1, 0) + recs(1, 100000, seq_len=6_000_000) + recs(3000, 0) + recs(1, 200000, seq_len=5_000_000) + recs(2000, 0)
    p0, p1 = tmp_path / "g.bam", tmp_path / "b.bam"
    oracle.write_bam_bytes(str(p0), chroms, body, level=1)
    oracle.write_bam_bgzf(str(p1), chroms, body, level=1)
    x0, y0 = api.ReadsDecoder(str(p0), an).read_all()
    x1, y1 = api.ReadsDecoder(str(p1), an).read_all()
    assert len(x0) == n1 + 5002 and (x0 == x1).all() and (y0 == y1).all()


def test_more_large_nh_records_than_the_side_list_holds(oracle, tmp_path):
    """Every record with an NH beyond the tuple's 11-bit field takes an entry of the batch's side list (mm_nh_big); the caller's list
    has big_cap entries.  A BGZF input whose parsed runs hold more such records than that must end the batch early instead of writing
    past the list (round-2 ADVICE: heap overflow in mm_reader_next's parsed-mode run; tools/sanitize_host.sh runs this under ASan).
    Tuples, NH values and read-name ids must equal the plain-gzip path's, whatever the batch capacity."""
    import mmannot_amd.api as api
    kat = os.path.join(gu.GOLD, "kat", "E1_overcomplete")
    an = api.Annotation(os.path.join(kat, "c.cfg"), os.path.join(kat, "a.gtf"))
    rng = np.random.default_rng(11)
    n = 150000
    nh = np.where(rng.random(n) < 0.9, rng.integers(2047, 70000, n), rng.integers(0, 2047, n)).astype(np.uint32)
    chroms = [("chr1", 100000)]
    body = oracle.bam_records_fixed(np.zeros(n, dtype=np.int64), rng.integers(0, 90000, n), np.zeros(n, dtype=np.uint16), rng.integers(0, 40000, n),
                                    rng.integers(20, 150, n), nh)
    p0, p1 = tmp_path / "g.bam", tmp_path / "b.bam"
    oracle.write_bam_bytes(str(p0), chroms, body, level=1)
    oracle.write_bam_bgzf(str(p1), chroms, body, level=1)
    x0, y0 = api.ReadsDecoder(str(p0), an).read_all()
    assert len(x0) == n and (y0 == nh).all()
    d = api.ReadsDecoder(str(p1), an)
    a, h = d.next(cap=1 << 20)                       # the wrapper's side list holds 65536 entries: the batch has to stop there
    assert 0 < len(a) < n and len(d.last_big) <= 65536 and len(d.last_big) > 65536 - 64
    assert (h == nh[:len(a)]).all() and (a == x0[:len(a)]).all()
    rest_a, rest_h = [a], [h]
    while True:
        a, h = d.next(cap=1 << 20)
        if len(a) == 0:
            break
        rest_a.append(a.copy()); rest_h.append(h.copy())
    x1, y1 = np.concatenate(rest_a), np.concatenate(rest_h)
    assert len(x1) == n and (x1 == x0).all() and (y1 == y0).all()


def test_bgzf_extra_subfields_and_block_checksums(oracle, tmp_path):
    """Round-2 ADVICE: (a) the 'BC' subfield is looked up in every block's extra field, wherever it stands (a block with further
    subfields is BGZF all the same); (b) every block's CRC-32 is checked - a damaged block that still inflates to ISIZE bytes is an
    error, not a batch of wrong tuples."""
    import mmannot_amd.api as api
    from mmannot_amd import MMError
    kat = os.path.join(gu.GOLD, "kat", "E1_overcomplete")
    an = api.Annotation(os.path.join(kat, "c.cfg"), os.path.join(kat, "a.gtf"))
    rng = np.random.default_rng(5)
    n = 30000
    chroms = [("chr1", 100000)]
    body = oracle.bam_records_fixed(np.zeros(n, dtype=np.int64), rng.integers(0, 90000, n), rng.integers(0, 32, n).astype(np.uint16) & 16,
                                    rng.integers(0, 9000, n), rng.integers(20, 150, n), rng.integers(0, 9, n).astype(np.uint32))
    p0 = tmp_path / "g.bam"
    oracle.write_bam_bytes(str(p0), chroms, body, level=1)
    payload = _bam_payload(str(p0))
    x0, y0 = api.ReadsDecoder(str(p0), an).read_all()
    for k, (before, after) in enumerate(((b"XY\x03\x00abc", b""), (b"", b"ZZ\x00\x00"), (b"Q1\x01\x00x", b"Q2\x02\x00yz"))):
        p = tmp_path / f"x{k}.bam"
        oracle.write_bgzf(str(p), payload, level=1, block=0x8000, extra_before=before, extra_after=after)
        x1, y1 = api.ReadsDecoder(str(p), an).read_all()
        assert len(x1) == n and (x1 == x0).all() and (y1 == y0).all()
    # stored (level 0) blocks: flipping one payload byte leaves a stream that inflates to the same length
    p = tmp_path / "s.bam"
    oracle.write_bgzf(str(p), payload, level=0, block=0x4000)
    x1, _ = api.ReadsDecoder(str(p), an).read_all()
    assert (x1 == x0).all()
    raw = bytearray(open(p, "rb").read())
    at = len(raw) // 2
    # a byte well inside some block's stored data (a name or coordinate byte), not a header: search for a position whose change
    # keeps the file decodable without the checksum
    raw[at] ^= 0x01
    open(p, "wb").write(bytes(raw))
    with pytest.raises(MMError):
        api.ReadsDecoder(str(p), an).read_all()
