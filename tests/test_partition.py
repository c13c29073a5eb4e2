"""CPU (host logic): mm_partition - the library-level partitioner that shards ONE input over N devices - and the
sharded oracle: dealing the tuples out by read name, counting every part on its own and adding the tables up must give the
unsharded answer for -y default and -y unique, because no read group is ever split (mmannot.cpp:1669-1701)."""
import ctypes as C

import numpy as np
import pytest

import synth_ref as sr


def _partition(api, lib, aln, big, n_parts, big_cap=1 << 12):
    n = len(aln)
    outs = [np.zeros(max(1, n), dtype=api.ALN_DTYPE) for _ in range(n_parts)]
    bigs = [np.zeros(big_cap, dtype=api.NH_BIG_DTYPE) for _ in range(n_parts)]
    out_n = (C.c_size_t * n_parts)(); out_nb = (C.c_size_t * n_parts)()
    out_mx = (C.c_uint32 * n_parts)(); out_lo = (C.c_uint64 * n_parts)()
    po = (C.c_void_p * n_parts)(*[o.ctypes.data for o in outs]); pb = (C.c_void_p * n_parts)(*[b.ctypes.data for b in bigs])
    rc = lib.mm_partition(aln.ctypes.data, n, big.ctypes.data if len(big) else None, len(big), n_parts, po, out_n, pb, big_cap, out_nb, out_mx, out_lo)
    assert rc == 0
    return [(outs[p][:out_n[p]], bigs[p][:out_nb[p]], int(out_mx[p]), int(out_lo[p])) for p in range(n_parts)]


@pytest.mark.parametrize("threads", [1, 5, 16])
@pytest.mark.parametrize("n_parts", [1, 2, 3, 8, 11])
def test_partition_is_stable_and_renumbers(n_parts, threads):
    """(threads: the batch is cut into per-thread slices, counted, then scattered behind per-part prefix offsets - the result must not
    depend on the number of threads; 11 parts: the division by a run-time value)"""
    import mmannot_amd.api as api
    from mmannot_amd._lib import lib
    assert lib().mm_partition_threads(threads) == threads
    rng = np.random.default_rng(5)
    n = 20003
    qid = rng.integers(0, 3000, n)
    nh = rng.integers(0, 9, n).astype(np.uint64)
    nh[rng.integers(0, n, 40)] = rng.integers(2047, 1 << 32, 40)          # some NH beyond the tuple's field
    aln = api.pack_alignments(rng.integers(-1, 5, n), rng.integers(1, 1 << 20, n), rng.integers(1, 1 << 20, n), rng.integers(0, 2, n), qid, nh)
    big = api.nh_side_list(nh)
    parts = _partition(api, lib(), aln, big, n_parts)
    assert sum(len(p[0]) for p in parts) == n
    for p, (a, b, mx, lo) in enumerate(parts):
        sel = qid % n_parts == p
        src = aln[sel]
        assert (a["start"] == src["start"]).all() and (a["end"] == src["end"]).all() and (a["tsn"] == src["tsn"]).all()   # file order kept
        assert (a["qid"] == src["qid"] // n_parts).all()                    # dense ids per part
        want_nh = nh[sel]
        assert (api.aln_nh(a, b) == np.minimum(want_nh, (1 << 32) - 1).astype(np.uint32)).all()      # the side list follows its tuples
        assert mx == (int(want_nh.max()) if len(want_nh) else 0)
        multi = want_nh > 1
        assert lo == (int((src["qid"][multi] // n_parts).min()) if multi.any() else 0)


@pytest.mark.parametrize("strategy", ["default", "unique"])
@pytest.mark.parametrize("n_parts", [2, 5])
def test_sharded_counts_add_up(oracle, strategy, n_parts):
    import mmannot_amd.api as api
    from mmannot_amd._lib import lib
    rng = np.random.default_rng(17)
    desc = sr.random_config(rng, 6, 3)
    ix = sr.random_index(rng, desc.E, nchr=3, n=2000, maxpos=100000)
    reads = sr.random_reads(rng, ix, n=15000, nchr=3, maxpos=100000, nh_max=7)      # malformed groups included
    ref = oracle.run(desc, ix, reads, overlap=-1.0, strandedness="F", strategy=strategy)
    aln = api.pack_alignments(reads["tid"], reads["start"], reads["end"], reads["strand"], reads["qid"], reads["nh"])
    total, stats = {}, None
    for a, b, _, _ in _partition(api, lib(), aln, api.nh_side_list(reads["nh"]), n_parts):
        part = dict(tid=api.aln_tid(a).astype(np.int32), start=a["start"].astype(np.int64), end=a["end"].astype(np.int64), strand=api.aln_strand(a),
                    nh=api.aln_nh(a, b), qid=a["qid"].astype(np.int64))
        r = oracle.run(desc, ix, part, overlap=-1.0, strandedness="F", strategy=strategy)
        for m, c in zip(r.row_mask.tolist(), r.row_count.tolist()):
            total[m] = total.get(m, 0) + c
        stats = r.stats if stats is None else {k: stats[k] + r.stats[k] for k in stats}
    assert stats == ref.stats
    assert total == dict(zip(ref.row_mask.tolist(), ref.row_count.tolist()))


def test_row_chunk_matches_the_library():
    import mmannot_amd.api as api
    from mmannot_amd._lib import lib
    m = np.random.default_rng(1).integers(1, 1 << 63, 2000, dtype=np.uint64)
    for c in (1, 2, 5, 8, 13):
        assert api.row_chunk(m, c).tolist() == [lib().mm_row_chunk(int(x), c) for x in m]


def _unpack12(api, pk, qid_base):
    """numpy twin of k_expand12 (mm_device.hip)"""
    w = pk["w"].astype(np.uint64)
    start = (w[:, 0] & 0x7FFFFFFF).astype(np.int64)
    ln = (w[:, 1] & 0x3FFF).astype(np.int64)
    nh8 = (w[:, 1] >> 14) & 0xFF
    tid1 = (w[:, 1] >> 22) | ((w[:, 2] >> 24) << 10)
    nh = np.where(nh8 == 255, api.NH_ESCAPE, nh8)
    out = np.empty(len(pk), dtype=api.ALN_DTYPE)
    out["start"] = start; out["end"] = start + ln - 1
    out["tsn"] = ((w[:, 0] >> 31) | (nh << 1) | (tid1 << (api.NH_BITS + 1))).astype(np.uint32)
    out["qid"] = (qid_base + (w[:, 2] & 0xFFFFFF)).astype(np.uint32)
    return out


def test_pack12_round_trip_and_refusals():
    """The optional 12-byte link format: packing is lossless (expanded tuples + side list give back start, end - also start-1, a record
    without CIGAR operations -, chromosome, strand, NH, qid) and a batch with a tuple that does not fit is refused as a whole."""
    import mmannot_amd.api as api
    from mmannot_amd import MMError
    rng = np.random.default_rng(12)
    n = 50000
    start = rng.integers(1, 1 << 30, n)
    ln = rng.integers(0, 16384, n); ln[:50] = 0; ln[50:100] = 16383
    nh = rng.integers(0, 300, n).astype(np.uint64); nh[rng.integers(0, n, 30)] = rng.integers(2047, 1 << 32, 30)
    tid = rng.integers(-1, (1 << 18) - 2, n)
    qid = 5_000_000 + rng.integers(0, 1 << 24, n)
    aln = api.pack_alignments(tid, start, start + ln - 1, rng.integers(0, 2, n), qid, nh)
    big = api.nh_side_list(nh)
    pk, big12 = api.pack12(aln, big, qid_base=5_000_000)
    assert pk.dtype.itemsize == 12 and len(big12) == int((nh >= 255).sum())
    back = _unpack12(api, pk, 5_000_000)
    assert (back["start"] == aln["start"]).all() and (back["end"] == aln["end"]).all() and (back["qid"] == aln["qid"]).all()
    assert (api.aln_tid(back) == api.aln_tid(aln)).all() and (api.aln_strand(back) == api.aln_strand(aln)).all()
    assert (api.aln_nh(back, big12) == api.aln_nh(aln, big)).all()
    for field, value in (("end", int(aln["start"][7]) + 16384), ("qid", 5_000_000 + (1 << 24)), ("qid", 4_999_999)):
        bad = aln.copy(); bad[field][7] = value
        with pytest.raises(MMError):
            api.pack12(bad, big, qid_base=5_000_000)
    bad = api.pack_alignments([(1 << 18) - 1], [10], [20], [1], [5_000_000], [1])      # chromosome index beyond 18 bits
    with pytest.raises(MMError):
        api.pack12(bad, None, qid_base=5_000_000)
