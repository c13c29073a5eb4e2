"""GPU: the HIP path (through the C ABI) against (a) the golden vectors of the compiled reference
and (b) the CPU oracle on seeded synthetic inputs — bit-exact: tables, the seven counters, and the
per-hit element sets."""
import json
import os

import numpy as np
import pytest

import golden_util as gu
import synth_ref as sr

pytestmark = pytest.mark.gpu

ALL1 = np.uint64(0xFFFFFFFFFFFFFFFF)


def to_cfg(mm, desc):
    return mm.Config.from_json(desc.to_json())


def run_gpu(mm, cfg, ix, reads, *, overlap, strandedness, strategy, batches=1, keep=True, bin_shift=0):
    index = mm.IntervalIndex(cfg, ix["chr"], ix["start"], ix["end"], ix["type"], ix["strand"], ix["n_chromosomes"], bin_shift=bin_shift)
    ctr = mm.Counter(index, overlap=overlap, strandedness=strandedness, strategy=strategy, keep_hit_masks=keep)
    n = len(reads["start"])
    aln = mm.pack_alignments(reads["tid"], reads["start"], reads["end"], reads["strand"], reads["qid"])
    nh = reads["nh"].astype(np.uint32)
    qb = int(reads["qid"].max()) + 1 if n else 1
    hm = []
    cuts = np.linspace(0, n, batches + 1).astype(int)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            ctr.add_batch(aln[a:b], nh[a:b], qb)
            ctr.sync()
            if keep:
                hm.append(ctr.hit_masks(b - a))
    ctr.flush()
    masks, sums = ctr.getCounts()
    stats = ctr.stats()
    table = mm.TableCount(cfg, 1)
    table.add_rows(masks, sums)
    res = dict(table=table.dump(["test_dataset"]), stats=stats, hit=np.concatenate(hm) if hm else None, masks=masks, sums=sums,
               stderr=ctr.dump("$REF/test_dataset.bam"))
    ctr.close(); index.close()
    return res


D, CASES, READS, HITMASKS = gu.load_test_dataset()
NAMES = [k for k in CASES if not k.startswith("fail") and "random" not in k]


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("batches", [1, 7])
def test_reference_golden(gpu_lib, name, batches):
    mm = gpu_lib
    c = CASES[name]
    o = gu.parse_args(c["args"])
    cfg = mm.Config.from_json(json.load(open(os.path.join(D, f"config_{c['config']}.json"))))
    ix = gu.load_index(os.path.join(D, f"index_{c['index']}.tsv.gz"))
    r = run_gpu(mm, cfg, ix, READS, overlap=o["overlap"], strandedness=o["strandedness"], strategy=o["strategy"], batches=batches)
    assert r["table"] == c["table"]
    for k, v in gu.parse_stderr_stats(c["stderr"]).items():
        assert r["stats"][k] == v, k
    assert r["stderr"] in c["stderr"]
    m = r["hit"][r["hit"] != ALL1]
    assert len(m) == len(HITMASKS[name]) and (m == HITMASKS[name]).all()
    r = run_gpu(mm, cfg, ix, READS, overlap=o["overlap"], strandedness=o["strandedness"], strategy=o["strategy"], batches=batches, keep=False)
    assert r["table"] == c["table"] and r["stderr"] in c["stderr"]   # the instance compiled for the launch (no side output)


SYNTH = [(seed, l, s, y) for seed in (1, 2, 3) for l in (-1.0, 0.0, 0.4, 0.83, 1.0, 25.0) for s in ("U", "F", "R")
         for y in ("default", "unique", "ratio")]


@pytest.mark.parametrize("seed,l,s,y", SYNTH)
def test_oracle_synthetic(gpu_lib, oracle, seed, l, s, y):
    mm = gpu_lib
    rng = np.random.default_rng(1000 * seed + int(abs(l) * 10))
    desc = sr.random_config(rng, n_lines=int(rng.integers(3, 12)), max_per_line=int(rng.integers(1, 5)))
    ix = sr.random_index(rng, desc.E, nchr=4, n=3000, maxpos=150000, skip_chr=3 if seed == 2 else None)
    reads = sr.random_reads(rng, ix, n=20000, nchr=4, maxpos=150000, nh_max=7)
    ref = oracle.run(desc, ix, reads, overlap=l, strandedness=s, strategy=y, want_hit_mask=True)
    cfg = to_cfg(mm, desc)
    for batches, shift in ((1, 0), (5, 8), (3, 14)):
        r = run_gpu(mm, cfg, ix, reads, overlap=l, strandedness=s, strategy=y, batches=batches, bin_shift=shift)
        assert (r["hit"] == ref.hit_mask).all(), f"hit masks differ (batches={batches}, shift={shift})"
        assert r["stats"] == ref.stats
        assert r["table"] == oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset")
        if y == "ratio":  # the double accumulators themselves must be bit-identical (file-order adds)
            got = dict(zip(r["masks"].tolist(), r["sums"].view(np.uint64).tolist()))
            want = dict(zip(ref.row_mask.tolist(), ref.row_sum.view(np.uint64).tolist()))
            assert got == want
    # without the per-hit side output k_probe runs the instance compiled for the launch (counting only / multi-hit records with or
    # without the fused histogram / ratio) instead of the general one: same table, same counters, same sums
    want_table = oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset")
    for batches, shift in ((1, 0), (4, 9)):
        r = run_gpu(mm, cfg, ix, reads, overlap=l, strandedness=s, strategy=y, batches=batches, bin_shift=shift, keep=False)
        assert r["stats"] == ref.stats and r["table"] == want_table, f"specialised instance differs (batches={batches}, shift={shift})"
        if y == "ratio":
            assert dict(zip(r["masks"].tolist(), r["sums"].view(np.uint64).tolist())) == dict(zip(ref.row_mask.tolist(), ref.row_sum.view(np.uint64).tolist()))
    if seed == 1 and s == "U":   # and the general instance forced onto the same launch
        os.environ["MMANNOT_AMD_GENERAL_PROBE"] = "1"
        try:
            r = run_gpu(mm, cfg, ix, reads, overlap=l, strandedness=s, strategy=y, batches=2, keep=False)
        finally:
            del os.environ["MMANNOT_AMD_GENERAL_PROBE"]
        assert r["stats"] == ref.stats and r["table"] == want_table
        os.environ["MMANNOT_AMD_FULL_INDEX_ONLY"] = "1"      # and every read through the full 16-byte index (the path of reads without CIGAR operations)
        try:
            r = run_gpu(mm, cfg, ix, reads, overlap=l, strandedness=s, strategy=y, batches=2)
        finally:
            del os.environ["MMANNOT_AMD_FULL_INDEX_ONLY"]
        assert (r["hit"] == ref.hit_mask).all() and r["stats"] == ref.stats and r["table"] == want_table


def test_name_grouped_input(gpu_lib, oracle):
    """An aligner's own output keeps the hits of a read adjacent, so read-name ids are non-decreasing and the name sort is
    skipped (k_rs_hist's order check; it matters when the batch's key range is too wide for the one-pass leaf layout).  Same answers as the oracle, malformed groups included; and one inversion anywhere -
    inside a wave, or across the wave (64) / round (512) / tile (8192) boundaries of the check - must send the batch down
    the general path and still give the oracle's answer."""
    mm = gpu_lib
    rng = np.random.default_rng(99)
    desc = sr.random_config(rng, n_lines=6, max_per_line=3)
    ix = sr.random_index(rng, desc.E, nchr=3, n=2500, maxpos=120000)
    cfg = to_cfg(mm, desc)
    index = mm.IntervalIndex(cfg, ix["chr"], ix["start"], ix["end"], ix["type"], ix["strand"], ix["n_chromosomes"])

    def gpu(reads, batches):
        ctr = mm.Counter(index, overlap=-1.0, strandedness="U", strategy="default")
        aln = mm.pack_alignments(reads["tid"], reads["start"], reads["end"], reads["strand"], reads["qid"])
        cuts = np.linspace(0, len(aln), batches + 1).astype(int)
        for a, b in zip(cuts[:-1], cuts[1:]):
            ctr.add_batch(aln[a:b], reads["nh"][a:b].astype(np.uint32), int(reads["qid"].max()) + 1)
        ctr.flush()
        masks, sums = ctr.getCounts()
        out = (dict(zip(masks.tolist(), sums.tolist())), ctr.stats(), ctr.grouped_batches())
        ctr.close()
        return out

    for nh_min in (1, 2):
        reads = sr.random_reads(rng, ix, n=21000, nchr=3, maxpos=120000, nh_max=7, shuffle=False, nh_min=nh_min)
        # (ids spread out: within 21 key bits the library takes the partition + k_leaf path without looking at the order at all)
        reads = dict(reads, qid=reads["qid"] * 1000)
        assert (np.diff(reads["qid"]) >= 0).all() and int(reads["qid"].max()) > (1 << 22)
        ref = oracle.run(desc, ix, reads, overlap=-1.0, strandedness="U", strategy="default")
        for batches in (1, 4):
            rows, st, grouped = gpu(reads, batches)
            assert st == ref.stats and rows == dict(zip(ref.row_mask.tolist(), ref.row_count.astype(float).tolist()))
            assert grouped == batches
        q, nh = reads["qid"], reads["nh"]
        for pos in (1, 37, 63, 64, 65, 255, 256, 257, 511, 512, 513, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 16383, 16384, 20990):
            j = pos
            while j < len(q) and not (q[j] != q[j - 1] and nh[j] > 1 and nh[j - 1] > 1):   # both records in the multi-hit pipeline (NH 0 and 1 are not)
                j += 1
            if j >= len(q):
                continue
            r2 = {k: v.copy() for k, v in reads.items()}
            for v in r2.values():
                v[[j - 1, j]] = v[[j, j - 1]]
            ref2 = oracle.run(desc, ix, r2, overlap=-1.0, strandedness="U", strategy="default")
            rows, st, grouped = gpu(r2, 1)
            assert grouped == 0, pos
            assert st == ref2.stats and rows == dict(zip(ref2.row_mask.tolist(), ref2.row_count.astype(float).tolist())), pos
    index.close()


@pytest.mark.parametrize("shuffle", [True, False])
@pytest.mark.parametrize("vals", ["beyond_payload", "beyond_field"])
def test_huge_nh_payload_by_index(gpu_lib, oracle, shuffle, vals):
    """NH travels in an 11-bit field of the tuple; larger values go through the batch's side list (mm_nh_big).  A batch holding
    an NH >= 65535 cannot carry (slot, NH) in the 32-bit sort payload either: the payload becomes the record index and the
    merge kernel reads the element set and NH through it (INFO_BIGNH).  Same answers as the oracle, in interleaved and in
    name-grouped order, across batches with and without such records."""
    mm = gpu_lib
    rng = np.random.default_rng(4242)
    desc = sr.random_config(rng, n_lines=5, max_per_line=3)
    ix = sr.random_index(rng, desc.E, nchr=3, n=2000, maxpos=100000)
    reads = sr.random_reads(rng, ix, n=24000, nchr=3, maxpos=100000, nh_max=6, shuffle=shuffle)
    nh = reads["nh"].copy()
    q = reads["qid"]
    # whole names get the value; 2046 is the largest NH the tuple's own field holds
    plan = {"beyond_payload": ((100, 70000), (9000, 65535), (17000, 4000000000), (3000, 2046), (5000, 2047), (12000, 5000), (20000, 65534)),
            "beyond_field": ((100, 2046), (3000, 2047), (9000, 2048), (12000, 5000), (17000, 65534), (20000, 40000))}[vals]
    for pos, val in plan:
        nh[q == int(q[pos])] = val
    reads = dict(reads, nh=nh)
    assert len(mm.nh_side_list(nh)) > 0
    for strategy in ("default", "ratio"):
        ref = oracle.run(desc, ix, reads, overlap=-1.0, strandedness="F", strategy=strategy)
        cfg = to_cfg(mm, desc)
        for batches in (1, 3):
            r = run_gpu(mm, cfg, ix, reads, overlap=-1.0, strandedness="F", strategy=strategy, batches=batches, keep=False)
            assert r["stats"] == ref.stats
            assert r["table"] == oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset")


def test_nh_side_list_errors(gpu_lib):
    """The side list is validated: an escaped tuple without an entry, and MM_BATCH_SINGLE_HIT broken, are errors at the next sync."""
    mm = gpu_lib
    rng = np.random.default_rng(3)
    desc = sr.random_config(rng, 3, 1)
    ix = sr.random_index(rng, desc.E, nchr=1, n=20, maxpos=20000)
    index = mm.IntervalIndex(to_cfg(mm, desc), ix["chr"], ix["start"], ix["end"], ix["type"], ix["strand"], 1)
    n = 100
    nh = np.full(n, 3, dtype=np.uint32); nh[17] = 9999
    aln = mm.pack_alignments(np.zeros(n), np.full(n, 100), np.full(n, 150), np.ones(n), np.arange(n), nh)
    ctr = mm.Counter(index)
    ctr.add_batch(aln, None, n)                     # tuple 17 escapes, no side list given
    with pytest.raises(mm.MMError):
        ctr.sync()
    ctr.close()
    ctr = mm.Counter(index)
    ctr.add_batch(aln, nh, n, single_hit=False)     # with the list: fine
    ctr.sync(); ctr.close()
    ctr = mm.Counter(index)
    with pytest.raises(mm.MMError):                 # a side list together with the single-hit promise is refused at once
        ctr.add_batch(aln, nh, n, single_hit=True)
    ctr.add_batch(mm.pack_alignments(np.zeros(n), np.full(n, 100), np.full(n, 150), np.ones(n), np.arange(n), np.full(n, 2)), None, n, single_hit=True)
    with pytest.raises(mm.MMError):                 # promise broken: NH = 2 everywhere
        ctr.sync()
    ctr.close()
    two = mm.pack_alignments(np.zeros(n), np.full(n, 100), np.full(n, 150), np.ones(n), np.arange(n) + 1000, np.full(n, 2))
    ctr = mm.Counter(index)
    ctr.add_batch(two, None, 1000 + n, qid_lo=1000)   # the hint holds: fine (and the host takes the leaf layout for granted)
    ctr.sync(); ctr.close()
    ctr = mm.Counter(index)
    ctr.add_batch(two, None, 1000 + n, qid_lo=1050)   # 50 multi-hit tuples below the promised lower bound
    with pytest.raises(mm.MMError):
        ctr.sync()
    ctr.close()
    ctr = mm.Counter(index)
    wide = mm.pack_alignments(np.zeros(n), np.full(n, 100), np.full(n, 150), np.ones(n), np.arange(n) * 40000, np.full(n, 2))
    ctr.reserve(n * 40000 + 1)
    ctr.add_batch(wide, None, 2000, qid_lo=0)          # ids far beyond the bound the launches were sized for: refused, not miscounted
    with pytest.raises(mm.MMError):
        ctr.sync()
    ctr.close(); index.close()


def test_empty_and_edge_batches(gpu_lib, oracle):
    mm = gpu_lib
    rng = np.random.default_rng(77)
    desc = sr.random_config(rng, 4, 2)
    ix = sr.random_index(rng, desc.E, nchr=2, n=50, maxpos=5000)
    cfg = to_cfg(mm, desc)
    index = mm.IntervalIndex(cfg, ix["chr"], ix["start"], ix["end"], ix["type"], ix["strand"], 2)
    ctr = mm.Counter(index, strandedness="U")
    ctr.add_batch(np.zeros(0, dtype=mm.ALN_DTYPE), np.zeros(0, dtype=np.uint32), 1)   # empty batch
    ctr.flush()
    assert ctr.stats()["nHits"] == 0 and len(ctr.getCounts()[0]) == 0
    assert "No hit." in ctr.dump("x.bam")
    with pytest.raises(mm.MMError):            # Counter::read after the flush needs clear()
        ctr.add_batch(np.zeros(1, dtype=mm.ALN_DTYPE), None, 1)
    ctr.clear()
    # a single read, NH absent (NULL nh array = all 1)
    s = int(ix["start"][0]); e = int(ix["start"][0])
    ctr.add_batch(mm.pack_alignments([int(ix["chr"][0])], [s], [e], [1], [0]), None, 1)
    ctr.flush()
    ref = oracle.run(desc, ix, dict(tid=np.array([ix["chr"][0]], dtype=np.int32), start=np.array([s]), end=np.array([e]),
                                    strand=np.array([1], dtype=np.uint8), nh=None, qid=np.array([0])), strandedness="U")
    assert ctr.stats() == ref.stats
    with pytest.raises(mm.MMError):            # -y random is not a device strategy
        mm.Counter(index, strategy="random")
    ctr.close(); index.close()


def test_synth_generator_twin(gpu_lib):
    """mm_synth_fill (device) == tests/synth_ref.synth_tuples (numpy), so full-size device-generated
    workloads can be checked against the oracle on a prefix."""
    import ctypes as C
    import torch
    mm = gpu_lib
    from mmannot_amd import _lib
    chr_len = np.array([30_000_000, 19_000_000, 23_000_000, 18_000_000, 27_000_000], dtype=np.int64)
    for nh_max, sc, first, pair in ((1, 0, 0, 0), (20, 0, 12345, 0), (20, 10, 1 << 20, 0), (5, 16, 777, 0), (1, 0, 4321, 250)):
        n = 100_000
        aln = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        p = _lib.SynthParams(42, first, 100, nh_max, len(chr_len), chr_len.ctypes.data, sc, pair)
        _lib.check(_lib.lib().mm_synth_fill(aln.data_ptr(), n, C.byref(p), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        a = np.ascontiguousarray(aln.cpu().numpy()).view(mm.ALN_DTYPE).reshape(-1)
        t = sr.synth_tuples(42, first, n, 100, nh_max, chr_len, sc, pair)
        assert (a["start"].astype(np.int64) == t["start"]).all() and (a["end"].astype(np.int64) == t["end"]).all()
        assert (mm.aln_tid(a) == t["tid"]).all() and (mm.aln_strand(a) == t["strand"]).all()
        assert (a["qid"].astype(np.int64) == t["qid"]).all()
        assert (mm.aln_nh(a) == t["nh"]).all()
        assert a.tobytes() == mm.pack_alignments(t["tid"], t["start"], t["end"], t["strand"], t["qid"], t["nh"]).tobytes()


@pytest.mark.parametrize("nh_max,strand,l", [(100, "F", -1.0), (3, "U", 0.5), (1, "R", -1.0)])
def test_ratio_large_stream(gpu_lib, oracle, nh_max, strand, l):
    """-y ratio at a size where rows receive 10^5..10^6 terms: the parallel binade-integer accumulation (k_ratio_rows) must
    reproduce the reference's sequential double sums bit for bit, across batches of uneven sizes."""
    from mmannot_amd import synth
    mm = gpu_lib
    an = synth.synth_annotation("hs38", seed=9, scale=0.05)
    n = 3_000_000
    rng = np.random.default_rng(5)
    pick = rng.integers(0, len(an.start), n)                 # reads placed on / next to annotated intervals: ~half of them count
    st = np.maximum(1, an.start[pick] + rng.integers(-60, 80, n))
    t = dict(tid=an.chr[pick].astype(np.int32), start=st.astype(np.int64), end=(st + 99).astype(np.int64),
             strand=rng.integers(0, 2, n).astype(np.uint8), nh=rng.integers(1, nh_max + 1, n).astype(np.uint32),
             qid=np.arange(n, dtype=np.int64))
    desc = oracle.ConfigDesc.from_json(an.config)
    ix = an.as_index_dict()
    ref = oracle.run(desc, ix, t, overlap=l, strandedness=strand, strategy="ratio")
    cfg = mm.Config.from_json(an.config)
    index = mm.IntervalIndex(cfg, an.chr, an.start, an.end, an.type, an.strand, an.n_chromosomes)
    ctr = mm.Counter(index, overlap=l, strandedness=strand, strategy="ratio")
    aln = mm.pack_alignments(t["tid"], t["start"], t["end"], t["strand"], t["qid"])
    for a, b in ((0, 1_000_003), (1_000_003, 1_000_004), (1_000_004, 2_500_000), (2_500_000, n)):
        ctr.add_batch(aln[a:b], t["nh"][a:b], int(t["qid"].max()) + 1)
        ctr.sync()
    ctr.flush()
    masks, sums = ctr.getCounts()
    assert ctr.stats() == ref.stats
    got = dict(zip(masks.tolist(), sums.view(np.uint64).tolist()))
    want = dict(zip(ref.row_mask.tolist(), ref.row_sum.view(np.uint64).tolist()))
    assert got == want
    assert max(ref.row_sum) > (1e4 if nh_max > 3 else 1e5)   # rows large enough to leave the serial warm-up
    ctr.close(); index.close()


def test_qid_bound_too_small_is_an_error_not_a_fault(gpu_lib):
    mm = gpu_lib
    rng = np.random.default_rng(1)
    desc = sr.random_config(rng, 3, 1)
    ix = sr.random_index(rng, desc.E, nchr=1, n=20, maxpos=20000)
    index = mm.IntervalIndex(to_cfg(mm, desc), ix["chr"], ix["start"], ix["end"], ix["type"], ix["strand"], 1)
    ctr = mm.Counter(index)
    n = 1000
    aln = mm.pack_alignments(np.zeros(n), np.full(n, 100), np.full(n, 150), np.ones(n), np.arange(n) + 5_000_000)   # ids far beyond the bound
    ctr.add_batch(aln, np.full(n, 2, dtype=np.uint32), 10)
    with pytest.raises(mm.MMError):
        ctr.sync()
    ctr.close(); index.close()


@pytest.mark.parametrize("chunks", [1, 5, "handoff", "handoff5", "device_tensors", "interleaved"])
def test_ratio_chain_across_contexts(gpu_lib, oracle, chunks):
    """Multi-GPU -y ratio in miniature: two contexts hold consecutive blocks of one stream, park their accumulation
    (defer_ratio) and chain the per-row sums; the result must be bit-identical to one context over the whole stream and to
    the CPU oracle."""
    from mmannot_amd import synth
    mm = gpu_lib
    an = synth.synth_annotation("hs38", seed=9, scale=0.05)
    n = 2_400_000
    rng = np.random.default_rng(6)
    pick = rng.integers(0, len(an.start), n)
    st = np.maximum(1, an.start[pick] + rng.integers(-60, 80, n))
    t = dict(tid=an.chr[pick].astype(np.int32), start=st.astype(np.int64), end=(st + 99).astype(np.int64),
             strand=rng.integers(0, 2, n).astype(np.uint8), nh=rng.integers(1, 40, n).astype(np.uint32), qid=np.arange(n, dtype=np.int64))
    desc = oracle.ConfigDesc.from_json(an.config)
    ref = oracle.run(desc, an.as_index_dict(), t, overlap=-1.0, strandedness="F", strategy="ratio")
    cfg = mm.Config.from_json(an.config)
    index = mm.IntervalIndex(cfg, an.chr, an.start, an.end, an.type, an.strand, an.n_chromosomes)
    aln = mm.pack_alignments(t["tid"], t["start"], t["end"], t["strand"], t["qid"])
    cut = 1_000_001
    a = mm.Counter(index, strategy="ratio", defer_ratio=True)
    b = mm.Counter(index, strategy="ratio", defer_ratio=True)
    for lo, hi in ((0, 400_000), (400_000, cut)):
        a.add_batch(aln[lo:hi], t["nh"][lo:hi], n); a.sync()
    for lo, hi in ((cut, 1_700_000), (1_700_000, n)):
        b.add_batch(aln[lo:hi], t["nh"][lo:hi], n); b.sync()
    a.flush(); b.flush()
    if chunks == "interleaved":   # mm_merge's order for batches dealt round-robin (a0 b0 a1 b1): tables handed over per batch, all on the devices
        a.close(); b.close()
        a = mm.Counter(index, strategy="ratio", defer_ratio=True); b = mm.Counter(index, strategy="ratio", defer_ratio=True)
        a.ratio_reserve(2, 800_000); b.ratio_reserve(2, 800_000)                # (buffers for the parked batches ahead of time)
        cuts = (0, 400_000, cut, 1_700_000, n)
        for k, ctr in enumerate((a, b, a, b)):
            ctr.add_batch(aln[cuts[k]:cuts[k + 1]], t["nh"][cuts[k]:cuts[k + 1]], n)
        a.sync(); b.sync()
        assert a.ratio_pending() == 2 and b.ratio_pending() == 2
        a.ratio_replay(0, 1); a.ratio_handoff_to(b); b.ratio_replay(0, 1); b.ratio_handoff_to(a)
        a.ratio_replay(1, 1); a.ratio_handoff_to(b); b.ratio_replay(1, 1)
        a.ratio_release(); b.ratio_release()
        assert a.ratio_pending() == 0 and b.ratio_pending() == 0
    elif chunks in ("handoff", "handoff5"):   # the chain without a host hop: mm_ratio_replay + mm_ratio_handoff (a peer copy between devices; here one device)
        nc = 5 if chunks == "handoff5" else 1
        for c in range(nc):
            a.ratio_replay(0, 1 << 40, c, nc)
            a.ratio_handoff_to(b, c, nc)
            b.ratio_replay(0, 1 << 40, c, nc)
        a.ratio_release(); b.ratio_release()
    elif chunks == "device_tensors":          # ... and through a buffer of the caller's (what dist.chain_ratio sends over RCCL)
        import torch
        buf = torch.zeros(2 * mm.Counter.TABLE_SLOTS, dtype=torch.int64, device="cuda")
        a.ratio_replay()
        a.export_table_device(buf.data_ptr(), buf.data_ptr() + 8 * mm.Counter.TABLE_SLOTS)
        a.sync()
        b.import_table_device(buf.data_ptr(), buf.data_ptr() + 8 * mm.Counter.TABLE_SLOTS)
        b.ratio_replay()
        a.ratio_release(); b.ratio_release()
    elif chunks == 1:
        a.resolve_ratio()
        ma, sa = a.getCounts()
        b.set_row_sums(ma, sa)
        b.resolve_ratio()
    else:   # the pipelined chain of mmannot_amd/dist.py in miniature: chunk by chunk (rows are independent of each other)
        assert sorted(set(mm.row_chunk(np.arange(1, 5000, dtype=np.uint64), chunks).tolist())) == list(range(chunks))
        from mmannot_amd._lib import lib
        assert all(lib().mm_row_chunk(int(m), chunks) == int(mm.row_chunk(np.array([m], dtype=np.uint64), chunks)[0]) for m in (1, 2, 3, 1 << 40, 1516545))
        for c in range(chunks):
            a.resolve_ratio_chunk(c, chunks)
            ma, sa = a.getCounts()
            sel = mm.row_chunk(ma, chunks) == c
            b.set_row_sums(ma[sel], sa[sel])
            b.resolve_ratio_chunk(c, chunks)
    mb, sb = b.getCounts()
    sta, stb = a.stats(), b.stats()
    assert {k: sta[k] + stb[k] for k in sta} == ref.stats
    got = dict(zip(mb.tolist(), sb.view(np.uint64).tolist()))
    want = dict(zip(ref.row_mask.tolist(), ref.row_sum.view(np.uint64).tolist()))
    assert got == want
    a.close(); b.close(); index.close()


def _home_slot(mask):
    """hash64 of mm_device.hip: home slot of an element set in the 65,536-slot device table."""
    return ((int(mask) * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)) >> 40 & 0xFFFF


def test_table_slot_65535_is_never_a_payload(gpu_lib, oracle):
    """A multi-hit record's element set travels to the merge kernel as its 16-bit table slot and 0xFFFF there means "no
    element".  Sets whose home slot is 65535 ({0,10,13,16,17,18,20}, {20,39,49}) or that linear probing pushes there (two
    sets with home 65534) must still reach their group's union (mmannot.cpp:1678-1690): one Order line of 21 tied elements,
    -l -1, NH = 2 reads inside intervals of exactly those elements.  New sets take the table's first 512 slots while there is
    room, so 2000 other sets come first (single-hit reads, in batches of their own) - the sets of interest then go to their home
    slots in the rest of the table.  The same input under -y ratio sorts by slots on both sides of 512: two sort passes."""
    import synth_ref  # noqa: F401  (path set-up)
    from oracle import mm_oracle as mo
    mm = gpu_lib
    E = 21
    desc = mo.ConfigDesc([mo.Element("s", f"t{e}", 0, 0) for e in range(E)])
    target = sum(1 << e for e in (0, 10, 13, 16, 17, 18, 20))
    assert target == 1516545 and _home_slot(target) == 65535
    allm = np.arange(1, 1 << E, dtype=np.uint64)
    home = ((allm * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) & np.uint64(0xFFFF)
    at_last = [target] + [int(m) for m in allm[home == 65535] if bin(int(m)).count("1") >= 2 and int(m) != target][:2]
    before_last = [int(m) for m in allm[home == 65534] if bin(int(m)).count("1") >= 2][:3]   # the 2nd and 3rd probe on to 65535
    assert target in at_last and len(before_last) == 3
    sets = at_last + before_last + [0b11, 1 << 5]
    rng = np.random.default_rng(65535)
    fillers = []
    while len(fillers) < 2000:
        m = int(rng.integers(3, 1 << E))
        if bin(m).count("1") >= 2 and m not in sets and m not in fillers:
            fillers.append(m)
    regions = sets + fillers
    chr_, start, end, typ = [], [], [], []
    for j, m in enumerate(regions):
        base = 10000 * (j + 1)
        for e in range(E):
            if m >> e & 1:
                chr_.append(0); start.append(base + 100); end.append(base + 900); typ.append(e)
    order = np.lexsort((np.arange(len(start)), start))
    ix = dict(chr=np.array(chr_, dtype=np.int32)[order], start=np.array(start, dtype=np.int64)[order], end=np.array(end, dtype=np.int64)[order],
              type=np.array(typ, dtype=np.int32)[order], strand=np.ones(len(start), dtype=np.uint8), n_chromosomes=1)
    # filler reads first: 3600 single-hit reads, one region after the other (every filler set is known after the first 2000)
    tid, st, en, nh, qid = [], [], [], [], []
    for q in range(3600):
        base = 10000 * (len(sets) + q % len(fillers) + 1)
        tid.append(0); st.append(base + 300); en.append(base + 349); nh.append(1); qid.append(q)
    n_fill = len(st)
    # then the reads of interest: name q has NH = 2; its two hits sit in regions a(q), b(q) (often the same region, sometimes one in empty space)
    for q in range(600):
        a = int(rng.integers(0, len(sets)))
        b = a if rng.random() < 0.5 else int(rng.integers(0, len(sets) + 1))
        for r in (a, b):
            base = 10000 * (r + 1) if r < len(sets) else 5000        # 5000: no interval there
            s = base + 200 + int(rng.integers(0, 50))
            tid.append(0); st.append(s); en.append(s + 49); nh.append(2); qid.append(3600 + q)
    perm = np.concatenate([np.arange(n_fill), n_fill + rng.permutation(len(st) - n_fill)])
    reads = dict(tid=np.array(tid, dtype=np.int32)[perm], start=np.array(st, dtype=np.int64)[perm], end=np.array(en, dtype=np.int64)[perm],
                 strand=np.ones(len(st), dtype=np.uint8), nh=np.array(nh, dtype=np.uint32)[perm], qid=np.array(qid, dtype=np.int64)[perm])
    cfg = to_cfg(mm, desc)
    for order_kind in ("shuffled", "grouped"):
        if order_kind == "grouped":
            o = np.argsort(reads["qid"], kind="stable")
            reads = {k: v[o] for k, v in reads.items()}
        ref = oracle.run(desc, ix, reads, overlap=-1.0, strandedness="U", strategy="default", want_hit_mask=True)
        assert target in set(ref.hit_mask.tolist()) and target in set(ref.row_mask.tolist())
        for batches in (2, 3, 1):       # (2 and 3: the first batch holds filler reads only)
            r = run_gpu(mm, cfg, ix, reads, overlap=-1.0, strandedness="U", strategy="default", batches=batches)
            assert (r["hit"] == ref.hit_mask).all()
            assert r["stats"] == ref.stats, (order_kind, batches)
            assert r["table"] == oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset"), (order_kind, batches)
    ref = oracle.run(desc, ix, reads, overlap=-1.0, strandedness="U", strategy="ratio", want_hit_mask=True)
    for batches, keep in ((1, True), (3, False)):
        r = run_gpu(mm, cfg, ix, reads, overlap=-1.0, strandedness="U", strategy="ratio", batches=batches, keep=keep)
        assert r["stats"] == ref.stats
        assert dict(zip(r["masks"].tolist(), r["sums"].view(np.uint64).tolist())) == dict(zip(ref.row_mask.tolist(), ref.row_sum.view(np.uint64).tolist()))


@pytest.mark.parametrize("stride,shuffle", [(1, True), (700, True), (700, False), (70000, True), (400000, True)])
def test_every_grouping_layout(gpu_lib, oracle, stride, shuffle):
    """The multi-hit merge picks its path per batch on the device: a key range of at most 21 bits -> one partition pass by bucket of
    4096 names + k_leaf (LDS reduction per bucket, file-order replay of the names whose record order matters); wider ranges -> two
    or three partition passes + k_bucket_bounds + k_leaf; name-grouped input with a wide range -> compaction + k_chunk_default.
    Same reads, read-name ids spread out by
    `stride` to force each layout (one, two and - stride 400000, 31 key bits, 50 GB of per-name state - three partition passes;
    name-grouped with a wide range - the compaction), malformed groups included: always the oracle's answer, whatever the batch split."""
    mm = gpu_lib
    rng = np.random.default_rng(2024)
    desc = sr.random_config(rng, n_lines=6, max_per_line=3)
    ix = sr.random_index(rng, desc.E, nchr=3, n=2500, maxpos=120000)
    reads = sr.random_reads(rng, ix, n=30000, nchr=3, maxpos=120000, nh_max=9, shuffle=shuffle)
    reads = dict(reads, qid=reads["qid"] * stride)
    ref = oracle.run(desc, ix, reads, overlap=0.3, strandedness="R", strategy="default", want_hit_mask=True)
    cfg = to_cfg(mm, desc)
    for batches in (1, 4):
        r = run_gpu(mm, cfg, ix, reads, overlap=0.3, strandedness="R", strategy="default", batches=batches)
        assert (r["hit"] == ref.hit_mask).all()
        assert r["stats"] == ref.stats, (stride, shuffle, batches)
        assert r["table"] == oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset"), (stride, shuffle, batches)


@pytest.mark.parametrize("order", ["interleaved", "grouped", "far"])
def test_leaf_merge_at_scale(gpu_lib, oracle, order):
    """The multi-hit merge on ~1.1 M records of 200 k read names, a quarter of them malformed (over-complete, under-complete, mixed NH)
    - dozens of 4096-name buckets per batch, every one holding order-dependent names, tens of replay chunks per bucket - in three
    arrangements: hits of a read a few thousand records apart (one partition pass), adjacent (the runs of equal names that k_leaf
    reduces inside a wave), and read-name ids spread out so that two partition passes + k_bucket_bounds are needed.  Table and
    counters equal the oracle's, in one batch and in five."""
    mm = gpu_lib
    rng = np.random.default_rng(77)
    desc = sr.random_config(rng, n_lines=5, max_per_line=3)
    ix = sr.random_index(rng, desc.E, nchr=3, n=3000, maxpos=150000)
    names = 200_000
    k = rng.integers(2, 9, names)                                    # NH of the name
    recs = k.copy()
    r = rng.random(names)
    over = r < 0.10; recs[over] += rng.integers(1, 2 * k[over] + 1)  # over-complete (e.g. both mates under one name)
    under = (r >= 0.10) & (r < 0.18); recs[under] = np.maximum(1, k[under] - rng.integers(1, k[under]))
    qid = np.repeat(np.arange(names, dtype=np.int64), recs)
    nh = np.repeat(k, recs).astype(np.uint32)
    mixed = np.repeat((r >= 0.18) & (r < 0.26), recs)
    nh[mixed] = rng.integers(1, 9, int(mixed.sum()))                 # mixed NH inside one name (NH 1 records bypass the group)
    n = len(qid)
    pick = rng.integers(0, len(ix["start"]), n)
    start = np.maximum(1, ix["start"][pick] + rng.integers(-80, 80, n))
    reads = dict(tid=ix["chr"][pick].astype(np.int32), start=start.astype(np.int64), end=(start + rng.integers(20, 120, n)).astype(np.int64),
                 strand=rng.integers(0, 2, n).astype(np.uint8), nh=nh, qid=qid)
    if order != "grouped":
        perm = np.argsort(np.arange(n) + rng.random(n) * 5000.0, kind="stable")      # hits of a read up to ~5000 records apart
        reads = {key: v[perm] for key, v in reads.items()}
    if order == "far":
        reads = dict(reads, qid=reads["qid"] * 300)                                   # 26 key bits: two partition passes
    ref = oracle.run(desc, ix, reads, overlap=-1.0, strandedness="F", strategy="default")
    cfg = to_cfg(mm, desc)
    for batches in (1, 5):
        got = run_gpu(mm, cfg, ix, reads, overlap=-1.0, strandedness="F", strategy="default", batches=batches, keep=False)
        assert got["stats"] == ref.stats, (order, batches)
        assert got["table"] == oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset"), (order, batches)


@pytest.mark.parametrize("strategy", ["default", "ratio", "unique"])
def test_twelve_byte_link_format(gpu_lib, oracle, strategy):
    """mm_annotate_batch12: 12 bytes per tuple cross the link and k_expand12 rebuilds the 16-byte tuples on the device - same per-hit
    sets, counters and table as the oracle (zero-length reads, unknown chromosomes, NH 0..8 and a few NH far beyond the 8-bit
    field, malformed groups), from host memory and from tuples already resident in HBM, in one batch and in four."""
    import torch
    mm = gpu_lib
    rng = np.random.default_rng(1212)
    desc = sr.random_config(rng, n_lines=6, max_per_line=3)
    ix = sr.random_index(rng, desc.E, nchr=3, n=2500, maxpos=120000)
    reads = sr.random_reads(rng, ix, n=30000, nchr=3, maxpos=120000, nh_max=8)
    long_ones = (reads["end"] - reads["start"] + 1) > 16383                      # (the generator makes a few 40 kb reads: beyond the 14-bit length)
    reads = {k: v[~long_ones] for k, v in reads.items()}
    nh = reads["nh"].copy()
    for pos, val in ((100, 255), (5000, 300), (9000, 70000), (20000, 254)):
        nh[reads["qid"] == int(reads["qid"][pos])] = val
    reads = dict(reads, nh=nh, qid=reads["qid"] + 1_000_000)
    ref = oracle.run(desc, ix, reads, overlap=-1.0, strandedness="F", strategy=strategy, want_hit_mask=True)
    cfg = to_cfg(mm, desc)
    index = mm.IntervalIndex(cfg, ix["chr"], ix["start"], ix["end"], ix["type"], ix["strand"], ix["n_chromosomes"])
    aln = mm.pack_alignments(reads["tid"], reads["start"], reads["end"], reads["strand"], reads["qid"], reads["nh"])
    n = len(aln)
    qb = int(reads["qid"].max()) + 1
    for resident in (False, True):
        for batches in (1, 4):
            ctr = mm.Counter(index, overlap=-1.0, strandedness="F", strategy=strategy, keep_hit_masks=True)
            hm = []
            cuts = np.linspace(0, n, batches + 1).astype(int)
            for a, b in zip(cuts[:-1], cuts[1:]):
                pk, big12 = mm.pack12(aln[a:b], mm.nh_side_list(reads["nh"][a:b]), qid_base=1_000_000)
                if resident:
                    dev = torch.from_numpy(pk.view(np.uint8).reshape(-1).copy()).cuda()
                    ctr.add_batch12(pk, big12, 1_000_000, qb, on_device_ptr=dev.data_ptr())
                else:
                    ctr.add_batch12(pk, big12, 1_000_000, qb, qid_lo=1_000_000)
                ctr.sync()
                hm.append(ctr.hit_masks(b - a))
            ctr.flush()
            masks, sums = ctr.getCounts()
            table = mm.TableCount(cfg, 1); table.add_rows(masks, sums)
            assert (np.concatenate(hm) == ref.hit_mask).all(), (resident, batches)
            assert ctr.stats() == ref.stats, (resident, batches)
            assert table.dump(["t"]) == oracle.format_table(desc, ref.row_mask, ref.row_count, "t"), (resident, batches)
            ctr.close()
    index.close()


@pytest.mark.parametrize("shift", [7, 10])
@pytest.mark.parametrize("l", [-1.0, 0.5, 25.0])
def test_reads_across_bin_boundaries(gpu_lib, oracle, shift, l):
    """The slot of a bin also lists the intervals that start in the first min(W, 256) bases of the next bin, so that a read ending
    there needs one slot: reads ending one base before, at and after that reach, intervals starting at each of those offsets, a bin
    with more records than a slot line holds inline, and an interval open across all of them - against the oracle, per hit."""
    mm = gpu_lib
    rng = np.random.default_rng(77 + shift)
    desc = sr.random_config(rng, n_lines=4, max_per_line=3)
    W = 1 << shift
    reach = min(W, 256)
    chr_, start, end, typ, strand = [], [], [], [], []
    def iv(c, s, length, t):
        chr_.append(c); start.append(s); end.append(s + length - 1); typ.append(t % desc.E); strand.append(1 + (t & 1))
    t = 0
    for c in (0, 1):
        iv(c, 2 * W + 17, 9 * W, t); t += 1                       # open at the left edge of every bin below
        for k in range(3, 9):
            B = k * W
            for off in (-1, 0, 1, reach - 2, reach - 1, reach, reach + 1, W - 1):
                iv(c, B + off, 30 + 7 * (t % 5), t); t += 1
            if k == 5:
                for j in range(9):                                # more than the seven inline records, inside the reach of bin 4
                    iv(c, B + 5 + j // 3, 40, t); t += 1
    order = np.lexsort((np.arange(len(start)), np.array(start), np.array(chr_)))
    ix = dict(chr=np.array(chr_, dtype=np.int32)[order], start=np.array(start, dtype=np.int64)[order], end=np.array(end, dtype=np.int64)[order],
              type=np.array(typ, dtype=np.int32)[order], strand=np.array(strand, dtype=np.uint8)[order], n_chromosomes=2)
    tid, rs, re_, rstr, nh, qid = [], [], [], [], [], []
    q = 0
    for c in (0, 1):
        for k in range(3, 9):
            B = k * W
            for back in range(1, min(W, 120), 7):
                for L in (20, 60, 100, reach - 1, reach, reach + 1, 2 * reach, W + reach, 3 * W):
                    s = B - back
                    for rep in range(2 if (back + L) % 5 == 0 else 1):           # a few two-hit names
                        tid.append(c); rs.append(s); re_.append(s + L - 1 + rep); rstr.append((back + L + rep) & 1)
                        nh.append(2 if (back + L) % 5 == 0 else 1); qid.append(q)
                    q += 1
    reads = dict(tid=np.array(tid, dtype=np.int32), start=np.array(rs, dtype=np.int64), end=np.array(re_, dtype=np.int64),
                 strand=np.array(rstr, dtype=np.uint8), nh=np.array(nh, dtype=np.uint32), qid=np.array(qid, dtype=np.uint32))
    cfg = to_cfg(mm, desc)
    for s_ in ("U", "F"):
        for y in ("default", "ratio"):
            ref = oracle.run(desc, ix, reads, overlap=l, strandedness=s_, strategy=y, want_hit_mask=True)
            r = run_gpu(mm, cfg, ix, reads, overlap=l, strandedness=s_, strategy=y, batches=2, bin_shift=shift)
            assert (r["hit"] == ref.hit_mask).all(), f"hit masks differ ({s_}, {y})"
            assert r["stats"] == ref.stats
            assert r["table"] == oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset")
            r = run_gpu(mm, cfg, ix, reads, overlap=l, strandedness=s_, strategy=y, batches=1, bin_shift=shift, keep=False)
            assert r["stats"] == ref.stats and r["table"] == oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset")
