"""GPU: BASELINE.json's configurations at their full sizes (config 2: 50 M reads NH=1 vs TAIR10-scale; config 3: 500 M reads
NH<=20 vs HS38-scale; config 4: 200 M read pairs vs a Flybase6-shaped annotation with introns and vicinity, -l 50, run under
-s U because the reference's parser rejects -s FR and so no oracle for it exists; config 5: 1 B reads NH<=100,
-y ratio), inputs generated on the device batch by batch.  The oracle cannot
run at these sizes in seconds, so the checks are size independent:
  * batch-split invariance: the whole stream cut into batches of two different sizes gives the identical table and counters
    (bit-identical double sums under -y ratio: the file-order accumulation may not depend on batching);
  * counter identities that hold for any input;
  * parity with the CPU oracle on a prefix of the very same stream (numpy twin of the device generator)."""
import ctypes as C

import numpy as np
import pytest

import synth_ref as sr

pytestmark = pytest.mark.gpu


def run_stream(mm, index, cfg, *, total, batch, seed, read_len, nh_max, scatter, chr_len, strategy, pair=0, overlap=-1.0, strand="F"):
    import torch
    from mmannot_amd import _lib
    dev = torch.device("cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctr = mm.Counter(index, overlap=overlap, strandedness=strand, strategy=strategy, stream=stream)
    a = torch.empty((batch, 4), dtype=torch.int32, device=dev)
    gpt = (nh_max / (nh_max * (nh_max + 1) / 2.0)) if nh_max > 1 else 1.0
    qb = int(total * gpt) + nh_max + 2
    done = 0
    while done < total:
        n = min(batch, total - done)
        p = _lib.SynthParams(seed, done, read_len, nh_max, len(chr_len), chr_len.ctypes.data, scatter, pair)
        _lib.check(_lib.lib().mm_synth_fill(a.data_ptr(), n, C.byref(p), stream))
        ctr.add_batch_device(a.data_ptr(), n, qb, single_hit=(nh_max == 1))
        ctr.sync()            # the tuple buffer is reused for the next batch
        done += n
    ctr.flush()
    masks, sums = ctr.getCounts()
    st = ctr.stats()
    ctr.close()
    o = np.argsort(masks)
    return masks[o], sums[o], st


CONFIGS = [
    ("c2", dict(shape="tair10", scale=1.0, total=50_000_000, read_len=50, nh_max=1, scatter=0, strategy="default")),
    ("c3", dict(shape="hs38", scale=1.5, total=500_000_000, read_len=100, nh_max=20, scatter=16, strategy="default")),
    ("c4", dict(shape="flybase6", scale=1.0, total=400_000_000, read_len=100, nh_max=1, scatter=0, strategy="default", pair=250, overlap=50.0,
                strand="U")),
    ("c5", dict(shape="hs38", scale=1.5, total=1_000_000_000, read_len=100, nh_max=100, scatter=16, strategy="ratio")),
]


@pytest.mark.parametrize("name,c", CONFIGS)
def test_full_size(gpu_lib, oracle, name, c):
    from mmannot_amd import synth
    mm = gpu_lib
    an = synth.synth_annotation(c["shape"], seed=3, scale=c["scale"])
    cfg = mm.Config.from_json(an.config)
    index = mm.IntervalIndex(cfg, an.chr, an.start, an.end, an.type, an.strand, an.n_chromosomes)
    chr_len = np.ascontiguousarray(an.chr_len, dtype=np.int64)
    ovl, sd, pair = c.get("overlap", -1.0), c.get("strand", "F"), c.get("pair", 0)
    kw = dict(total=c["total"], seed=1234, read_len=c["read_len"], nh_max=c["nh_max"], scatter=c["scatter"], chr_len=chr_len, strategy=c["strategy"],
              pair=pair, overlap=ovl, strand=sd)
    # windows of 2^scatter tuples must not be cut differently by the two runs' flush points only at the very end: both runs
    # process the same total stream, so every group is complete in both
    m1, s1, st1 = run_stream(mm, index, cfg, batch=1 << 24, **kw)
    m2, s2, st2 = run_stream(mm, index, cfg, batch=(1 << 22) * 5, **kw)
    assert st1 == st2
    assert np.array_equal(m1, m2)
    assert np.array_equal(s1.view(np.uint64), s2.view(np.uint64))          # bit-identical, also the ratio sums
    # identities
    assert st1["nHits"] == c["total"]
    if c["strategy"] == "default":
        assert int(np.rint(s1).sum()) <= st1["nReads"] and np.array_equal(s1, np.rint(s1))
        assert st1["nUnique"] + st1["nRescued"] <= st1["nReads"]
        if c["nh_max"] == 1:
            assert st1["nReads"] == c["total"] and st1["nMultiple"] == 0 and st1["nRescued"] == 0
            assert int(s1.sum()) == c["total"] - st1["nUnassigned"]
            single = sum(int(v) for m, v in zip(m1, s1) if bin(int(m)).count("1") == 1)
            assert single == st1["nUnique"]
    else:
        assert st1["nReads"] == c["total"] and st1["nRescued"] == 0
        assert s1.sum() < c["total"] - st1["nUnassigned"] + 1
    # prefix parity with the oracle (same generator, numpy twin); the prefix ends on a window boundary so every group in it is whole
    n = 1 << 21
    t = sr.synth_tuples(1234, 0, n, c["read_len"], c["nh_max"], chr_len, c["scatter"], pair)
    desc = oracle.ConfigDesc.from_json(an.config)
    ref = oracle.run(desc, an.as_index_dict(), t, overlap=ovl, strandedness=sd, strategy=c["strategy"])
    mp, sp, stp = run_stream(mm, index, cfg, **dict(kw, total=n), batch=1 << 20)
    fo = np.argsort(ref.row_mask)
    assert stp == ref.stats
    assert np.array_equal(mp, ref.row_mask[fo]) and np.array_equal(sp.view(np.uint64), ref.row_sum[fo].view(np.uint64))
    index.close()


def test_one_very_large_batch(gpu_lib):
    """A single batch of 40 M alignments (more than 4096 sort tiles, so k_rs_scan loops) gives what 2^22-alignment batches give."""
    from mmannot_amd import synth
    mm = gpu_lib
    an = synth.synth_annotation("hs38", seed=3, scale=0.5)
    cfg = mm.Config.from_json(an.config)
    index = mm.IntervalIndex(cfg, an.chr, an.start, an.end, an.type, an.strand, an.n_chromosomes)
    chr_len = np.ascontiguousarray(an.chr_len, dtype=np.int64)
    kw = dict(total=40_000_000, seed=77, read_len=100, nh_max=20, scatter=16, chr_len=chr_len, strategy="default")
    m1, s1, st1 = run_stream(mm, index, cfg, batch=40_000_000, **kw)
    m2, s2, st2 = run_stream(mm, index, cfg, batch=1 << 22, **kw)
    assert st1 == st2 and np.array_equal(m1, m2) and np.array_equal(s1, s2)
    kw = dict(kw, strategy="ratio", nh_max=100)
    m1, s1, st1 = run_stream(mm, index, cfg, batch=40_000_000, **kw)
    m2, s2, st2 = run_stream(mm, index, cfg, batch=1 << 22, **kw)
    assert st1 == st2 and np.array_equal(m1, m2) and np.array_equal(s1.view(np.uint64), s2.view(np.uint64))
    index.close()
