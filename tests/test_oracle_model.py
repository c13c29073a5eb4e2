"""CPU: a third, independent derivation — a brute-force pure-Python model of SURVEY.md section 8a's
closed form (all intervals of the chromosome with start <= read.end, in index order, last writer
wins; plus the 'leading isBefore run' exception for reads without CIGAR operations) — against the
C oracle (which follows the reference's binned walk).  Small inputs only."""
from collections import defaultdict

import numpy as np
import pytest

import synth_ref as sr


def f32(x):
    return np.float32(x)


def model_probe(desc, ix, tid, s, e, rstrand, mode, lval):
    E = desc.E
    line_of, es, up, down = desc.arrays()
    if tid < 0 or tid >= ix["n_chromosomes"]:
        return 0
    sel = np.nonzero(ix["chr"] == tid)[0]
    ev = {}
    pmax = -1
    for i in sel:
        st, en, t, sd = int(ix["start"][i]), int(ix["end"][i]), int(ix["type"][i]), int(ix["strand"][i])
        before_all = pmax < s        # every earlier interval of the chromosome ends before the read
        pmax = max(pmax, en)
        if st > e:
            break
        if en < s and before_all:
            continue                  # swallowed by the first loop of scan(), mmannot.cpp:1306-1308
        if es[t] == 1 and not ((sd == 1) == bool(rstrand)):
            continue
        if es[t] == 2 and ((sd == 1) == bool(rstrand)):
            continue
        if mode == 0:
            o = 1 if (st <= s and e <= en) else 0
        else:
            lo, hi = max(s, st), min(e, en)
            ov = hi - lo if lo < hi else 0
            if mode == 1:
                o = ov if f32(f32(e - s + 1) * f32(lval)) <= f32(ov) else 0
            else:
                o = ov if f32(ov) >= f32(lval) else 0
        if o == 0:
            continue
        d = 0
        if up[t]:
            p = en
            d = s - p if p < s else (p - e if p > e else 0)
        elif down[t]:
            p = st
            d = s - p if p < s else (p - e if p > e else 0)
        ev[t] = (o, d)
    if not ev:
        return 0
    first = min(ev)
    L = [t for t in ev if line_of[t] == line_of[first]]
    mo_ = max(ev[t][0] for t in L)
    S = [t for t in L if ev[t][0] == mo_]
    if len(S) > 1:
        md = min(ev[t][1] for t in S)
        S = [t for t in S if ev[t][1] == md]
    m = 0
    for t in S:
        m |= 1 << t
    return m


def model_count(masks, nh, qid, strategy):
    pending, count = {}, defaultdict(float)
    st = dict(nHits=0, nReads=0, nUnique=0, nAmbiguous=0, nMultiple=0, nUnassigned=0, nRescued=0)
    for i, m in enumerate(masks):
        h, q = int(nh[i]), int(qid[i])
        if strategy == "unique" and h != 1:
            continue
        st["nHits"] += 1
        pc = bin(m).count("1")
        if pc == 0: st["nUnassigned"] += 1
        elif pc > 1: st["nAmbiguous"] += 1
        elif h == 1: st["nUnique"] += 1
        if h > 1 and strategy == "default":
            st["nMultiple"] += 1
            if q not in pending:
                pending[q] = [h - 1, m]; st["nReads"] += 1
            else:
                g = pending[q]; g[0] -= 1; g[1] |= m
                if g[0] == 0:
                    if g[1]:
                        count[g[1]] += 1.0
                        if bin(g[1]).count("1") == 1: st["nRescued"] += 1
                    del pending[q]
        else:
            if m:
                count[m] += (1.0 / h if h else float("inf")) if strategy == "ratio" else 1.0
            st["nReads"] += 1
    for g in pending.values():
        if g[1]:
            count[g[1]] += 1.0
            if bin(g[1]).count("1") == 1: st["nRescued"] += 1
    return count, st


@pytest.mark.parametrize("seed,l,strategy", [(1, -1.0, "default"), (2, 0.0, "default"), (3, 0.6, "ratio"), (4, 30.0, "unique"),
                                             (5, -1.0, "ratio"), (6, 12.0, "default")])
def test_model_vs_oracle(oracle, seed, l, strategy):
    rng = np.random.default_rng(seed)
    desc = sr.random_config(rng, n_lines=5, max_per_line=3)
    ix = sr.random_index(rng, desc.E, nchr=2, n=300, maxpos=20000, long_frac=0.1)
    reads = sr.random_reads(rng, ix, n=2500, nchr=2, maxpos=20000, nh_max=5, zero_len_frac=0.15)
    mode = oracle.overlap_mode_of(l)
    for s in ("U", "F", "R"):
        ref = oracle.run(desc, ix, reads, overlap=l, strandedness=s, strategy=strategy, want_hit_mask=True)
        masks = []
        for i in range(len(reads["start"])):
            rs = 1 if s == "U" else (int(reads["strand"][i]) if s == "F" else 1 - int(reads["strand"][i]))
            masks.append(model_probe(desc, ix, int(reads["tid"][i]), int(reads["start"][i]), int(reads["end"][i]), rs, mode, l))
        proc = np.array([not (strategy == "unique" and int(h) != 1) for h in reads["nh"]])
        got = np.array(masks, dtype=np.uint64)
        assert (got[proc] == ref.hit_mask[proc]).all()
        count, st = model_count(masks, reads["nh"], reads["qid"], strategy)
        assert st == ref.stats
        want = dict(zip(ref.row_mask.tolist(), ref.row_sum.tolist()))
        assert {k: v for k, v in count.items()} == want


def test_oracle_large_table(oracle):
    """> 1024 distinct element sets: the oracle's table must survive its own growth."""
    rng = np.random.default_rng(1250)
    desc = sr.random_config(rng, n_lines=int(rng.integers(3, 12)), max_per_line=int(rng.integers(1, 5)))
    ix = sr.random_index(rng, desc.E, nchr=4, n=3000, maxpos=150000)
    reads = sr.random_reads(rng, ix, n=20000, nchr=4, maxpos=150000, nh_max=7)
    ref = oracle.run(desc, ix, reads, overlap=25.0, strandedness="U", strategy="default", want_hit_mask=True)
    assert len(ref.row_mask) > 1024
    count, st = model_count([int(m) for m in ref.hit_mask], reads["nh"], reads["qid"], "default")
    assert st == ref.stats
    assert dict(count) == dict(zip(ref.row_mask.tolist(), ref.row_sum.tolist()))
