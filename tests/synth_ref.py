"""Seeded synthetic inputs for the parity tests (numpy), including the numpy twin of the device
generator k_synth (mmannot_amd/csrc/mm_device.hip).  Test infrastructure only."""
from __future__ import annotations

import numpy as np

from oracle import mm_oracle as mo

M64 = (1 << 64) - 1


def random_config(rng, n_lines=8, max_per_line=3, with_vicinity=True):
    els = []
    for ln in range(n_lines):
        k = int(rng.integers(1, max_per_line + 1))
        for j in range(k):
            strand = int(rng.choice([0, 0, 1, 2]))
            typ = f"t{ln}_{j}"
            els.append(mo.Element(f"src{ln}", typ, strand, ln))
    if with_vicinity:
        ln = n_lines
        els.append(mo.Element("srcV", "upstream", 0, ln))
        els.append(mo.Element("srcV", "downstream", 0, ln))
        els.append(mo.Element("srcV", "gene", int(rng.choice([0, 2])), ln))
    return mo.ConfigDesc(els[:64])


def random_index(rng, E, nchr=3, n=2000, maxpos=200000, long_frac=0.05, skip_chr=None):
    chr_ = rng.integers(0, nchr, n)
    if skip_chr is not None:
        chr_[chr_ == skip_chr] = (skip_chr + 1) % nchr
    start = rng.integers(1, maxpos, n)
    length = np.where(rng.random(n) < long_frac, rng.integers(1000, maxpos // 2, n), rng.integers(1, 600, n))
    # many duplicate starts / abutting ends to exercise tie order and boundary arithmetic
    dup = rng.random(n) < 0.2
    start[dup] = (start[dup] // 500) * 500 + 1
    end = start + length - 1
    typ = rng.integers(0, E, n)
    strand = rng.integers(1, 3, n)
    order = np.lexsort((np.arange(n), start, chr_))  # stable: (chr, start), ties in generation order
    return dict(chr=chr_[order].astype(np.int32), start=start[order].astype(np.int64), end=end[order].astype(np.int64),
                type=typ[order].astype(np.int32), strand=strand[order].astype(np.uint8), n_chromosomes=nchr)


def random_reads(rng, index, n=5000, nchr=3, maxpos=200000, nh_max=6, zero_len_frac=0.03, unknown_frac=0.02,
                 malformed_groups=True, shuffle=True, nh_min=1):
    """Reads grouped by name; a share of the groups is over-/under-complete or mixes NH values.  shuffle=False keeps the
    hits of a read adjacent (an aligner's own output order)."""
    tid, start, end, strand, nh, qid = [], [], [], [], [], []
    q = 0
    ends = index["end"]
    while len(start) < n:
        k = int(rng.integers(nh_min, nh_max + 1))
        recs = k
        nhs = [k] * k
        if malformed_groups and k > 1:
            r = rng.random()
            if r < 0.10:
                recs = k + int(rng.integers(1, 2 * k + 1)); nhs = [k] * recs          # over-complete (e.g. both mates share a name)
            elif r < 0.18:
                recs = max(1, k - int(rng.integers(1, k))); nhs = [k] * recs          # under-complete
            elif r < 0.26:
                nhs = [int(rng.integers(1, nh_max + 1)) for _ in range(recs)]          # mixed NH inside one name
            elif r < 0.30:
                nhs = [0] * recs                                                       # NH stored in a signed tag type -> nHits 0
        for j in range(recs):
            t = int(rng.integers(0, nchr))
            if rng.random() < unknown_frac:
                t = -1 if rng.random() < 0.5 else nchr + 3
            if rng.random() < 0.5 and len(ends):
                # land near an interval boundary
                iv = int(rng.integers(0, len(ends)))
                base = int(index["start"][iv] if rng.random() < 0.5 else index["end"][iv])
                s = max(1, base + int(rng.integers(-60, 60)))
                t = int(index["chr"][iv]) if t >= 0 and t < nchr else t
            else:
                s = int(rng.integers(1, maxpos + 5000))
            ln = int(rng.integers(20, 120))
            if rng.random() < 0.02:
                ln = int(rng.integers(1000, 40000))
            e = s + ln - 1
            if rng.random() < zero_len_frac:
                e = s - 1                                                              # record without CIGAR operations
            tid.append(t); start.append(s); end.append(e); strand.append(int(rng.integers(0, 2))); nh.append(nhs[j]); qid.append(q)
        q += 1
    a = dict(tid=np.array(tid, dtype=np.int32), start=np.array(start, dtype=np.int64), end=np.array(end, dtype=np.int64),
             strand=np.array(strand, dtype=np.uint8), nh=np.array(nh, dtype=np.uint32), qid=np.array(qid, dtype=np.int64))
    # interleave: the hits of one read are generally not adjacent in a coordinate-sorted file
    perm = np.argsort(rng.random(len(a["start"])) + np.arange(len(a["start"])) / 40.0, kind="stable") if shuffle else np.arange(len(a["start"]))
    return {k: v[perm][:n] for k, v in a.items()}


# --------------------------------------------------------------------------- numpy twin of k_synth
def splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_tuples(seed, first, n, read_len, nh_max, chr_len, scatter_log2, pair_insert=0):
    """Exactly what mm_synth_fill writes (mm_device.hip:k_synth)."""
    with np.errstate(over="ignore"):
        i = np.uint64(first) + np.arange(n, dtype=np.uint64)
        L = i.copy()
        if scatter_log2:
            W = np.uint64(1) << np.uint64(scatter_log2)
            r = i & (W - np.uint64(1))
            L = (i & ~(W - np.uint64(1))) | ((r * np.uint64(0x9E3779B1) + (np.uint64(seed) & (W - np.uint64(1)))) & (W - np.uint64(1)))
        nh = np.ones(n, dtype=np.uint32)
        qid = L.copy()
        if nh_max > 1:
            K = np.uint64(nh_max); T = K * (K + np.uint64(1)) // np.uint64(2)
            blk = L // T; off = L % T
            k = ((np.sqrt(8.0 * off.astype(np.float64) + 1.0) - 1.0) * 0.5).astype(np.uint64)
            k = np.where(k * (k + np.uint64(1)) // np.uint64(2) > off, k - np.uint64(1), k)
            k = np.where((k + np.uint64(1)) * (k + np.uint64(2)) // np.uint64(2) <= off, k + np.uint64(1), k)
            nh = (k + np.uint64(1)).astype(np.uint32)
            qid = blk * K + k
        mate = np.zeros(n, dtype=np.uint64)
        foot = np.uint64(read_len)
        if pair_insert:
            assert nh_max == 1
            mate = L & np.uint64(1); L = L >> np.uint64(1); qid = L.copy(); foot = np.uint64(read_len + pair_insert + 63)
        h = splitmix64(np.uint64(seed) ^ (L * np.uint64(0xD1342543DE82EF95)))
        h2 = splitmix64(h)
        cum = np.concatenate([[0], np.cumsum(np.asarray(chr_len, dtype=np.uint64))]).astype(np.uint64)
        g = h % cum[-1]
        c = np.searchsorted(cum, g, side="right") - 1
        ln = cum[c + 1] - cum[c]
        span = np.where(ln > foot, ln - foot + np.uint64(1), np.uint64(1))
        pos0 = h2 % span + mate * (np.uint64(pair_insert) + ((h >> np.uint64(32)) & np.uint64(63)))
        start = (pos0 + np.uint64(1)).astype(np.int64)
        end = start + read_len - 1
        strand = (((h >> np.uint64(63)) ^ mate) & np.uint64(1)).astype(np.uint8)
    return dict(tid=c.astype(np.int32), start=start, end=end, strand=strand, nh=nh, qid=(qid & np.uint64(0xFFFFFFFF)).astype(np.int64))
