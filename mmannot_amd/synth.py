"""Seeded synthetic annotations of the shapes BASELINE.json's configs name (bench.py, tests).

    shape "hs38"   : configHS38-shaped Order (22 elements / 20 lines), 25 chromosomes with
                     human-like lengths, ~60 k genes -> ~1.3 M intervals        (configs[2], [4])
    shape "tair10" : configTAIR10-shaped Order (14 elements / 13 lines), 5 chromosomes,
                     ~33 k genes -> ~0.45 M intervals                           (configs[1])
    shape "flybase6": configFlybase6-shaped Order (16 elements / 15 lines: Introns + Vicinity of
                     FlyBase:gene, a second source for piRNA), 8 chromosome arms,
                     ~17 k genes -> ~0.2 M intervals                            (configs[3])

A gene model is (chromosome, strand, merged exons, CDS hull).  From it the generator emits
  * the intervals the reference derives from such a gene (mmannot.cpp:1227-1266: CDS pieces, 5'UTR,
    3'UTR, introns, upstream, downstream, then the exons under the gene's own type), and
  * on request the Ensembl-style GTF text (gene / transcript / exon / CDS lines) describing the very
    same genes, for the CLI path and for timing the compiled reference on identical input.
Everything is a pure function of (shape, seed, scale).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

STRAND_ALL, STRAND_F, STRAND_R = 0, 1, 2

# (source, type, strand) per Order entry; inner lists are Order lines
_SMALL_HS = ["miRNA", "snoRNA", "snRNA", "lincRNA", "piRNA", "rRNA", "misc_RNA", "transcribed_processed_pseudogene",
             "processed_pseudogene", "polymorphic_pseudogene", "unprocessed_pseudogene", "pseudogene"]
_SMALL_TAIR = ["miRNA", "tRNA", "snoRNA", "snRNA", "rRNA", "transposable_element_gene", "pseudogenic_transcript"]
_SMALL_FLY = ["miRNA", "tRNA", "snoRNA", "snRNA", "rRNA", "ncRNA", "pre_miRNA"]


def order_lines(shape: str) -> List[List[Tuple[str, str, int]]]:
    if shape == "hs38":
        pc, pt = "protein_coding", "processed_transcript"
        lines = [[(pc, "CDS", STRAND_F)], [(pc, "5'UTR", STRAND_F)], [(pc, "3'UTR", STRAND_F)]]
        lines += [[(s, "gene", STRAND_ALL)] for s in _SMALL_HS]
        lines += [[(pc, "intron", STRAND_ALL)], [(pt, "intron", STRAND_ALL)], [(pc, "gene", STRAND_R)],
                  [(pc, "upstream", STRAND_ALL), (pc, "downstream", STRAND_ALL)],
                  [(pt, "upstream", STRAND_ALL), (pt, "downstream", STRAND_ALL)]]
        return lines
    if shape == "tair10":
        t = "TAIR10"
        lines = [[(t, "CDS", STRAND_F)], [(t, "5'UTR", STRAND_F)], [(t, "3'UTR", STRAND_F)]]
        lines += [[(t, s, STRAND_ALL)] for s in _SMALL_TAIR]
        lines += [[(t, "gene", STRAND_R)], [(t, "intron", STRAND_ALL)], [(t, "upstream", STRAND_ALL), (t, "downstream", STRAND_ALL)]]
        return lines
    if shape == "flybase6":
        f = "FlyBase"
        lines = [[(f, "CDS", STRAND_F)], [(f, "5'UTR", STRAND_F)], [(f, "3'UTR", STRAND_F)]]
        lines += [[(f, s, STRAND_ALL)] for s in _SMALL_FLY]
        lines += [[("piRNA", "piRNA", STRAND_ALL)], [(f, "pseudogene", STRAND_ALL)], [(f, "gene", STRAND_R)], [(f, "intron", STRAND_ALL)],
                  [(f, "upstream", STRAND_ALL), (f, "downstream", STRAND_ALL)]]
        return lines
    raise ValueError(shape)


def config_json(shape: str):
    out = []
    for ln, line in enumerate(order_lines(shape)):
        for (src, typ, sd) in line:
            out.append(dict(source=src, type=typ, strand=sd, line=ln))
    return out


def config_text(shape: str) -> str:
    """The same Order as a configuration *file* (Synonyms/Introns/Vicinity sections as the shape needs)."""
    lines = order_lines(shape)
    main = {"hs38": [("protein_coding", "gene"), ("processed_transcript", "gene")], "tair10": [("TAIR10", "gene")],
            "flybase6": [("FlyBase", "gene")]}[shape]
    txt = ["Synonyms:", "mRNA:transcript", "five_prime_UTR:5'UTR", "three_prime_UTR:3'UTR", "Introns:"]
    txt += [f"{s}:{t}" for s, t in main] + ["Vicinity:"] + [f"{s}:{t}" for s, t in main] + ["Order:"]
    for line in lines:
        txt.append(",".join(f"{s}:{t}" + ({STRAND_F: " +", STRAND_R: " -"}.get(sd, "")) for s, t, sd in line))
    return "\n".join(txt) + "\n"


HS38_CHR = [("1", 248.9), ("2", 242.2), ("3", 198.3), ("4", 190.2), ("5", 181.5), ("6", 170.8), ("7", 159.3), ("8", 145.1),
            ("9", 138.4), ("10", 133.8), ("11", 135.1), ("12", 133.3), ("13", 114.4), ("14", 107.0), ("15", 102.0), ("16", 90.3),
            ("17", 83.3), ("18", 80.4), ("19", 58.6), ("20", 64.4), ("21", 46.7), ("22", 50.8), ("X", 156.0), ("Y", 57.2), ("MT", 0.0166)]
TAIR10_CHR = [("1", 30.43), ("2", 19.70), ("3", 23.46), ("4", 18.59), ("5", 26.98)]
FLY_CHR = [("2L", 23.51), ("2R", 25.29), ("3L", 28.11), ("3R", 32.08), ("4", 1.35), ("X", 23.54), ("Y", 3.67), ("mitochondrion_genome", 0.0195)]


@dataclass
class Annotation:
    shape: str
    config: list                      # config_json
    chr_names: List[str]
    chr_len: np.ndarray               # int64
    chr: np.ndarray                   # int32   sorted by (chr, start)
    start: np.ndarray                 # int64
    end: np.ndarray
    type: np.ndarray                  # int32 element id
    strand: np.ndarray                # uint8
    genes: Optional[list] = None      # gene models (for gtf_text)

    @property
    def n_chromosomes(self) -> int:
        return len(self.chr_names)

    def as_index_dict(self) -> Dict[str, np.ndarray]:
        return dict(chr=self.chr, start=self.start, end=self.end, type=self.type, strand=self.strand, n_chromosomes=self.n_chromosomes)


def _elem_id(cfg, source, typ):
    """Config::getOrder with literal sources (no wildcards in the synthetic configs), mmannot.cpp:414-424."""
    for i, e in enumerate(cfg):
        if e["source"] == source and (e["type"] == "" or e["type"] == typ):
            return i
    return -1


def synth_annotation(shape: str = "hs38", seed: int = 3, scale: float = 1.0, up: int = 1000, down: int = 1000,
                     keep_genes: bool = False, scaffolds: int = 0) -> Annotation:
    """scaffolds > 0: that many unplaced scaffolds of 60 kb behind the chromosomes (an assembly with its unlocalised sequences: hundreds to
    thousands of reference names), genes placed on them at the same density."""
    rng = np.random.default_rng(seed)
    cfg = config_json(shape)
    chrs = list({"hs38": HS38_CHR, "tair10": TAIR10_CHR, "flybase6": FLY_CHR}[shape]) + [(f"scaffold_{k}", 0.06) for k in range(scaffolds)]
    names = [c for c, _ in chrs]
    clen = np.array([int(m * 1e6) for _, m in chrs], dtype=np.int64)
    total = clen.sum()
    if shape == "hs38":
        main = [("protein_coding", "gene", int(21000 * scale), True, 11.0, 140, 0.55, 1600, 1.45),
                ("processed_transcript", "gene", int(14000 * scale), False, 3.5, 180, 0.6, 2500, 1.4)]
        small = [(s, "gene", int(n * scale)) for s, n in zip(_SMALL_HS, [1900, 950, 1900, 7500, 120, 550, 2200, 500, 10200, 60, 2600, 300])]
        small_len = (60, 2000)
    elif shape == "tair10":
        main = [("TAIR10", "gene", int(27400 * scale), True, 5.6, 170, 0.6, 120, 0.75)]
        small = [("TAIR10", s, int(n * scale)) for s, n in zip(_SMALL_TAIR, [180, 690, 71, 13, 15, 3900, 920])]
        small_len = (60, 4000)
    else:
        main = [("FlyBase", "gene", int(13900 * scale), True, 4.0, 250, 0.8, 100, 1.5)]
        small = [("FlyBase", s, int(n * scale)) for s, n in zip(_SMALL_FLY, [480, 310, 290, 30, 110, 2500, 260])]
        small += [("piRNA", "piRNA", int(400 * scale)), ("FlyBase", "pseudogene", int(330 * scale))]
        small_len = (60, 3000)
    vic = {s for s, t, *_ in main}
    out_chr, out_s, out_e, out_t, out_sd = [], [], [], [], []
    genes = []

    def place(n):
        g = np.sort(rng.integers(0, total, n))
        c = np.searchsorted(np.cumsum(clen), g, side="right")
        off = g - np.concatenate([[0], np.cumsum(clen)])[c]
        return c.astype(np.int64), off.astype(np.int64) + 1

    def emit(c, s, e, t, sd):
        if t >= 0:
            out_chr.append(c); out_s.append(s); out_e.append(e); out_t.append(t); out_sd.append(sd)

    gid = 0
    for (src, typ, n, coding, exm, exmed, exsig, inmed, insig) in main:
        cc, pos = place(n)
        t_cds, t_u5, t_u3 = _elem_id(cfg, src, "CDS"), _elem_id(cfg, src, "5'UTR"), _elem_id(cfg, src, "3'UTR")
        t_in, t_up, t_dn, t_g = _elem_id(cfg, src, "intron"), _elem_id(cfg, src, "upstream"), _elem_id(cfg, src, "downstream"), _elem_id(cfg, src, typ)
        for k in range(n):
            c, p = int(cc[k]), int(pos[k])
            ne = 1 + min(int(rng.poisson(exm - 1.0)), 80)
            exl = np.clip(rng.lognormal(np.log(exmed), exsig, ne), 25, 6000).astype(np.int64)
            inl = np.clip(rng.lognormal(np.log(inmed), insig, max(ne - 1, 0)), 40, 900000).astype(np.int64)
            span = int(exl.sum() + inl.sum())
            if p + span >= clen[c]:
                p = max(1, int(clen[c]) - span - 1)
                if p + span >= clen[c]:
                    continue
            es = np.empty(ne, dtype=np.int64)
            es[0] = p
            if ne > 1:
                es[1:] = p + np.cumsum(exl[:-1] + inl)
            ee = es + exl - 1
            sd = STRAND_F if rng.random() < 0.5 else STRAND_R
            gs, ge = int(es[0]), int(ee[-1])
            c0 = c1 = None
            if coding:
                T = int(exl.sum())
                u5 = min(int(rng.lognormal(np.log(150), 0.8)), T // 4)
                u3 = min(int(rng.lognormal(np.log(500), 0.9)), T // 3)
                if sd == STRAND_R:
                    u5, u3 = u3, u5                                     # genomic-left UTR is the 3'UTR on '-'
                cum = np.concatenate([[0], np.cumsum(exl)])

                def to_genomic(x):                                       # exonic offset -> genomic coordinate
                    j = int(np.searchsorted(cum, x, side="right") - 1)
                    j = min(j, ne - 1)
                    return int(es[j] + (x - cum[j]))
                c0, c1 = to_genomic(u5), to_genomic(T - u3 - 1)
                if c1 < c0:
                    c0, c1 = gs, ge
                for a, b in zip(es, ee):                                 # CDS = exons ∩ [c0,c1]  (Gene::setCds, mmannot.cpp:931-936)
                    lo, hi = max(int(a), c0), min(int(b), c1)
                    if lo <= hi:
                        emit(c, lo, hi, t_cds, sd)
                left = [(max(int(a), gs), min(int(b), c0 - 1)) for a, b in zip(es, ee)]
                right = [(max(int(a), c1 + 1), min(int(b), ge)) for a, b in zip(es, ee)]
                u5l, u3l = (left, right) if sd == STRAND_F else (right, left)   # swapped on '-' (mmannot.cpp:943)
                for lo, hi in u5l:
                    if lo <= hi:
                        emit(c, lo, hi, t_u5, sd)
                for lo, hi in u3l:
                    if lo <= hi:
                        emit(c, lo, hi, t_u3, sd)
            for j in range(ne - 1):                                       # introns (mmannot.cpp:743-751)
                emit(c, int(ee[j]) + 1, int(es[j + 1]) - 1, t_in, sd)
            if src in vic:                                                # Gene::setUpDownStream mmannot.cpp:945-954
                if sd == STRAND_F:
                    emit(c, 1 if gs <= up else gs - up, gs - 1, t_up, sd)
                    emit(c, ge + 1, ge + down, t_dn, sd)
                else:
                    emit(c, ge + 1, ge + up, t_up, sd)
                    emit(c, 1 if gs <= down else gs - down, gs - 1, t_dn, sd)
            for a, b in zip(es, ee):
                emit(c, int(a), int(b), t_g, sd)
            if keep_genes:
                genes.append((gid, c, src, typ, sd, es.copy(), ee.copy(), c0, c1))
            gid += 1
    for (src, typ, n) in small:
        cc, pos = place(n)
        ln = rng.integers(small_len[0], small_len[1], n)
        t_g = _elem_id(cfg, src, typ)
        for k in range(n):
            c, p = int(cc[k]), int(pos[k])
            e = min(p + int(ln[k]) - 1, int(clen[c]))
            sd = STRAND_F if rng.random() < 0.5 else STRAND_R
            emit(c, p, e, t_g, sd)
            if keep_genes:
                genes.append((gid, c, src, typ, sd, np.array([p], dtype=np.int64), np.array([e], dtype=np.int64), None, None))
            gid += 1
    chr_ = np.array(out_chr, dtype=np.int32); st = np.array(out_s, dtype=np.int64); en = np.array(out_e, dtype=np.int64)
    ty = np.array(out_t, dtype=np.int32); sdv = np.array(out_sd, dtype=np.uint8)
    order = np.lexsort((np.arange(len(st)), st, chr_))
    return Annotation(shape, cfg, names, clen, chr_[order], st[order], en[order], ty[order], sdv[order], genes if keep_genes else None)


def gtf_text(an: Annotation) -> str:
    """Ensembl-style GTF of the gene models (needs keep_genes=True), genes in chromosome/creation order."""
    assert an.genes is not None
    rows = []
    for (gid, c, src, typ, sd, es, ee, c0, c1) in sorted(an.genes, key=lambda g: (g[1], int(g[5][0]), g[0])):
        ch, s = an.chr_names[c], "+" if sd == STRAND_F else "-"
        attr = f'gene_id "G{gid:07d}"; transcript_id "T{gid:07d}";'
        gtype = typ
        rows.append(f"{ch}\t{src}\t{gtype}\t{int(es[0])}\t{int(ee[-1])}\t.\t{s}\t.\tgene_id \"G{gid:07d}\";")
        if typ == "gene":
            rows.append(f"{ch}\t{src}\ttranscript\t{int(es[0])}\t{int(ee[-1])}\t.\t{s}\t.\t{attr}")
            for a, b in zip(es, ee):
                rows.append(f"{ch}\t{src}\texon\t{int(a)}\t{int(b)}\t.\t{s}\t.\t{attr}")
                if c0 is not None:
                    lo, hi = max(int(a), c0), min(int(b), c1)
                    if lo <= hi:
                        rows.append(f"{ch}\t{src}\tCDS\t{lo}\t{hi}\t.\t{s}\t0\t{attr}")
    return "\n".join(rows) + "\n"
