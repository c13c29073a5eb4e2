#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — generates the committed golden fixtures under tests/golden/
by running the *compiled reference* (oracle/_ref/mmannot*, built by oracle/build_ref.sh from
/root/reference/mmannot.cpp) in the build container.  Only derived data is written: sorted
interval dumps, alignment tuples, count tables, stderr statistics, per-hit region sets.
No reference source or script text is stored.

Usage:  python oracle/gen_golden.py [test_dataset] [kat] [synth]
"""
from __future__ import annotations

import gzip
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import mm_oracle as mo  # noqa: E402

REF = os.environ.get("REF", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

AMB_CONFIG = """Introns:
protein_coding:gene
Vicinity:
protein_coding:gene
Order:
protein_coding:CDS,protein_coding:5'UTR,protein_coding:3'UTR
rRNA:gene,snRNA:gene,misc_RNA:gene,lincRNA:gene
processed_pseudogene:gene -,unprocessed_pseudogene:gene +,protein_coding:intron
protein_coding:upstream,protein_coding:downstream,protein_coding:gene
"""


def write_iv_dump(iv, path):
    with gzip.open(path, "wt") as f:
        f.write("#idx\tchrId\tchrName\tstart\tend\ttype\tstrand\tid\n")
        for i in range(len(iv["start"])):
            f.write(f"{i}\t{iv['chr'][i]}\t{iv['chr_name'][i]}\t{iv['start'][i]}\t{iv['end'][i]}\t{iv['type'][i]}\t{iv['strand'][i]}\t{iv['id'][i]}\n")


def gen_test_dataset():
    out = os.path.join(GOLD, "test_dataset")
    os.makedirs(out, exist_ok=True)
    gtf = os.path.join(REF, "test_dataset.gtf")
    bam = os.path.join(REF, "test_dataset.bam")
    tmp = tempfile.mkdtemp(prefix="mmgold")
    amb_cfg = os.path.join(tmp, "cfg_amb.txt")
    open(amb_cfg, "w").write(AMB_CONFIG)
    configs = {"hs38": os.path.join(REF, "configHS38.txt"), "amb": amb_cfg}
    for k, p in configs.items():
        desc = mo.parse_config_order(open(p).read())
        json.dump(desc.to_json(), open(os.path.join(out, f"config_{k}.json"), "w"), indent=0)
    json.dump({"amb_config_text": AMB_CONFIG}, open(os.path.join(out, "config_amb_text.json"), "w"))

    # (case name, variant, config key, extra args)
    cases = [
        ("U_default", "mmannot_dump", "hs38", ["-s", "U"]),
        ("U_l0", "mmannot_dump", "hs38", ["-s", "U", "-l", "0"]),
        ("U_l1", "mmannot_dump", "hs38", ["-s", "U", "-l", "1"]),
        ("U_l10", "mmannot_dump", "hs38", ["-s", "U", "-l", "10"]),
        ("U_l0.5", "mmannot_dump", "hs38", ["-s", "U", "-l", "0.5"]),
        ("U_l0.99", "mmannot_dump", "hs38", ["-s", "U", "-l", "0.99"]),
        ("U_unique", "mmannot_dump", "hs38", ["-s", "U", "-y", "unique"]),
        ("U_ratio", "mmannot_dump", "hs38", ["-s", "U", "-y", "ratio"]),
        ("U_random", "mmannot_dump", "hs38", ["-s", "U", "-y", "random"]),
        ("U_ratio_l0.5", "mmannot_dump", "hs38", ["-s", "U", "-y", "ratio", "-l", "0.5"]),
        ("U_d0D0", "mmannot_dump", "hs38", ["-s", "U", "-d", "0", "-D", "0"]),
        ("U_d5000D200", "mmannot_dump", "hs38", ["-s", "U", "-d", "5000", "-D", "200"]),
        ("pF_default", "mmannot_patched_dump", "hs38", ["-s", "F"]),
        ("pR_default", "mmannot_patched_dump", "hs38", ["-s", "R"]),
        ("pF_ratio", "mmannot_patched_dump", "hs38", ["-s", "F", "-y", "ratio"]),
        ("pF_l5", "mmannot_patched_dump", "hs38", ["-s", "F", "-l", "5"]),
        ("pR_unique_l0.3", "mmannot_patched_dump", "hs38", ["-s", "R", "-y", "unique", "-l", "0.3"]),
        ("amb_U_default", "mmannot_dump", "amb", ["-s", "U", "-d", "5000", "-D", "5000"]),
        ("amb_U_l0.3", "mmannot_dump", "amb", ["-s", "U", "-d", "5000", "-D", "5000", "-l", "0.3"]),
        ("amb_U_l20", "mmannot_dump", "amb", ["-s", "U", "-d", "5000", "-D", "5000", "-l", "20"]),
        ("amb_pF_l1", "mmannot_patched_dump", "amb", ["-s", "F", "-d", "5000", "-D", "5000", "-l", "1"]),
        ("amb_pR_ratio", "mmannot_patched_dump", "amb", ["-s", "R", "-d", "5000", "-D", "5000", "-y", "ratio"]),
    ]
    meta = {}
    hitmasks = {}
    index_written = {}
    reads_saved = False
    for name, variant, ck, extra in cases:
        dump = os.path.join(tmp, name + ".dump")
        tsv = os.path.join(tmp, name + ".tsv")
        rc, so, se = mo.run_ref(variant, ["-a", gtf, "-r", bam, "-c", configs[ck], "-o", tsv] + extra, dump=dump)
        assert rc == 0, (name, se)
        iv, hits = mo.parse_dump(dump)
        # index fixture per (config, -d, -D)
        d = extra[extra.index("-d") + 1] if "-d" in extra else "1000"
        D = extra[extra.index("-D") + 1] if "-D" in extra else "1000"
        ikey = f"{ck}_d{d}_D{D}"
        if ikey not in index_written:
            write_iv_dump(iv, os.path.join(out, f"index_{ikey}.tsv.gz"))
            index_written[ikey] = True
        # true record strands come from the patched -s F dump (strandF(record.strand) = record.strand)
        if name == "pF_default" and not reads_saved:
            chr_names = []
            for c, nm in zip(iv["chr"], iv["chr_name"]):
                while len(chr_names) <= c:
                    chr_names.append(None)
                chr_names[c] = nm
            cmap = {nm: i for i, nm in enumerate(chr_names)}
            qid = mo.names_to_qid(hits["name"])
            np.savez_compressed(os.path.join(out, "reads.npz"),
                                tid=np.array([cmap.get(c, -1) for c in hits["chr_name"]], dtype=np.int32),
                                start=np.array(hits["start"], dtype=np.int64), end=np.array(hits["end"], dtype=np.int64),
                                strand=np.array(hits["strand"], dtype=np.uint8), nh=np.array(hits["nh"], dtype=np.uint32),
                                qid=qid.astype(np.int64))
            json.dump({"chromosomes": chr_names}, open(os.path.join(out, "chromosomes.json"), "w"))
            reads_saved = True
        hitmasks[name] = np.array(hits["mask"], dtype=np.uint64)
        meta[name] = dict(variant=variant, config=ck, index=ikey, args=extra, table=open(tsv).read(),
                          stderr=se.replace(REF, "$REF"), n_hit_lines=len(hits["mask"]))
        print(name, "rows", meta[name]["table"].count("\n") - 1, file=sys.stderr)
    # reference failure modes that the CLI must reproduce
    rc, so, se = mo.run_ref("mmannot", ["-a", gtf, "-r", bam, "-c", os.path.join(REF, "configTest.txt"), "-s", "U"])
    meta["fail_configTest"] = dict(rc=rc, stderr=se.replace(REF, "$REF"), stdout=so)
    rc, so, se = mo.run_ref("mmannot", ["-a", gtf, "-r", bam, "-c", configs["hs38"], "-s", "FR"])
    meta["fail_sFR"] = dict(rc=rc, stderr=se.replace(REF, "$REF"), stdout=so)
    json.dump(meta, open(os.path.join(out, "cases.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(out, "hitmasks.npz"), **hitmasks)


def gen_builder():
    """Interval dumps that pin the product's config+GTF builder (mm_annot.cpp) beyond the HS38 config:
    the wildcard/synonym config shipped as config.txt on the chrY GTF, and seeded synthetic GTFs
    (mmannot_amd.synth.gtf_text) under their own configs, also with -d/-D variants."""
    from mmannot_amd import synth
    out = os.path.join(GOLD, "builder")
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="mmgoldb")
    bam = os.path.join(REF, "test_dataset.bam")
    meta = {}

    def one(name, cfg_path, gtf_path, extra):
        dump = os.path.join(tmp, name + ".dump")
        rc, so, se = mo.run_ref("mmannot_dump", ["-a", gtf_path, "-r", bam, "-c", cfg_path, "-s", "U", "-o", os.path.join(tmp, "t.tsv")] + extra, dump=dump)
        meta[name] = dict(rc=rc, stderr_head=se.split("Reading BAM")[0].replace(REF, "$REF").replace(tmp, "$TMP"), args=extra)
        if rc == 0:
            iv, _ = mo.parse_dump(dump)
            write_iv_dump(iv, os.path.join(out, f"index_{name}.tsv.gz"))
        print("builder", name, "rc", rc, file=sys.stderr)

    one("chrY_config_txt", os.path.join(REF, "config.txt"), os.path.join(REF, "test_dataset.gtf"), [])
    one("chrY_configTest", os.path.join(REF, "configTest.txt"), os.path.join(REF, "test_dataset.gtf"), [])
    for shape, seed, scale in (("tair10", 5, 0.02), ("hs38", 6, 0.01)):
        an = synth.synth_annotation(shape, seed=seed, scale=scale, keep_genes=True)
        gtf = os.path.join(tmp, f"{shape}.gtf"); open(gtf, "w").write(synth.gtf_text(an))
        cfgp = os.path.join(tmp, f"{shape}.cfg"); open(cfgp, "w").write(synth.config_text(shape))
        one(f"synth_{shape}_s{seed}", cfgp, gtf, [])
        one(f"synth_{shape}_s{seed}_d300_D50", cfgp, gtf, ["-d", "300", "-D", "50"])
        meta[f"synth_{shape}_s{seed}"]["synth"] = dict(shape=shape, seed=seed, scale=scale)
        meta[f"synth_{shape}_s{seed}_d300_D50"]["synth"] = dict(shape=shape, seed=seed, scale=scale)
    # attribute-column syntax the GTF parser has to read the way the reference does: GFF3 style, missing quotes / semicolons,
    # duplicate keys, comma lists, an unterminated quote, blanks around '='  (an attribute column without any separator makes
    # the reference itself crash - SIGSEGV after "cannot deduce id" - so that case has no oracle)
    quirks_cfg = "Introns:\nsrc:gene\nVicinity:\nsrc:gene\nOrder:\nsrc:CDS +\nsrc:5'UTR +\nsrc:3'UTR +\nsrc:featA\nsrc:featB -\nsrc:gene\nsrc:intron\nsrc:upstream,src:downstream\n"
    rows = [
        ("c1", "src", "featA", 1000, 1500, "+", "ID=gene1;Name=foo"),
        ("c1", "src", "featA", 2000, 2500, "-", 'gene_id "g2"; transcript_id "t2"'),
        ("c1", "src", "featA", 3000, 3500, "+", "gene_id g3; transcript_id t3;"),
        ("c1", "src", "featA", 4000, 4500, "+", 'gene_id "g4a"; gene_id "g4b";'),
        ("c1", "src", "featB", 5000, 5500, "-", "ID=a5,b5;Parent=p5,q5"),
        ("c1", "src", "featB", 6000, 6500, "+", 'gene_id "g6; transcript_id "t6";'),
        ("c1", "src", "featA", 8000, 8500, "+", "ID = g8 ; Parent = p8"),
        ("c1", "src", "featA", 9000, 9500, "+", "Parent=p9.1"),
        ("c1", "src", "featB", 9600, 9900, "-", 'transcript_id "t10" ;   gene_name "x y z";  '),
        ("c1", "src", "gene", 20000, 26000, "+", 'gene_id "G1"; gene_name "with ; semicolon";'),
        ("c1", "src", "transcript", 20000, 26000, "+", 'gene_id "G1"; transcript_id "T1";'),
        ("c1", "src", "exon", 20000, 21000, "+", 'gene_id "G1"; transcript_id "T1"; exon_number "1";'),
        ("c1", "src", "exon", 23000, 26000, "+", "Parent=T1"),
        ("c1", "src", "CDS", 20500, 21000, "+", 'transcript_id "T1"'),
        ("c1", "src", "CDS", 23000, 24000, "+", "Parent=G1,Tx"),
        ("c2", "src", "gene", 500, 9000, "-", "ID=G2"),
        ("c2", "src", "mRNA", 500, 9000, "-", "ID=T2;Parent=G2"),
        ("c2", "src", "exon", 500, 1500, "-", "Parent=T2"),
        ("c2", "src", "exon", 1400, 3000, "-", "Parent=T2"),
        ("c2", "src", "exon", 8000, 9000, "-", "Parent = T2 ;"),
        ("c2", "src", "CDS", 1000, 8500, "-", "Parent=T2"),
        ("c2", "other", "thing", 100, 200, "+", 'gene_id "unused1";'),
        ("c2", "src", "exon", 150, 180, "+", 'gene_id "unused1";'),
    ]
    quirks_gtf = "".join("\t".join([c, so, ty, str(a), str(b), ".", sd, ".", attr]) + "\n" for c, so, ty, a, b, sd, attr in rows)
    gq = os.path.join(tmp, "quirks.gtf"); open(gq, "w").write(quirks_gtf)
    cq = os.path.join(tmp, "quirks.cfg"); open(cq, "w").write("Synonyms:\nmRNA:transcript\n" + quirks_cfg)
    one("quirks", cq, gq, [])
    meta["quirks"]["gtf_text"] = quirks_gtf
    meta["quirks"]["cfg_text"] = "Synonyms:\nmRNA:transcript\n" + quirks_cfg
    json.dump(meta, open(os.path.join(out, "cases.json"), "w"), indent=1)


if __name__ == "__main__":
    what = sys.argv[1:] or ["test_dataset"]
    if not mo.have_ref():
        sys.exit("oracle/_ref not built (run oracle/build_ref.sh in the build container)")
    if "test_dataset" in what:
        gen_test_dataset()
    if "builder" in what:
        gen_builder()
    if "kat" in what:
        from oracle import gen_kat
        gen_kat.main()
    if "synth" in what:
        from oracle import gen_synth_golden
        gen_synth_golden.main()
