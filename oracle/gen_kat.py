#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — edge-case known-answer tests (SURVEY.md section 8c, E1..E10 and more).
Inputs are hand-built here (tiny GTF / config / BAM / SAM files, committed under tests/golden/kat/<name>/);
expected outputs come from running the compiled reference (oracle/_ref) on them in the build container:
count table, stderr, exit code, the per-hit dump (tuples + element sets) and the -m file."""
from __future__ import annotations

import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import mm_oracle as mo  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "kat")

GTF_ABC = ("chr1\tsrc\tfeatA\t1000\t2000\t.\t+\t.\tgene_id \"A\";\n"
           "chr1\tsrc\tfeatB\t3000\t4000\t.\t-\t.\tgene_id \"B\";\n"
           "chr1\tsrc\tfeatC\t5000\t6000\t.\t+\t.\tgene_id \"C\";\n")
CFG_ABC = "Order:\nsrc:featA\nsrc:featB\nsrc:featC\n"
CHROMS = [("chr1", 100000), ("chrZ", 50000)]


def rec(name, tid, pos1, cigar=(("M", 50),), flag=0, nh=None, nh_type="C", extra=b"", seq_len=None):
    sl = sum(l for op, l in cigar if op in "MIS=X") if seq_len is None else seq_len
    return mo._bam_record(tid, pos1 - 1, flag, name, cigar, sl, mo.tag_nh(nh, nh_type) + extra)


def tagZ(key, s):
    return key.encode() + b"Z" + s.encode() + b"\0"


def tagC(key, v):
    return key.encode() + b"C" + bytes([v])


KATS = {}

KATS["E1_overcomplete"] = dict(gtf=GTF_ABC, cfg=CFG_ABC, bam=[rec("a", 0, 1100, nh=2), rec("a", 0, 3100, nh=2), rec("a", 0, 5100, nh=2)],
                               runs=[["-s", "U"], ["-s", "U", "-y", "unique"], ["-s", "U", "-y", "ratio"]])
KATS["E2_mixed_nh"] = dict(gtf=GTF_ABC, cfg=CFG_ABC, bam=[rec("b", 0, 1100, nh=3), rec("b", 0, 3100, nh=1), rec("b", 0, 1200, nh=3), rec("b", 0, 5100, nh=3)],
                           runs=[["-s", "U"], ["-s", "U", "-y", "ratio"]])
KATS["E3_cigar"] = dict(gtf=GTF_ABC, cfg=CFG_ABC,
                        bam=[rec("c1", 0, 1100, cigar=(("M", 10), ("N", 3000), ("M", 10))), rec("c2", 0, 1995, cigar=(("M", 10),)),
                             rec("c3", 0, 1991, cigar=(("S", 5), ("M", 8), ("D", 2), ("I", 3))), rec("c4", 0, 1500, cigar=(), seq_len=0),
                             rec("c5", 0, 2001, cigar=(), seq_len=0), rec("c6", 0, 1000, cigar=(("=", 20), ("X", 5), ("P", 2), ("H", 3)))],
                        runs=[["-s", "U"], ["-s", "U", "-l", "1"]])
KATS["E4_nh_types"] = dict(gtf=GTF_ABC, cfg=CFG_ABC,
                           bam=[rec("n1", 0, 1100, nh=1, nh_type="c"), rec("n2", 0, 3100, nh=1, nh_type="i"), rec("n3", 0, 5100), rec("n4", 0, 5200, nh=300, nh_type="S"),
                                rec("n5", 0, 1300, nh=2, nh_type="I"), rec("n5", 0, 1400, nh=2, nh_type="C")],
                           runs=[["-s", "U"], ["-s", "U", "-y", "unique"], ["-s", "U", "-y", "ratio"]])
KATS["E5_placement"] = dict(gtf=GTF_ABC, cfg=CFG_ABC,
                            bam=[rec("p1", -1, 0, flag=4, cigar=(), seq_len=0), rec("p2", 1, 100), rec("p3", 1, 200), rec("p4", 0, 1100, flag=4), rec("p5", 0, 90000)],
                            runs=[["-s", "U"], ["-s", "U", "-y", "unique"]])
KATS["E7_overlap"] = dict(gtf=GTF_ABC, cfg=CFG_ABC,
                          bam=[rec("o1", 0, 1991, cigar=(("M", 11),)), rec("o2", 0, 2000, cigar=(("M", 10),)), rec("o3", 0, 1999, cigar=(("M", 10),))],
                          runs=[["-s", "U", "-l", v] for v in ("0", "1", "2", "9", "10", "0.8", "0.82", "0.9")])
GTF_E8a = ("chr1\tsrc\tfeatA\t1000\t2000\t.\t+\t.\tgene_id \"A1\";\nchr1\tsrc\tfeatA\t1000\t1500\t.\t+\t.\tgene_id \"A2\";\n"
           "chr1\tsrc\tfeatB\t1000\t1800\t.\t+\t.\tgene_id \"B1\";\n")
GTF_E8b = ("chr1\tsrc\tfeatA\t1000\t1500\t.\t+\t.\tgene_id \"A2\";\nchr1\tsrc\tfeatA\t1000\t2000\t.\t+\t.\tgene_id \"A1\";\n"
           "chr1\tsrc\tfeatB\t1000\t1800\t.\t+\t.\tgene_id \"B1\";\n")
KATS["E8a_tie_order"] = dict(gtf=GTF_E8a, cfg="Order:\nsrc:featA,src:featB\n", bam=[rec("t1", 0, 1400, cigar=(("M", 201),))], runs=[["-s", "U", "-l", "1"], ["-s", "U"]])
KATS["E8b_tie_order"] = dict(gtf=GTF_E8b, cfg="Order:\nsrc:featA,src:featB\n", bam=[rec("t1", 0, 1400, cigar=(("M", 201),))], runs=[["-s", "U", "-l", "1"], ["-s", "U"]])
GTF_E9 = ("chr1\tpc\tgene\t10000\t11000\t.\t+\t.\tgene_id \"G1\";\nchr1\tpc\tgene\t13000\t14000\t.\t+\t.\tgene_id \"G2\";\n")
KATS["E9_distance"] = dict(gtf=GTF_E9, cfg="Vicinity:\npc:gene\nOrder:\npc:gene\npc:upstream,pc:downstream\n",
                           bam=[rec("d1", 0, 11800), rec("d2", 0, 12000), rec("d3", 0, 11975, cigar=(("M", 51),))],
                           runs=[["-s", "U", "-d", "1500", "-D", "1500"], ["-s", "U", "-d", "1500", "-D", "1500", "-l", "10"]])
KATS["E10_rescue"] = dict(gtf=GTF_ABC, cfg=CFG_ABC, bam=[rec("r", 0, 1100, nh=4), rec("r", 0, 1200, nh=4), rec("r", 0, 3100, nh=4), rec("r", 0, 1300, nh=4),
                                                         rec("q", 0, 1100, nh=2), rec("q", 0, 5100, nh=2), rec("u", 0, 5200)],
                          runs=[["-s", "U", "-e", "75", "-m", "$M"], ["-s", "U", "-e", "80", "-m", "$M"], ["-s", "U", "-e", "75"], ["-s", "U", "-m", "$M"],
                                ["-s", "U", "-e", "50", "-m", "$M", "-y", "ratio"], ["-s", "U", "-M", "$MM"], ["-s", "U", "-m", "$M", "-M", "$MM", "-e", "60"],
                                ["-s", "U", "-y", "random"], ["-s", "U", "-y", "random", "-m", "$M", "-M", "$MM"]])
# BWA-style XA alternatives: equal-NM alternatives are replayed as further reads, nHits = alternatives + 1; tag order matters
KATS["X1_xa"] = dict(gtf=GTF_ABC, cfg=CFG_ABC,
                     bam=[rec("x1", 0, 1100, extra=tagC("NM", 1) + tagZ("XA", "chr1,-3100,50M,1;chr1,+5100,50M,2;chrZ,+10,50M,1;")),
                          rec("x2", 0, 3200, extra=tagZ("XA", "chr1,+1200,50M,1;") + tagC("NM", 0)),
                          rec("x3", 0, 5300, nh=7, extra=tagC("NM", 0) + tagZ("XA", "chr1,+1250,20M5D20M,0;"))],
                     runs=[["-s", "U"], ["-s", "U", "-y", "unique"]])
SAM = ("@HD\tVN:1.0\n@SQ\tSN:chr1\tLN:100000\n"
       "s1\t0\tchr1\t1100\t255\t50M\t*\t0\t0\t*\t*\tNH:i:1\n"
       "s2\t16\tchr1\t3100\t255\t20M10D20M\t*\t0\t0\t*\t*\tNH:i:2\n"
       "s2\t0\tchr1\t5100\t255\t50M\t*\t0\t0\t*\t*\tNH:i:2\n"
       "s3\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\tNH:i:1\n"
       "s4\t0\tchr1\t1200\t255\t*\t*\t0\t0\t*\t*\tNM:i:0\n"
       "s5\t0\tchr1\t1950\t255\t10M100N50M\t*\t0\t0\t*\t*\tNM:i:0\tXA:Z:chr1,+1300,50M,0;\n")
KATS["S1_sam"] = dict(gtf=GTF_ABC, cfg=CFG_ABC, sam=SAM, runs=[["-s", "U"], ["-s", "U", "-l", "0.5"], ["-s", "U", "-M", "$MM"]])
# -M on overlapping intervals of one element (several intervals behind one hit) and on multi-hit reads
GTF_M = ("chr1\tsrc\tfeatA\t1000\t2000\t.\t+\t.\tgene_id \"A1\";\nchr1\tsrc\tfeatA\t1200\t2500\t.\t+\t.\tgene_id \"A2\";\n"
         "chr1\tsrc\tfeatB\t3000\t4000\t.\t-\t.\tgene_id \"B\";\nchr1\tsrc\tfeatA\t3100\t3300\t.\t+\t.\tgene_id \"A3\";\n")
KATS["M1_intervals"] = dict(gtf=GTF_M, cfg="Order:\nsrc:featA,src:featB\n",
                            bam=[rec("m1", 0, 1300), rec("m2", 0, 1100), rec("m3", 0, 3150, nh=2), rec("m3", 0, 1300, nh=2), rec("m4", 0, 3150), rec("m5", 0, 3500),
                                 rec("m6", 0, 1300, nh=3), rec("m6", 0, 9000, nh=3)],
                            runs=[["-s", "U", "-M", "$MM"], ["-s", "U", "-l", "10", "-M", "$MM", "-m", "$M"]])


def main():
    if not mo.have_ref():
        sys.exit("oracle/_ref not built")
    os.makedirs(GOLD, exist_ok=True)
    for name, k in KATS.items():
        d = os.path.join(GOLD, name)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "a.gtf"), "w").write(k["gtf"])
        open(os.path.join(d, "c.cfg"), "w").write(k["cfg"])
        if "bam" in k:
            reads = os.path.join(d, "r.bam")
            mo.write_bam(reads, CHROMS, k["bam"], level=6)
        else:
            reads = os.path.join(d, "r.sam")
            open(reads, "w").write(k["sam"])
        cases = []
        tmp = tempfile.mkdtemp(prefix="mmkat")
        for run in k["runs"]:
            mfile = os.path.join(tmp, "m.txt")
            mmfile = os.path.join(tmp, "mm.txt")
            args = [mfile if a == "$M" else (mmfile if a == "$MM" else a) for a in run]
            for variant, tag in (("mmannot_dump", "verbatim"), ("mmannot_patched_dump", "patched")):
                a2 = list(args)
                if tag == "patched":                 # the patched reference gives -s F/R their intended meaning
                    a2[a2.index("-s") + 1] = "F"
                dump = os.path.join(tmp, "d.txt")
                tsv = os.path.join(tmp, "t.tsv")
                for f in (mfile, mmfile, dump, tsv):
                    if os.path.exists(f):
                        os.remove(f)
                rc, so, se = mo.run_ref(variant, ["-a", os.path.join(d, "a.gtf"), "-r", reads, "-c", os.path.join(d, "c.cfg"), "-o", tsv] + a2, dump=dump, cwd=tmp)
                iv, hits = mo.parse_dump(dump) if os.path.exists(dump) else ({}, dict(name=[], chr_name=[], start=[], end=[], strand=[], nh=[], mask=[]))
                cases.append(dict(args=["$M" if a == mfile else ("$MM" if a == mmfile else a) for a in a2], variant=tag,
                                  mmfile=open(mmfile).read() if (os.path.exists(mmfile) and "$MM" in run) else None, rc=rc, table=open(tsv).read() if os.path.exists(tsv) else None,
                                  stderr=se.replace(d, "$D").replace(tmp, "$T"), mfile=open(mfile).read() if (os.path.exists(mfile) and "$M" in run) else None,
                                  hits=dict(name=hits["name"], chr=hits["chr_name"], start=hits["start"], end=hits["end"], strand=hits["strand"],
                                            nh=hits["nh"], mask=[int(m) for m in hits["mask"]])))
        json.dump(cases, open(os.path.join(d, "cases.json"), "w"), indent=1)
        print("kat", name, [(c["rc"], (c["table"] or "").count("\n") - 1) for c in cases], file=sys.stderr)


if __name__ == "__main__":
    main()
