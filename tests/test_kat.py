"""Edge-case known-answer tests (inputs hand-built by oracle/gen_kat.py, expected outputs from the
compiled reference).  CPU part: the product's host code (GTF/config builder + BAM/SAM decoder) feeds the CPU oracle and
must reproduce the reference's tuples, table and statistics.  GPU part: the `mmannot-amd` binary end to end."""
import json
import os
import subprocess

import numpy as np
import pytest

from golden_util import check_read_stats_file

import golden_util as gu

KAT = os.path.join(gu.GOLD, "kat")
NAMES = sorted(os.listdir(KAT))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "mmannot_amd", "mmannot-amd")


def load(name):
    d = os.path.join(KAT, name)
    reads = os.path.join(d, "r.bam") if os.path.exists(os.path.join(d, "r.bam")) else os.path.join(d, "r.sam")
    return d, reads, json.load(open(os.path.join(d, "cases.json")))


def opts(args):
    a = [x for x in args]
    for flag in ("-m", "-e", "-M"):
        while flag in a:
            i = a.index(flag); del a[i:i + 2]
    return gu.parse_args(a)


@pytest.mark.parametrize("name", NAMES)
def test_kat_host_plus_oracle(oracle, name):
    import mmannot_amd.api as api
    d, reads, cases = load(name)
    for c in cases:
        o = opts(c["args"])
        if "-m" in c["args"] or "-M" in c["args"] or "random" in c["args"]:
            continue                     # side channels (-m/-e, -M, rand() replay) are host bookkeeping of the CLI: covered by the CLI test
        an = api.Annotation(os.path.join(d, "c.cfg"), os.path.join(d, "a.gtf"), o["d"], o["D"])
        compat = 1 if c["variant"] == "verbatim" else 0      # verbatim runs use -s U, where the strand does not matter
        dec = api.ReadsDecoder(reads, an, unique_only=(o["strategy"] == "unique"), strand_compat=compat)
        aln, nh = dec.read_all(cap=3)    # tiny capacity: batches split records from their XA alternatives
        log = dec.take_log()
        # tuples == what the reference handed to scan()
        h = c["hits"]
        keep = np.ones(len(aln), dtype=bool) if o["strategy"] != "unique" else (nh == 1)
        assert (aln["start"][keep] == np.array(h["start"], dtype=np.int64)).all()
        want_end = np.array([e if e < 2**63 else e - 2**64 for e in h["end"]], dtype=np.int64)   # start 0 without CIGAR: the reference wraps to 2^64-1
        assert (aln["end"][keep].astype(np.int64) == want_end).all()
        assert (nh[keep] == np.array(h["nh"], dtype=np.int64)).all()
        if c["variant"] == "patched" and len(aln):
            assert (api.aln_strand(aln)[keep] == np.array(h["strand"])).all()
        names = [dec.name(int(q)) for q in aln["qid"]] if False else None
        # read names -> dense ids in first-appearance order
        want_q = {}
        for nm in h["name"]:
            want_q.setdefault(nm, len(want_q))
        if o["strategy"] != "unique":
            assert aln["qid"].tolist() == [want_q[nm] for nm in h["name"]]
        # every stderr line of the decoder appears in the reference's stderr, in order
        pos = 0
        ref_err = c["stderr"].replace("$D", d)
        for ln in log.splitlines():
            k = ref_err.find(ln, pos)
            assert k >= 0, (ln, ref_err)
            pos = k + len(ln)
        # oracle on the product's host output == reference table and statistics
        cfg = oracle.ConfigDesc([oracle.Element(*(lambda nm, sd, ln: (nm.split(":")[0], ":".join(nm.split(":")[1:]), sd, ln))(
            (n[:-4] if n.endswith((" (+)", " (-)")) else n), int(s), int(l))) for n, s, l in zip(an.names, an.elem_strand, an.line_of)])
        ix = dict(chr=an.chr, start=an.start, end=an.end, type=an.type, strand=an.strand, n_chromosomes=len(an.chromosomes))
        rd = dict(tid=api.aln_tid(aln), start=aln["start"], end=aln["end"], strand=api.aln_strand(aln), nh=nh, qid=aln["qid"])
        r = oracle.run(cfg, ix, rd, overlap=o["overlap"], strandedness=o["strandedness"], strategy=o["strategy"], want_hit_mask=True)
        assert oracle.format_table(cfg, r.row_mask, r.row_count, "r") == c["table"]
        assert oracle.format_stats(r.stats, "$D/" + os.path.basename(reads)) in c["stderr"]
        m = r.hit_mask[r.hit_mask != np.uint64(0xFFFFFFFFFFFFFFFF)]
        assert m.tolist() == h["mask"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_kat_cli(gpu_lib, tmp_path, name):
    d, reads, cases = load(name)
    for c in cases:
        mfile, mmfile = str(tmp_path / "m.txt"), str(tmp_path / "mm.txt")
        args = [mfile if a == "$M" else (mmfile if a == "$MM" else a) for a in c["args"]]
        if c["variant"] == "verbatim":
            args += ["--compat-strand", "plus"]
        out = str(tmp_path / "t.tsv")
        p = subprocess.run([CLI, "-a", os.path.join(d, "a.gtf"), "-r", reads, "-c", os.path.join(d, "c.cfg"), "-o", out] + args,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=dict(os.environ, MM_CLI_MARK_FLUSH="1"))
        assert p.returncode == c["rc"], p.stderr.decode()
        assert open(out).read() == c["table"], (c["args"], p.stderr.decode())
        assert p.stderr.decode().replace(d, "$D") == c["stderr"], c["args"]
        if c["mfile"] is not None:      # reads still pending at end of file are printed in hash-map order by the reference
            check_read_stats_file(open(mfile).read(), c["mfile"])
        if c.get("mmfile") is not None:  # -M output is sorted by the reference: byte-exact
            assert open(mmfile).read() == c["mmfile"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_kat_cli_two_devices(gpu_lib, tmp_path, name):
    """The known-answer cases with the one input file sharded over two contexts (--devices 0,0), two tuples per batch."""
    d, reads, cases = load(name)
    env = dict(os.environ, MM_CLI_BATCH="64")
    for c in cases:
        if any(a in ("-m", "-M") for a in c["args"]) or "random" in c["args"]:
            continue
        args = list(c["args"])
        if c["variant"] == "verbatim":
            args += ["--compat-strand", "plus"]
        out = str(tmp_path / "t.tsv")
        p = subprocess.run([CLI, "-a", os.path.join(d, "a.gtf"), "-r", reads, "-c", os.path.join(d, "c.cfg"), "-o", out, "--devices", "0,0"] + args,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
        assert p.returncode == c["rc"], p.stderr.decode()
        assert open(out).read() == c["table"], (c["args"], p.stderr.decode())
        assert p.stderr.decode().replace(d, "$D") == c["stderr"], c["args"]
