#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — end-to-end goldens on seeded synthetic inputs: a synthetic GTF + configuration
(mmannot_amd.synth) and a synthetic BAM (tests/synth_ref.synth_tuples -> oracle.mm_oracle.bam_records_fixed) are run
through the compiled reference (oracle/_ref); tables and stderr are committed.  The tests regenerate the very same
files from the seeds (nothing but seeds and expected outputs is stored) and run the `mmannot-amd` binary on them."""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mm_oracle as mo  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "synth_e2e")

WORKLOADS = {
    "tair10": dict(shape="tair10", seed=5, scale=0.02, n=60000, read_len=50, nh_max=6, scatter=8, rseed=11),
    "hs38": dict(shape="hs38", seed=6, scale=0.01, n=80000, read_len=100, nh_max=20, scatter=12, rseed=12),
    # BASELINE.json configs[3]: read pairs (mates share a name) against a Flybase6-shaped annotation with introns + vicinity
    "flybase6": dict(shape="flybase6", seed=7, scale=0.05, n=60000, read_len=100, nh_max=1, scatter=0, rseed=13, pair=250),
}
RUNS = [["-s", "F"], ["-s", "R", "-l", "0.5"], ["-s", "U", "-y", "unique"], ["-s", "F", "-y", "ratio"], ["-s", "F", "-l", "30", "-d", "300", "-D", "50"],
        ["-s", "R", "-y", "ratio", "-l", "1"], ["-s", "F", "-M", "$MM"], ["-s", "U", "-y", "random"], ["-s", "F", "-l", "0.3", "-m", "$M", "-M", "$MM", "-e", "60"],
        ["-s", "U", "-l", "50", "-d", "1000", "-D", "1000"]]


def make_inputs(w, d):
    """-> (gtf path, cfg path, bam path); deterministic in the workload parameters"""
    import synth_ref as sr
    from mmannot_amd import synth
    an = synth.synth_annotation(w["shape"], seed=w["seed"], scale=w["scale"], keep_genes=True)
    gtf = os.path.join(d, "a.gtf"); open(gtf, "w").write(synth.gtf_text(an))
    cfg = os.path.join(d, "c.cfg"); open(cfg, "w").write(synth.config_text(w["shape"]))
    # shrink the chromosomes to the annotated span so that enough reads land on features
    span = np.zeros(an.n_chromosomes, dtype=np.int64)
    np.maximum.at(span, an.chr, an.end + 2000)
    span = np.maximum(span, 10000)
    t = sr.synth_tuples(w["rseed"], 0, w["n"], w["read_len"], w["nh_max"], span, w["scatter"], w.get("pair", 0))
    flag = np.where(t["strand"] == 1, 0, 16).astype(np.uint16)
    body = mo.bam_records_fixed(t["tid"], t["start"] - 1, flag, t["qid"], np.full(w["n"], w["read_len"]), t["nh"] if w["nh_max"] > 1 else None)
    bam = os.path.join(d, "r.bam")
    mo.write_bam_bytes(bam, [(c, int(l)) for c, l in zip(an.chr_names, span)], body, level=1)
    return gtf, cfg, bam


def main():
    if not mo.have_ref():
        sys.exit("oracle/_ref not built")
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    for name, w in WORKLOADS.items():
        d = tempfile.mkdtemp(prefix="mmsynth")
        gtf, cfg, bam = make_inputs(w, d)
        cases = []
        for run in RUNS:
            tsv = os.path.join(d, "t.tsv")
            mf, mmf = os.path.join(d, "m.txt"), os.path.join(d, "mm.txt")
            for f in (mf, mmf):
                if os.path.exists(f):
                    os.remove(f)
            args = [mf if a == "$M" else (mmf if a == "$MM" else a) for a in run]
            rc, so, se = mo.run_ref("mmannot_patched", ["-a", gtf, "-r", bam, "-c", cfg, "-o", tsv] + args)
            cases.append(dict(args=run, rc=rc, table=open(tsv).read(), stderr=se.replace(d, "$D"),
                              mfile=open(mf).read() if "$M" in run else None, mmfile=open(mmf).read() if "$MM" in run else None))
            print("synth_e2e", name, run, rc, cases[-1]["table"].count("\n") - 1, "rows", file=sys.stderr)
        # two input files -> two columns (TableCount::addCounter), custom sample names (-n), stdout instead of -o
        if name == "tair10":
            w2 = dict(w, rseed=w["rseed"] + 100, n=w["n"] // 2)
            d2 = tempfile.mkdtemp(prefix="mmsynth2")
            _, _, bam2 = make_inputs(w2, d2)
            rc, so, se = mo.run_ref("mmannot_patched", ["-a", gtf, "-c", cfg, "-s", "F", "-r", bam, bam2, "-n", "first", "second"])
            out[name + "_two_files"] = dict(workload=w, workload2=w2, rc=rc, stdout=so, stderr=se.replace(d, "$D").replace(d2, "$E"))
            # -y random over several files: rand() runs on and the Counter's seen / chosenId / numberSeen maps (keyed by read name)
            # survive Counter::clear (mmannot.cpp:1742-1747), so later columns depend on the earlier files (the two BAMs share names)
            rnd = []
            for files, tag in (([bam, bam2], "ab"), ([bam2, bam, bam2], "bab")):
                rc, so, se = mo.run_ref("mmannot_patched", ["-a", gtf, "-c", cfg, "-s", "U", "-y", "random", "-r"] + files)
                rnd.append(dict(files=tag, rc=rc, stdout=so, stderr=se.replace(d, "$D").replace(d2, "$E")))
            out[name + "_two_files_random"] = dict(workload=w, workload2=w2, runs=rnd)
        out[name] = dict(workload=w, cases=cases)
    json.dump(out, open(os.path.join(GOLD, "cases.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
