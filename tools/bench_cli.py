#!/usr/bin/env python3
"""End-to-end wall time of the `mmannot-amd` binary vs the compiled reference on the same files (informational: BAM decode and
the read-name dictionary dominate both; the hot path itself is what bench.py measures).

    python tools/bench_cli.py [--reads-log2 22] [--shape tair10|hs38]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads-log2", type=int, default=22)
    ap.add_argument("--shape", default="tair10")
    ap.add_argument("--seq", action="store_true", help="records carry 100 bp of sequence+quality like a real BAM")
    a = ap.parse_args()
    import synth_ref as sr
    from mmannot_amd import synth
    from oracle import mm_oracle as mo
    an = synth.synth_annotation(a.shape, seed=3, scale=1.0 if a.shape == "tair10" else 0.5, keep_genes=True)
    d = tempfile.mkdtemp(prefix="mmcli")
    gtf = os.path.join(d, "a.gtf"); open(gtf, "w").write(synth.gtf_text(an))
    cfg = os.path.join(d, "c.cfg"); open(cfg, "w").write(synth.config_text(a.shape))
    n = 1 << a.reads_log2
    t = sr.synth_tuples(7, 0, n, 100, 6, an.chr_len, 8)
    body = mo.bam_records_fixed(t["tid"], t["start"] - 1, np.where(t["strand"] == 1, 0, 16).astype(np.uint16), t["qid"], np.full(n, 100), t["nh"],
                                **({"seq_len": 100} if a.seq else {}))
    bam = os.path.join(d, "r.bam")
    mo.write_bam_bgzf(bam, [(c, int(l)) for c, l in zip(an.chr_names, an.chr_len)], body, level=1)
    res = {}
    for name, exe in (("reference (oracle/_ref/mmannot_patched)", os.path.join(ROOT, "oracle", "_ref", "mmannot_patched")),
                      ("mmannot-amd", os.path.join(ROOT, "mmannot_amd", "mmannot-amd"))):
        if not os.path.exists(exe):
            continue
        out = os.path.join(d, name.split()[0] + ".tsv")
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-a", gtf, "-c", cfg, "-r", bam, "-s", "F", "-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, MM_CLI_TIMING="1"))
        dt = time.perf_counter() - t0
        for line in p.stderr.decode().splitlines():
            if line.startswith("[mmannot-amd timing]") or line.startswith("[reader]"):
                print("   ", line)
        res[name] = (dt, open(out).read() if p.returncode == 0 else p.stderr.decode()[-300:])
        print(f"{name}: {dt:.2f} s wall, {n / dt / 1e6:.2f} M records/s end to end (rc {p.returncode})")
    tabs = [v[1] for v in res.values()]
    if len(tabs) == 2:
        print("tables identical:", tabs[0] == tabs[1])


if __name__ == "__main__":
    main()
