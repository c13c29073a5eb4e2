#!/usr/bin/env python3
"""Fixed costs of a run (configuration + GTF -> intervals -> device index) on a one-read BAM: `mmannot-amd` vs the compiled reference.

    python tools/bench_fixed.py [--shape hs38|tair10|flybase6] [--scale 1.5]
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
    ap.add_argument("--shape", default="hs38")
    ap.add_argument("--scale", type=float, default=1.5)
    a = ap.parse_args()
    from mmannot_amd import synth
    from oracle import mm_oracle as mo
    an = synth.synth_annotation(a.shape, seed=3, scale=a.scale, keep_genes=True)
    d = tempfile.mkdtemp(prefix="mmfixed")
    gtf = os.path.join(d, "a.gtf"); open(gtf, "w").write(synth.gtf_text(an))
    cfg = os.path.join(d, "c.cfg"); open(cfg, "w").write(synth.config_text(a.shape))
    bam = os.path.join(d, "one.bam")
    mo.write_bam_bytes(bam, [(c, int(l)) for c, l in zip(an.chr_names, an.chr_len)],
                       mo.bam_records_fixed(np.array([0]), np.array([100]), np.array([0], dtype=np.uint16), np.array([0]), np.array([100]), None), level=1)
    print(f"GTF: {os.path.getsize(gtf) / 1e6:.0f} MB, {len(an.start)} intervals")
    for name, exe in (("reference", os.path.join(ROOT, "oracle", "_ref", "mmannot_patched")), ("mmannot-amd", os.path.join(ROOT, "mmannot_amd", "mmannot-amd"))):
        if not os.path.exists(exe):
            continue
        for rep in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([exe, "-a", gtf, "-c", cfg, "-r", bam, "-s", "F", "-o", os.path.join(d, "o.tsv")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, MM_CLI_TIMING="1"))
            dt = time.perf_counter() - t0
            extra = [l for l in p.stderr.decode().splitlines() if l.startswith("[mmannot-amd timing]")]
            print(f"{name}: {dt:.2f} s (rc {p.returncode})", extra[0] if extra else "")


if __name__ == "__main__":
    main()
