#!/usr/bin/env python3
"""Host-only: records / second of the reads decoder (mm_reads.cpp) on a synthetic BGZF BAM - inflate + framing + parsing on the
pipe's threads, read-name dictionary on the caller's thread.  No GPU needed.

    python tools/bench_decoder.py [--reads-log2 23] [--seq] [--scatter 8] [--threads N]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads-log2", type=int, default=23)
    ap.add_argument("--seq", action="store_true", help="records carry 100 bp of sequence + qualities like a real BAM")
    ap.add_argument("--scatter", type=int, default=8, help="hits of a read spread over 2^scatter records (0: adjacent, an aligner's own order)")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    import synth_ref as sr
    import mmannot_amd.api as api
    from mmannot_amd import synth
    from oracle import mm_oracle as mo
    an = synth.synth_annotation("tair10", seed=3, scale=0.2, keep_genes=True)
    d = tempfile.mkdtemp(prefix="mmdec")
    gtf = os.path.join(d, "a.gtf"); open(gtf, "w").write(synth.gtf_text(an))
    cfg = os.path.join(d, "c.cfg"); open(cfg, "w").write(synth.config_text("tair10"))
    n = 1 << a.reads_log2
    t = sr.synth_tuples(7, 0, n, 100, 6, an.chr_len, a.scatter)
    body = mo.bam_records_fixed(t["tid"], t["start"] - 1, np.where(t["strand"] == 1, 0, 16).astype(np.uint16), t["qid"], np.full(n, 100), t["nh"],
                                **({"seq_len": 100} if a.seq else {}))
    bam = os.path.join(d, "r.bam")
    mo.write_bam_bgzf(bam, [(c, int(l)) for c, l in zip(an.chr_names, an.chr_len)], body, level=1)
    ann = api.Annotation(cfg, gtf)
    import ctypes as C
    from mmannot_amd._lib import lib
    aln = np.zeros(1 << 22, dtype=api.ALN_DTYPE); big = np.zeros(1 << 16, dtype=api.NH_BIG_DTYPE)
    for label, env in (("framed + parsed by the pipe's threads", {}), ("sequential general parser (MM_READER_GENERAL=1)", {"MM_READER_GENERAL": "1"})):
        os.environ.pop("MM_READER_GENERAL", None)
        os.environ.update(env)
        if a.threads:
            os.environ["MM_READER_THREADS"] = str(a.threads)
        for rep in range(3):
            dec = api.ReadsDecoder(bam, ann)
            t0 = time.perf_counter()
            tot = 0
            nn, nb, mx = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
            while True:
                rc = lib().mm_reader_next(dec._h, aln.ctypes.data, len(aln), C.byref(nn), big.ctypes.data, len(big), C.byref(nb), C.byref(mx))
                assert rc == 0
                if nn.value == 0:
                    break
                tot += nn.value
            dt = time.perf_counter() - t0
            print(f"{label}, run {rep}: {tot} records in {dt:.2f} s = {tot / dt / 1e6:.1f} M records/s ({os.path.getsize(bam) / dt / 1e6:.0f} MB/s of BGZF), "
                  f"{dec.qid_bound()} names", flush=True)
            dec.close()


if __name__ == "__main__":
    main()
