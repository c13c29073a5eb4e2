#!/usr/bin/env python3
"""Host throughput of mm_partition (one decoded batch dealt out over N devices by read name): tuples / second.

    python tools/bench_partition.py [--log2 24] [--parts 8] [--threads 16]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=24)
    ap.add_argument("--parts", type=int, nargs="+", default=[2, 8])
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 4, 8, 16])
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    n = 1 << args.log2
    rng = np.random.default_rng(1)
    for th in args.threads:
        os.environ["MM_PARTITION_THREADS"] = str(th)
        # (the thread count is read once per process: one child per setting)
        code = f"""
import ctypes as C, numpy as np, time, sys
sys.path.insert(0, {os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')!r})
import mmannot_amd.api as api
from mmannot_amd._lib import lib
n = {n}
rng = np.random.default_rng(1)
aln = np.empty(n, dtype=api.ALN_DTYPE)
aln["start"] = rng.integers(1, 1 << 28, n, dtype=np.int64); aln["end"] = aln["start"] + 99
aln["qid"] = (np.arange(n, dtype=np.int64) // 10 + rng.integers(0, 6000, n)).astype(np.uint32)
aln["tsn"] = (rng.integers(1, 20, n).astype(np.uint32) << 1) | (np.uint32(3) << 12)
L = lib()
for N in {args.parts!r}:
    outs = [np.zeros(n, dtype=api.ALN_DTYPE) for _ in range(N)]
    bigs = [np.zeros(64, dtype=api.NH_BIG_DTYPE) for _ in range(N)]
    out_n = (C.c_size_t * N)(); out_nb = (C.c_size_t * N)(); out_mx = (C.c_uint32 * N)(); out_lo = (C.c_uint64 * N)()
    po = (C.c_void_p * N)(*[o.ctypes.data for o in outs]); pb = (C.c_void_p * N)(*[b.ctypes.data for b in bigs])
    best = 1e9
    for r in range({args.reps} + 1):
        t0 = time.perf_counter()
        rc = L.mm_partition(aln.ctypes.data, n, None, 0, N, po, out_n, pb, 64, out_nb, out_mx, out_lo)
        dt = time.perf_counter() - t0
        assert rc == 0
        if r: best = min(best, dt)
    print(f"threads {th}  parts {{N}}: {{n / best / 1e9:.2f}} G tuples/s ({{best * 1e3:.2f}} ms per 2^{args.log2})", flush=True)
"""
        import subprocess
        subprocess.run([sys.executable, "-c", code.replace("{th}", str(th))], check=True, env=os.environ)


if __name__ == "__main__":
    main()
