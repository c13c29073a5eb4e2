#!/usr/bin/env python3
"""Soak of the two probes against the oracle (GPU box): many seeded random configurations, annotations (uniform and clustered - bins of
every kind the pipelined probe distinguishes) and read sets, every combination run with the pipelined probe forced
(MMANNOT_AMD_FAST=1), with the walk forced (MMANNOT_AMD_NO_FAST=1) and with the context's own choice; table, counters and - under
-y ratio - the sums' bit patterns must equal the oracle's (oracle/ is the checker here, as in tests/).

    python tools/soak_probe.py [n_seeds=60] [first_seed=1000]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_ref as sr                                   # noqa: E402
from oracle import mm_oracle as oracle                   # noqa: E402
import mmannot_amd as mm                                  # noqa: E402
from test_gpu_parity import run_gpu, to_cfg               # noqa: E402
from test_gpu_paths import _clustered_index               # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad = 0
    t0 = time.time()
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        desc = sr.random_config(rng, n_lines=int(rng.integers(2, 12)), max_per_line=int(rng.integers(1, 5)))
        if seed % 2:
            ix = _clustered_index(rng, desc, nchr=int(rng.integers(1, 5)), maxpos=int(rng.integers(120_000, 600_000)))
            maxpos = int(ix["end"].max()) + 5000
        else:
            maxpos = int(rng.integers(50_000, 2_000_000))
            ix = sr.random_index(rng, desc.E, nchr=int(rng.integers(1, 6)), n=int(rng.integers(200, 8000)), maxpos=maxpos, skip_chr=None)
        nchr = ix["n_chromosomes"]
        reads = sr.random_reads(rng, ix, n=int(rng.integers(3000, 40_000)), nchr=nchr, maxpos=maxpos, nh_max=int(rng.integers(1, 9)))
        l = float(rng.choice([-1.0, -1.0, 0.0, 0.3, 0.83, 1.0, 20.0, 60.0]))
        s = str(rng.choice(["U", "F", "R"]))
        y = str(rng.choice(["default", "default", "unique", "ratio"]))
        ref = oracle.run(desc, ix, reads, overlap=l, strandedness=s, strategy=y, want_hit_mask=False)
        want = oracle.format_table(desc, ref.row_mask, ref.row_count, "test_dataset")
        cfg = to_cfg(mm, desc)
        for env in ({"MMANNOT_AMD_FAST": "1"}, {"MMANNOT_AMD_NO_FAST": "1"}, {}):
            for k in ("MMANNOT_AMD_FAST", "MMANNOT_AMD_NO_FAST"):
                os.environ.pop(k, None)
            os.environ.update(env)
            for nb, shift in ((1, 0), (3, int(rng.choice([0, 7, 9, 11])))):
                r = run_gpu(mm, cfg, ix, reads, overlap=l, strandedness=s, strategy=y, batches=nb, bin_shift=shift, keep=False)
                ok = r["stats"] == ref.stats and r["table"] == want
                if ok and y == "ratio":
                    ok = dict(zip(r["masks"].tolist(), r["sums"].view(np.uint64).tolist())) == dict(zip(ref.row_mask.tolist(), ref.row_sum.view(np.uint64).tolist()))
                if not ok:
                    bad += 1
                    print(f"MISMATCH seed {seed} env {env} batches {nb} shift {shift} -l {l} -s {s} -y {y}", flush=True)
        if (seed - first) % 10 == 9:
            print(f"{seed - first + 1} seeds, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"soak: {n_seeds} seeds x 3 probe settings x 2 batchings, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
