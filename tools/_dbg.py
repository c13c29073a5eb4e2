import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth_ref as sr
from oracle import mm_oracle as oracle
import mmannot_amd as mm
from test_gpu_parity import run_gpu, to_cfg
seed, l, s, y = 1, 0.0, "U", "default"
rng = np.random.default_rng(1000 * seed + int(abs(l) * 10))
desc = sr.random_config(rng, n_lines=int(rng.integers(3, 12)), max_per_line=int(rng.integers(1, 5)))
ix = sr.random_index(rng, desc.E, nchr=4, n=3000, maxpos=150000, skip_chr=None)
reads = sr.random_reads(rng, ix, n=20000, nchr=4, maxpos=150000, nh_max=7)
ref = oracle.run(desc, ix, reads, overlap=l, strandedness=s, strategy=y, want_hit_mask=True)
cfg = to_cfg(mm, desc)
for env in ({}, {"MMANNOT_AMD_NO_EXT": "1"}, {"MMANNOT_AMD_NO_FAST": "1"}):
    for k in ("MMANNOT_AMD_NO_EXT", "MMANNOT_AMD_NO_FAST"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for nread in (20000, 8192, 3000, 256, 64):
        rd = {k: v[:nread] for k, v in reads.items()}
        rf = oracle.run(desc, ix, rd, overlap=l, strandedness=s, strategy="unique" if False else y, want_hit_mask=True)
        r = run_gpu(mm, cfg, ix, rd, overlap=l, strandedness=s, strategy=y, batches=1, bin_shift=0, keep=False)
        bad = {k: (r["stats"][k], rf.stats[k]) for k in rf.stats if r["stats"][k] != rf.stats[k]}
        print(env, nread, "OK" if not bad and r["table"] == oracle.format_table(desc, rf.row_mask, rf.row_count, "test_dataset") else bad, flush=True)
