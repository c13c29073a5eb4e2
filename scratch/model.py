import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, synth_ref as sr
from oracle import mm_oracle as mo
from collections import defaultdict
seed, l, s, y = 1, 25.0, 'U', 'default'
rng = np.random.default_rng(1000 * seed + int(abs(l) * 10))
desc = sr.random_config(rng, n_lines=int(rng.integers(3, 12)), max_per_line=int(rng.integers(1, 5)))
ix = sr.random_index(rng, desc.E, nchr=4, n=3000, maxpos=150000)
reads = sr.random_reads(rng, ix, n=20000, nchr=4, maxpos=150000, nh_max=7)
ref = mo.run(desc, ix, reads, overlap=l, strandedness=s, strategy=y, want_hit_mask=True)
# groups by the oracle's hit masks (trusted: equal to GPU's)
hm = ref.hit_mask
pending = {}
count = defaultdict(float)
for i in range(len(hm)):
    nh = int(reads['nh'][i]); q = int(reads['qid'][i]); m = int(hm[i])
    if nh > 1:
        if q not in pending: pending[q] = [nh-1, m]
        else:
            g = pending[q]; g[0] -= 1; g[1] |= m
            if g[0] == 0:
                if g[1]: count[g[1]] += 1
                del pending[q]
    else:
        if m: count[m] += 1
for q,g in pending.items():
    if g[1]: count[g[1]] += 1
want = dict(zip(ref.row_mask.tolist(), ref.row_sum.tolist()))
diff = {k:(count.get(k),want.get(k)) for k in set(count)|set(want) if count.get(k)!=want.get(k)}
print('model vs oracle diffs', {hex(k):v for k,v in diff.items()}, 'nrows', len(count))
