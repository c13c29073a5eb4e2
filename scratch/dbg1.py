import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import numpy as np, synth_ref as sr, mmannot_amd as mm
from oracle import mm_oracle as mo
seed, l, s, y = 1, 25.0, 'U', 'default'
rng = np.random.default_rng(1000 * seed + int(abs(l) * 10))
desc = sr.random_config(rng, n_lines=int(rng.integers(3, 12)), max_per_line=int(rng.integers(1, 5)))
ix = sr.random_index(rng, desc.E, nchr=4, n=3000, maxpos=150000)
reads = sr.random_reads(rng, ix, n=20000, nchr=4, maxpos=150000, nh_max=7)
ref = mo.run(desc, ix, reads, overlap=l, strandedness=s, strategy=y, want_hit_mask=True)
cfg = mm.Config.from_json(desc.to_json())
for batches, shift in ((1,0),(5,8),(3,14),(1,0)):
    index = mm.IntervalIndex(cfg, ix["chr"], ix["start"], ix["end"], ix["type"], ix["strand"], 4, bin_shift=shift)
    ctr = mm.Counter(index, overlap=l, strandedness=s, strategy=y, keep_hit_masks=True)
    aln = mm.pack_alignments(reads["tid"], reads["start"], reads["end"], reads["strand"], reads["qid"])
    n=len(aln); cuts=np.linspace(0,n,batches+1).astype(int)
    for a,b in zip(cuts[:-1],cuts[1:]):
        ctr.add_batch(aln[a:b], reads["nh"][a:b], int(reads["qid"].max())+1); ctr.sync()
    ctr.flush()
    masks,sums = ctr.getCounts()
    got = {}
    dup = 0
    for m_,s_ in zip(masks.tolist(), sums.tolist()):
        if m_ in got: dup += 1
        got[m_] = got.get(m_,0)+s_
    want = dict(zip(ref.row_mask.tolist(), ref.row_sum.tolist()))
    diff = {k:(got.get(k),want.get(k)) for k in set(got)|set(want) if got.get(k)!=want.get(k)}
    print('batches',batches,'shift',shift,'rows got',len(masks),'want',len(want),'dups',dup,'ndiff',len(diff),'sum got',sum(got.values()),'want',sum(want.values()), 'stats eq', ctr.stats()==ref.stats)
    for k,v in list(diff.items())[:8]: print('   ', hex(k), v)
    ctr.close(); index.close()
