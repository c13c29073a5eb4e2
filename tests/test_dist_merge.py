"""CPU, world_size 2, gloo: the cross-rank merge of per-shard count tables (mmannot_amd/dist.py).
The reads of a seeded synthetic input are sharded by read name; each rank counts its shard with the CPU
oracle (the GPU is not available here — this test covers only the exchange step) and the merged table
must equal the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth_ref as sr
    from oracle import mm_oracle as mo
    from mmannot_amd.dist import merge_counts
    rng = np.random.default_rng(11)
    desc = sr.random_config(rng, 6, 3)
    ix = sr.random_index(rng, desc.E, nchr=3, n=1500, maxpos=80000)
    reads = sr.random_reads(rng, ix, n=12000, nchr=3, maxpos=80000, nh_max=6)
    full = mo.run(desc, ix, reads, overlap=-1.0, strandedness="F", strategy="default")
    mine = reads["qid"] % world == rank                      # shard by read name: groups are never split
    shard = {k: v[mine] for k, v in reads.items()}
    part = mo.run(desc, ix, shard, overlap=-1.0, strandedness="F", strategy="default")
    m, s, st = merge_counts(part.row_mask, part.row_sum, part.stats)
    order = np.argsort(m)
    fo = np.argsort(full.row_mask)
    ok = (np.array_equal(m[order], full.row_mask[fo]) and np.array_equal(s[order], full.row_sum[fo]) and st == full.stats)
    # an empty shard must merge too
    if rank == 1:
        e = merge_counts(np.zeros(0, dtype=np.uint64), np.zeros(0), {k: 0 for k in part.stats})
    else:
        e = merge_counts(part.row_mask, part.row_sum, part.stats)
    ok = ok and np.array_equal(np.sort(e[0]), np.sort(part.row_mask) if rank == 0 else np.sort(e[0]))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_merge_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
