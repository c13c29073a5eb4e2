"""Cross-rank merge of the per-GPU count tables.

Reads are sharded by read name (every hit of one name on one rank), so each rank finishes with its
own (element-set -> count) table and its own seven counters.  The merge is the only exchange step:
  1. all_gather of the table sizes, then of the (padded) keys  -> identical sorted key dictionary everywhere
  2. all_reduce(SUM, int64) of the dense count vector + the seven counters
Payload is a few KB: latency-bound, any algorithm; backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in
the CPU tests.  -y ratio is excluded: its double sums are order dependent (mmannot.cpp:1730) and are not
summed across shards.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np
import torch
import torch.distributed as dist


def _alone(group) -> bool:
    """No exchange to make: no process group, or a group of one (unless MM_BENCH_FORCE_DIST asks for the collectives anyway - the
    way to run the RCCL calls of this module on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not os.environ.get("MM_BENCH_FORCE_DIST")


def warm_up(device=None, group=None, p2p: bool = False) -> None:
    """One tiny call of every collective the merge uses (and, for the -y ratio chain, a send/recv round along the ranks):
    RCCL sets its channels and kernels up on first use, which would otherwise land inside a timed region."""
    if _alone(group):
        return
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    x = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.all_gather([torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)], x, group=group)
    dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    dist.broadcast(x, src=ws - 1, group=group)
    if dev.type != "cpu":
        dist.all_gather_into_tensor(torch.zeros(ws, dtype=torch.int64, device=dev), x, group=group)
    if p2p:
        if rank > 0:
            dist.recv(x, src=rank - 1, group=group)
        if rank < ws - 1:
            dist.send(x, dst=rank + 1, group=group)
    dist.barrier(group=group)


def merge_counts(masks: np.ndarray, sums: np.ndarray, stats: Dict[str, int], device=None, group=None) -> Tuple[np.ndarray, np.ndarray, Dict[str, int]]:
    """masks uint64[n], sums float64[n] (integral), stats dict -> merged (masks, sums, stats), identical on every rank."""
    if _alone(group):
        return masks, sums, stats
    ws = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    counts = np.rint(sums).astype(np.int64)
    if not np.array_equal(counts.astype(np.float64), sums):
        raise ValueError("merge_counts: non-integral sums (-y ratio is not mergeable by summation)")
    n = torch.tensor([len(masks)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(sizes, n, group=group)
    mx = max(1, int(max(int(s.item()) for s in sizes)))
    keys = torch.zeros(mx, dtype=torch.int64, device=dev)
    keys[:len(masks)] = torch.from_numpy(masks.view(np.int64).copy()).to(dev)
    allk = [torch.zeros(mx, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(allk, keys, group=group)
    cat = torch.cat([k[:int(s.item())] for k, s in zip(allk, sizes)]) if mx else keys[:0]
    dictionary = torch.unique(cat, sorted=True)
    dense = torch.zeros(len(dictionary) + 7, dtype=torch.int64, device=dev)
    if len(masks):
        pos = torch.searchsorted(dictionary, keys[:len(masks)])
        dense[pos] = torch.from_numpy(counts).to(dev)
    names = list(stats.keys())
    dense[len(dictionary):] = torch.tensor([stats[k] for k in names], dtype=torch.int64, device=dev)
    dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=group)
    out = dense.cpu().numpy()
    mk = dictionary.cpu().numpy().view(np.uint64)
    return mk, out[:len(mk)].astype(np.float64), {k: int(v) for k, v in zip(names, out[len(mk):])}


def merge_counts_device(counter, device, group=None, rehearsal: bool = False) -> Tuple[np.ndarray, np.ndarray, Dict[str, int]]:
    """The same merge with the tables staying on the devices.  Every rank's table has the same fixed size (Counter.TABLE_SLOTS sets and
    counts + the seven counters, 1 MB), so ONE all_gather moves everything - no sizes first, no dictionary, no all_reduce - and every
    rank adds the other ranks' rows into its own table itself (Counter.merge_counts_device: one kernel, integer sums, the same result
    everywhere), then reads its table back as it would alone.  `device`: the rank's GPU (backend nccl = RCCL over xGMI); with a CPU
    backend (gloo rehearsals) the buffer takes the detour through host memory that the collective then needs.
    Returns (masks, sums, stats) like merge_counts (rows in this rank's table order, not sorted).  The counter holds the merged job
    afterwards.  rehearsal=True: the same calls on empty tables (nothing is added to the counter) - first use of a collective sets up
    RCCL channels and loads kernels, which a caller may want out of a timed region."""
    if _alone(group):
        m, s_ = counter.getCounts()
        return m, s_, counter.stats()
    S = counter.TABLE_SLOTS
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.zeros(2 * S + 8, dtype=torch.int64, device=device)
    if not rehearsal:
        counter.export_counts_device(mine.data_ptr(), mine[S:].data_ptr(), mine[2 * S:].data_ptr())
    counter.sync()                                    # (the export ran on the context's stream, the collective runs on RCCL's)
    if dist.get_backend(group) != "nccl":             # (CPU backend: gloo rehearsals)
        src = mine.cpu()
        parts = [torch.empty_like(src) for _ in range(ws)]
        dist.all_gather(parts, src, group=group)
        allt = torch.cat(parts).to(device)
    else:
        allt = torch.empty(ws * (2 * S + 8), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allt, mine, group=group)
    torch.cuda.current_stream(device).synchronize()   # (the gathered tables are complete before the context's stream reads them)
    counter.merge_counts_device(allt.data_ptr(), ws, 2 * S + 8, rank)
    stats = counter.stats()                           # (synchronises the context's stream: allt may go)
    m, s_ = counter.getCounts()
    return m, s_, stats


def chain_ratio(counter, stats: Dict[str, int], device=None, group=None, n_chunks: int = 1):
    """-y ratio across ranks that hold consecutive blocks of the file (rank order = file order).  Every rank has already probed
    and sorted its batches with defer_ratio; the ordered double accumulation (mmannot.cpp:1730) is then chained: receive the
    rows of rank-1, seed, replay the parked batches, send the rows to rank+1; the last rank broadcasts the final table.
    n_chunks > 1 pipelines the chain by row chunks (rows do not depend on each other): for chunk c = 0..n_chunks-1 a rank receives
    the chunk's rows, replays its parked batches for those rows only (mm_ratio_resolve_chunk) and sends them on - the next rank
    works on chunk c while this one goes on with c+1, (ranks + n_chunks - 1) / n_chunks replays instead of `ranks`.  Measured
    (DESIGN.md section 5): with rows travelling through the host on every hop the fixed cost per hop outweighs the overlap, so
    the default stays 1 there.  With device tensors (backend nccl = RCCL) the table stays on the devices (mm_table_export_device /
    mm_table_import_device / mm_ratio_replay, no host hop) and chunking pays.  Returns (masks, sums, stats) identical on every rank."""
    from .api import row_chunk
    chain_ratio.last_hop_ms = None
    if _alone(group):
        counter.resolve_ratio()
        m, s = counter.getCounts()
        return m, s, stats
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")

    def send_rows(m, s, dst):
        dist.send(torch.tensor([len(m)], dtype=torch.int64, device=dev), dst=dst, group=group)
        if len(m):
            dist.send(torch.from_numpy(m.view(np.int64).copy()).to(dev), dst=dst, group=group)
            dist.send(torch.from_numpy(s.view(np.int64).copy()).to(dev), dst=dst, group=group)   # the bit patterns travel, not rounded values

    def recv_rows(src):
        n = torch.zeros(1, dtype=torch.int64, device=dev)
        dist.recv(n, src=src, group=group)
        k = int(n.item())
        if k == 0:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.float64)
        m = torch.zeros(k, dtype=torch.int64, device=dev); s = torch.zeros(k, dtype=torch.int64, device=dev)
        dist.recv(m, src=src, group=group); dist.recv(s, src=src, group=group)
        return m.cpu().numpy().view(np.uint64), s.cpu().numpy().view(np.float64)

    n_chunks = max(1, int(n_chunks))
    if dev.type == "cuda":
        # Round 3: the table never leaves the devices.  What travels is the whole table - 65,536 element sets and the bit patterns of
        # their sums, 1 MB, fixed size, so no count has to be exchanged first - from the context's stream into a tensor, over RCCL
        # (send / recv are ordered against that stream) and from the tensor into the next context's rows; nothing waits for the host
        # until the chain's last replay is enqueued.  Rows are independent of each other, so the chain is pipelined by row chunks: a
        # rank works on chunk c + 1 while the next one replays chunk c.
        S = counter.TABLE_SLOTS
        # (send / recv are ordered against torch's current stream.  A context created on that stream - bench.py's are - needs nothing
        # more; one with a stream of its own is synchronised by hand around every hand-over: correct, just not overlapped)
        foreign = getattr(counter, "stream", 0) != torch.cuda.current_stream(dev).cuda_stream
        bufs = [torch.zeros(2 * S, dtype=torch.int64, device=dev) for _ in range(2)]       # (a send may still be reading the previous chunk's)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_chunks)]
        for c in range(n_chunks):
            buf = bufs[c & 1]
            ev[c][0].record()
            if rank > 0:
                dist.recv(buf, src=rank - 1, group=group)
                if foreign:
                    torch.cuda.current_stream(dev).synchronize()
                counter.import_table_device(buf.data_ptr(), buf.data_ptr() + 8 * S, S, c, n_chunks)
            counter.ratio_replay(0, 1 << 62, c, n_chunks)
            if rank < ws - 1:
                counter.export_table_device(buf.data_ptr(), buf.data_ptr() + 8 * S)
                if foreign:
                    counter.sync()
                dist.send(buf, dst=rank + 1, group=group)
            ev[c][1].record()
        counter.ratio_release()
        chain_ratio.last_hop_ms = [a.elapsed_time(b) for a, b in ev]      # per chunk on this rank: wait for the predecessor + replay + hand-over
    else:
        for c in range(n_chunks):
            if rank > 0:
                m, s = recv_rows(rank - 1)
                counter.set_row_sums(m, s)
            counter.resolve_ratio_chunk(c, n_chunks)
            if rank < ws - 1:
                masks, sums = counter.getCounts()
                sel = row_chunk(masks, n_chunks) == c
                send_rows(np.ascontiguousarray(masks[sel]), np.ascontiguousarray(sums[sel]), rank + 1)
    masks, sums = counter.getCounts()
    # final table from the last rank to everybody
    n = torch.tensor([len(masks)], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=ws - 1, group=group)
    k = int(n.item())
    m = torch.zeros(k, dtype=torch.int64, device=dev); s = torch.zeros(k, dtype=torch.int64, device=dev)
    if rank == ws - 1 and k:
        m.copy_(torch.from_numpy(masks.view(np.int64).copy())); s.copy_(torch.from_numpy(sums.view(np.int64).copy()))
    if k:
        dist.broadcast(m, src=ws - 1, group=group); dist.broadcast(s, src=ws - 1, group=group)
    names = list(stats.keys())
    st = torch.tensor([stats[x] for x in names], dtype=torch.int64, device=dev)
    dist.all_reduce(st, op=dist.ReduceOp.SUM, group=group)
    return m.cpu().numpy().view(np.uint64), s.cpu().numpy().view(np.float64), {x: int(v) for x, v in zip(names, st.cpu().tolist())}
