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

from typing import Dict, Tuple

import numpy as np
import torch
import torch.distributed as dist


def merge_counts(masks: np.ndarray, sums: np.ndarray, stats: Dict[str, int], device=None, group=None) -> Tuple[np.ndarray, np.ndarray, Dict[str, int]]:
    """masks uint64[n], sums float64[n] (integral), stats dict -> merged (masks, sums, stats), identical on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
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
