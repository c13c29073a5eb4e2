#!/usr/bin/env python3
"""bench.py — annotated alignments / second of the overlap-and-count hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--batch-log2 B] [--scaling weak|strong]

One process per GPU (the driver launches N>1 through torch.distributed.run; backend "nccl" = RCCL).
A step = one batch of 2^B synthetic 16-byte alignment tuples pushed through mm_annotate_batch: probe -> strand / -l filter ->
priority selection -> multi-hit merge -> per-set counts.  Two legs run in the same invocation, on the same batches:

  leg "resident"  -> `value` (and `kernel_only`, the same figure with the step's own roofline): the tuple batches are resident in HBM
      when the timed region starts; the region holds all kernels, the end-of-file flush and the cross-rank table merge.  This is the
      leg the `roofline` block of the dominant kernel is measured on, with HIP events around its launches.  (Rounds 1 and 2 quoted the
      host-fed rate as `value`, after SURVEY.md 8d; the task statement wants the HBM-resident one, the PCIe-inclusive one beside it.)
  leg "from host" -> `from_host`: the same batches start in PINNED HOST memory; the timed region also holds the hipMemcpyAsync of
      every batch (the library's copy stream, overlapped with the kernels of the previous batch).  `h2d` reports the link rate -
      this leg runs at the PCIe link's pace (90 % of Gen5 x16), an order of magnitude below the kernels'.

Prints ONE JSON line on rank 0.

Workloads (BASELINE.json configs):
  c3 (default): 100 bp single-end reads, read groups with NH = 1..20 (exactly NH hits each, scattered over
      windows of 2^16 tuples), HS38-scale synthetic annotation (~1.27 M intervals, 22 Order elements),
      -y default -s F -l -1.  This is the configuration the north-star target is quoted on.
  c2: 50 bp reads, NH = 1, TAIR10-scale annotation (~0.55 M intervals, 14 elements).
  c4: 100 bp read pairs (mates share a name, opposite strands), Flybase6-shaped annotation with introns and
      vicinity (-d/-D 1000), -l 50, under -s U: the reference's parser rejects -s FR, so -s U is
      the only setting of this configuration that has an oracle.
  c5: NH = 1..100 groups, -y ratio, HS38-scale annotation.
--scaling weak (default): every rank runs K batches of 2^B tuples of a stream of its own.  strong: ONE seeded stream of K batches of
2^B tuples is dealt out over the ranks the way the product shards an input (mm_partition: by read name, qid mod N, names renumbered
qid / N, file order kept - every hit of a read on one GPU, which the multi-hit countdown needs, mmannot.cpp:1669-1701; -y ratio:
consecutive blocks of the stream, its double sums being file-order dependent, mmannot.cpp:1730), so the job's table - and its
`table_digest` - is the same for every N.

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (python -m torch.distributed.run as a
CHILD process, before anything in this process touches the GPU) and exits with their status; under a launcher, a WORLD_SIZE that
disagrees with --gpus is an error.  With more than one rank, rank 0 re-runs every rank's share on its own GPU afterwards (untimed)
and checks the merged table and counters against it (`merge_check`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_PROBE = 32.0   # algorithmic bytes / alignment of the probe kernel: 16 B tuple + 16 B index
B_ALG_H2D = 16.0     # bytes / alignment that have to cross the link (SURVEY.md 8d "stream in")
PCIE_PEAK_GBS = 63.0  # PCIe Gen5 x16, one direction: 32 GT/s x 16 lanes x 128/130

# BASELINE.json configs -> (annotation shape, scale, read length, NH max, scatter window log2, mate distance, strategy, -l, -s)
WORKLOADS = {
    "c2": dict(shape="tair10", scale=1.0, read_len=50, nh_max=1, scatter=0, pair=0, strategy="default", overlap=-1.0, strand="F",
               name="c2: synthetic 50bp SE reads, NH=1, TAIR10-scale annotation, -y default -s F -l -1"),
    "c3": dict(shape="hs38", scale=1.5, read_len=100, nh_max=20, scatter=16, pair=0, strategy="default", overlap=-1.0, strand="F",
               name="c3: synthetic 100bp SE reads, NH=1..20 groups, HS38-scale annotation, -y default -s F -l -1"),
    "c4": dict(shape="flybase6", scale=1.0, read_len=100, nh_max=1, scatter=0, pair=250, strategy="default", overlap=50.0, strand="U",
               name="c4: synthetic 2x100bp read pairs, Flybase6-shaped annotation with introns + vicinity 1000, -y default -s U -l 50"),
    "c5": dict(shape="hs38", scale=1.5, read_len=100, nh_max=100, scatter=16, pair=0, strategy="ratio", overlap=-1.0, strand="F",
               name="c5: synthetic 100bp SE reads, NH=1..100 groups, HS38-scale annotation, -y ratio -s F -l -1"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30, help="timed batches (default 30 x 2^24 = 503 M alignments: BASELINE.json's configs[2] on one GPU)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-log2", type=int, default=24)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--bin-shift", type=int, default=0, help="log2 of the index bin width (0 = the library's choice)")
    ap.add_argument("--scatter-log2", type=int, default=-1, help="informational: window over which a read's hits are spread (0 = hits adjacent, "
                    "an aligner's own output order; default: the workload's)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--annotation-scale", type=float, default=0.0, help="scale sensitivity: multiply the workload's gene counts by this instead of its own "
                    "factor (c3: 1.5 -> 1.27 M intervals; 3.5 -> ~3 M, BASELINE.json's '~3M features')")
    ap.add_argument("--scaffolds", type=int, default=0, help="scale sensitivity: that many 60-kb scaffolds behind the chromosomes (> 256 reference names)")
    ap.add_argument("--probe-events", type=int, default=1, help="measurement aid: 0 = no HIP events around the probe launches of the timed legs "
                    "(the line of record needs them: roofline.avg_launch_ms)")
    ap.add_argument("--legs", default="both", choices=("both", "host", "resident"), help="profiling aid: run only one leg (the JSON line of record needs both)")
    ap.add_argument("--tuple-bytes", type=int, default=16, choices=(16, 12),
                    help="bytes per tuple on the link in the from-host leg (`from_host`): 16 = mm_aln (SURVEY.md 8d); "
                         "12 = the packed link format (mm_pack12 where the tuples are produced, expanded on the device).  With 16 the packed "
                         "format is still measured, in a leg of its own, and reported as `from_host_packed12`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log2", type=int, default=24)
    args = ap.parse_args()

    # N > 1 asked for plainly (no launcher): start the N ranks as a child process - this process has not touched the GPU (and will not)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
                         f"launch with --nproc-per-node {args.gpus} (or run `python bench.py --gpus {args.gpus}` without a launcher)\n")
        sys.exit(2)

    import torch
    import torch.distributed as dist
    import mmannot_amd as mm
    from mmannot_amd import _lib, synth
    from mmannot_amd.dist import chain_ratio, merge_counts_device, warm_up

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback exists)"
    # MM_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("MM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    numa = bind_to_gpu_numa_node(local_rank) if world > 1 else None
    # MM_BENCH_FORCE_DIST=1: the multi-rank code path (process group, warm-up, merge collectives, chain) with whatever world size the
    # launcher gave, also 1 - so that the RCCL backend can be exercised on a one-GPU box
    multi = world > 1 or bool(os.environ.get("MM_BENCH_FORCE_DIST"))
    if multi:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    n_joined = dist.get_world_size() if multi else 1          # ranks that really joined the group: what `n_gpus` reports
    assert n_joined == args.gpus, f"{n_joined} ranks joined the process group, --gpus says {args.gpus}"

    # ---------------- annotation (same on every rank) -> device index
    wl = dict(WORKLOADS[args.workload])
    if args.scatter_log2 >= 0:
        wl["scatter"] = args.scatter_log2
        wl["name"] += f" [hits of a read spread over 2^{args.scatter_log2} tuples]"
    want_cpu = (not args.no_cpu_baseline) and world == 1 and rank == 0
    if args.annotation_scale > 0:
        wl["scale"] = args.annotation_scale
        wl["name"] += f" [annotation scale {args.annotation_scale:g}]"
    if args.scaffolds:
        wl["name"] += f" [+ {args.scaffolds} scaffolds]"
    an = synth.synth_annotation(wl["shape"], seed=args.seed, scale=wl["scale"], keep_genes=want_cpu, scaffolds=args.scaffolds)
    read_len, nh_max, scatter, pair, strategy, wname = wl["read_len"], wl["nh_max"], wl["scatter"], wl["pair"], wl["strategy"], wl["name"]
    cfg = mm.Config.from_json(an.config)
    index = mm.IntervalIndex(cfg, an.chr, an.start, an.end, an.type, an.strand, an.n_chromosomes, device=local_rank, bin_shift=args.bin_shift)
    info = index.info()
    stream = torch.cuda.current_stream().cuda_stream
    ctr = mm.Counter(index, overlap=wl["overlap"], strandedness=wl["strand"], strategy=strategy, stream=stream,
                     defer_ratio=(strategy == "ratio" and multi))

    # ---------------- inputs: W+K batches per rank
    # weak: a stream of this rank's own (seed + rank), rank-local read-name ids.
    # strong: ONE stream (the job), dealt out as the product does it - default / unique: by read name, qid mod N, names renumbered
    #   qid / N, file order kept (mm_partition); ratio: rank r holds the r-th consecutive block of the stream.
    Bg = 1 << args.batch_log2                      # tuples per batch of a stream
    nb = args.warmup + args.steps
    strong = args.scaling == "strong"
    by_name = strong and strategy != "ratio"
    chr_len = np.ascontiguousarray(an.chr_len, dtype=np.int64)
    single = nh_max == 1
    T_grp = nh_max * (nh_max + 1) // 2
    W_sc = 1 << scatter
    job_seed = args.seed * 1000003

    # (the device generator knows 64 reference sequences: with --scaffolds the reads land on the chromosomes and the first scaffolds - the
    # index, the chromosome table in global memory and the general probe instance that > 256 names select are what that run is about)
    gen_len = np.ascontiguousarray(chr_len[:64])

    def gen(seed, first, n):
        a = torch.empty((n, 4), dtype=torch.int32, device=dev)
        p = _lib.SynthParams(seed, first, read_len, nh_max, len(gen_len), gen_len.ctypes.data, scatter, pair)
        _lib.check(_lib.lib().mm_synth_fill(a.data_ptr(), n, C.byref(p), stream))
        return a

    # read-name ids of a stream cut into batches of `bsz` tuples: the decoder of a real input knows the smallest id among a batch's
    # multi-hit tuples and the size of its dictionary; here both follow from the generator.  The hits of a read are spread over windows
    # of 2^scatter tuples: inside the batch (the default), or - a window wider than a batch, e.g. --scatter-log2 28 - over many
    # batches, which is what a coordinate-sorted file looks like (every batch then touches the names of a whole window: a wide key
    # range, several partition passes).
    def stream_qb(n_tuples):      # exclusive bound of the read-name ids of a stream of n_tuples
        if nh_max <= 1:
            return n_tuples + 2
        last_l = -((-n_tuples) // W_sc) * W_sc
        return (last_l // T_grp + 2) * nh_max + 2

    def stream_qlo(first):        # no multi-hit tuple at or behind stream position `first` has an id below this
        return 0 if nh_max <= 1 else ((first // W_sc) * W_sc // T_grp) * nh_max

    def stream_qbound(end, qb_all):   # exclusive bound of the ids seen in the stream's first `end` tuples
        if nh_max <= 1:
            return qb_all
        hi = -((-end) // W_sc) * W_sc
        return min(qb_all, (hi // T_grp + 2) * nh_max + 2)

    def ratio_block(r, s_):       # strong, -y ratio: rank r's s_-th batch of the job's nb * Bg tuples = (first tuple, number of tuples).
        tot = nb * Bg             # Rank r holds the r-th of `world` consecutive blocks (the remainder of the division spread over the
        lo, hi = tot * r // world, tot * (r + 1) // world          # blocks, nothing dropped), each cut into nb batches
        return lo + (hi - lo) * s_ // nb, (hi - lo) * (s_ + 1) // nb - (hi - lo) * s_ // nb

    alns, qlo_of, qbound_of = [], [], []
    if not strong:
        B = Bg
        qb = stream_qb(nb * B)
        for s_ in range(nb):
            alns.append(gen(job_seed + rank, s_ * B, B))
            qlo_of.append(stream_qlo(s_ * B)); qbound_of.append(stream_qbound((s_ + 1) * B, qb))
    elif by_name:
        qb_job = stream_qb(nb * Bg)
        qb = qb_job // world + 2
        for s_ in range(nb):
            g = gen(job_seed, s_ * Bg, Bg)
            q = g[:, 3].to(torch.int64) & 0xFFFFFFFF
            mine = g[(q % world) == rank].contiguous()
            mine[:, 3] = ((mine[:, 3].to(torch.int64) & 0xFFFFFFFF) // world).to(torch.int32)
            alns.append(mine)
            qlo_of.append(stream_qlo(s_ * Bg) // world); qbound_of.append(min(qb, stream_qbound((s_ + 1) * Bg, qb_job) // world + 2))
            del g, q
        B = max(1, Bg // world)
    else:                        # strong, -y ratio: consecutive blocks of the same nb * Bg tuples for every N
        B = max(1, Bg // world)
        qb = stream_qb(nb * Bg)
        for s_ in range(nb):
            first, cnt = ratio_block(rank, s_)
            alns.append(gen(job_seed, first, cnt) if cnt else torch.empty((0, 4), dtype=torch.int32, device=dev))
            qlo_of.append(stream_qlo(first)); qbound_of.append(stream_qbound(first + cnt, qb))
    torch.cuda.synchronize()
    sizes = [int(a.shape[0]) for a in alns]
    if not strong:
        total_tuples, job_tuples_timed = world * nb * B, world * args.steps * B
    elif by_name:
        total_tuples, job_tuples_timed = nb * Bg, args.steps * Bg
    else:                        # (every rank's first `warmup` batches are its warm-up: what is timed is the rest of every block)
        total_tuples = nb * Bg
        job_tuples_timed = sum(ratio_block(r, s_)[1] for r in range(world) for s_ in range(args.warmup, nb))

    def qlo(step):
        return qlo_of[step]

    def qbound(step):
        return qbound_of[step]
    if nh_max > 1 and strategy == "default":
        ctr.reserve(qb)         # the per-read-name state is sized once, before any timing (a decoder that knows the number of reads)
    if strategy == "ratio" and multi:
        ctr.ratio_reserve(nb, max(sizes))      # buffers of the parked batches: allocated once, before any timing

    _pinned = {}

    def pinned16():   # pinned host copies of the batches (made once): the library's copy stream moves them (hipMemcpyAsync)
        if 16 not in _pinned:
            _pinned[16] = [a.cpu().pin_memory() for a in alns]
        return _pinned[16]

    def pinned12():   # ... packed to 12 bytes per tuple where tuples are produced (here: once, outside every timed region)
        if 12 not in _pinned:
            packed, bases12 = [], []
            for h in pinned16():
                h16 = h.numpy().view(mm.ALN_DTYPE).reshape(-1)
                base = int(h16["qid"].min()) if len(h16) else 0
                out = torch.empty(max(1, len(h16)) * 12, dtype=torch.uint8).pin_memory()
                nbig = C.c_size_t(0)
                _lib.check(_lib.lib().mm_pack12(h16.ctypes.data, len(h16), None, 0, base, C.c_void_p(out.data_ptr()), None, 0, C.byref(nbig)))
                packed.append(out); bases12.append(base)
            _pinned[12] = packed
            _pinned["bases"] = bases12
        return _pinned[12]

    def run_leg(from_host: bool, pk12: bool = False):
        """W warm-up steps, then K timed steps + end-of-file flush + cross-rank merge, bracketed by barrier + synchronize.
        Returns (elapsed seconds = max over ranks, phases of the timed region, masks, sums, stats)."""
        host = None
        if from_host:
            host = pinned12() if pk12 else pinned16()
        ctr.clear()
        ctr.set_profiling(0)

        def step(s):
            if sizes[s] == 0:
                return
            if host is not None and pk12:
                ctr.add_batch_host12(host[s].data_ptr(), sizes[s], _pinned["bases"][s], qbound(s), single_hit=single, qid_lo=qlo(s))
            elif host is not None:
                ctr.add_batch_host(host[s].data_ptr(), sizes[s], qbound(s), single_hit=single, qid_lo=qlo(s))
            else:
                ctr.add_batch_device(alns[s].data_ptr(), sizes[s], qbound(s), single_hit=single, qid_lo=qlo(s))

        for s in range(args.warmup):
            step(s)
        ctr.sync()
        # HIP events around the probe launches and the copies only: every recorded event costs a few microseconds of stream time
        ctr.set_profiling(2 if args.probe_events else 0)
        if multi:
            # first use of a collective (and of torch.unique / scatter_add) sets up RCCL channels and loads kernels: not part of a step
            if strategy == "ratio":
                warm_up(device=comm_dev, p2p=True)
            else:
                merge_counts_device(ctr, dev, rehearsal=True)      # (empty tables: the counter keeps its warm-up batches as they are)
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(args.warmup, nb):
            step(s)
        ctr.flush()                                   # end of file: pending groups (mmannot.cpp:1783-1792)
        if strategy == "ratio" and multi:         # ranks hold consecutive blocks of the stream: chain the ordered double sums
            # (row chunks pipeline the chain; they pay when the hops stay on the devices - RCCL - and cost more than they hide through the host)
            masks, sums, stats = chain_ratio(ctr, ctr.stats(), device=comm_dev, n_chunks=int(os.environ.get("MM_RATIO_CHUNKS", "8" if comm_dev.type == "cuda" else "1")))
        elif multi:                               # tables add up: one all_gather of the fixed-size tables, summed on every GPU (dist.merge_counts_device)
            masks, sums, stats = merge_counts_device(ctr, dev)
        else:
            stats = ctr.stats()
            masks, sums = ctr.getCounts()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        t1 = time.perf_counter()
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=comm_dev)
        if multi:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        phases = ctr.phase_ms()
        ctr.set_profiling(0)
        # size-independent sanity of the whole job (every tuple was handed to scan(); every group counted at most once)
        assert stats["nHits"] == total_tuples, (stats, total_tuples)
        assert strategy == "ratio" or int(np.rint(sums).sum()) <= stats["nReads"]
        return float(el.item()), phases, masks, sums, stats

    res_host = run_leg(True, args.tuple_bytes == 12) if args.legs in ("both", "host") else None
    res_pk = None
    if args.legs == "both" and args.tuple_bytes == 16:      # the same job with 12 bytes per tuple on the link
        try:
            res_pk = run_leg(True, True)
        except _lib.MMError as e:                          # a batch does not fit the packed format (e.g. > 2^24 read names in one batch): no such leg
            if e.code != 3:
                raise
    res_dev = run_leg(False) if args.legs in ("both", "resident") else None
    probe_stats = ctr.probe_stats() if res_dev is not None else None       # (of the resident leg: warm-up + timed batches)
    probe_path = ctr.probe_path()          # the context has measured its two probes on its first batches and kept the faster one
    if res_host is not None and res_pk is not None:
        oh, op = np.argsort(res_host[2]), np.argsort(res_pk[2])
        assert res_host[4] == res_pk[4] and np.array_equal(res_host[2][oh], res_pk[2][op])
        assert np.array_equal(np.ascontiguousarray(res_host[3][oh]).view(np.uint64), np.ascontiguousarray(res_pk[3][op]).view(np.uint64))
    if res_host is not None and res_dev is not None:   # the two legs are the same job
        oh, od = np.argsort(res_host[2]), np.argsort(res_dev[2])
        assert res_host[4] == res_dev[4] and np.array_equal(res_host[2][oh], res_dev[2][od])
        assert np.array_equal(np.ascontiguousarray(res_host[3][oh]).view(np.uint64), np.ascontiguousarray(res_dev[3][od]).view(np.uint64))

    # phase breakdown: resident batches once more on a cleared context, every phase bracketed by events (not part of any rate)
    n_phase = min(args.steps, 10)
    if strategy == "ratio" and multi:      # (deferred accumulation: its merge phase runs inside the rank-to-rank chain)
        phases_all = (res_dev or res_host)[1]
        n_phase = args.steps
    else:
        ctr.clear()
        for s in range(args.warmup):
            ctr.add_batch_device(alns[s].data_ptr(), sizes[s], qbound(s), single_hit=single, qid_lo=qlo(s))
        ctr.sync()
        ctr.set_profiling(1)
        for s in range(args.warmup, args.warmup + n_phase):
            ctr.add_batch_device(alns[s].data_ptr(), sizes[s], qbound(s), single_hit=single, qid_lo=qlo(s))
        ctr.sync()
        phases_all = ctr.phase_ms()
        ctr.set_profiling(0)

    # ---------------- more than one rank: is the merged result what ONE GPU gives for the same work?  Rank 0 re-runs every rank's share
    # (untimed, resident, batch by batch from the generator) and compares tables, counters and - under -y ratio - the sums' bit patterns.
    merge_check = None
    if multi and rank == 0:
        lead_ = res_dev if res_dev is not None else res_host
        merge_check = check_against_one_gpu(mm, torch, index, wl, gen, dict(strong=strong, by_name=by_name, world=world, nb=nb, B=B, Bg=Bg, ratio_block=ratio_block,
                                                                               job_seed=job_seed, single=single, nh_max=nh_max),
                                            (stream_qb, stream_qlo, stream_qbound), lead_[2], lead_[3], lead_[4])
        assert merge_check["verified"], merge_check

    if rank == 0:
        lead = res_dev if res_dev is not None else res_host          # `value`: the resident leg (--legs host: only that one ran)
        elapsed, _, masks, sums, stats = lead
        job = job_tuples_timed
        B = int(round(sum(sizes[args.warmup:]) / max(1, args.steps)))       # this rank's mean timed batch (strong scaling by name: sizes vary)
        sharding = ("one seeded stream dealt out by read name: qid mod N, names renumbered qid / N, file order kept (as mm_partition); "
                    "tables merged by all_gather of keys + all_reduce of counts") if by_name else \
                   ("one seeded stream cut into N consecutive blocks, the ordered double sums chained rank to rank" if strong else
                    ("every rank its own seeded stream; " + ("ordered double sums chained rank to rank" if strategy == "ratio"
                                                             else "tables merged by all_gather of keys + all_reduce of counts")))
        out = {
            "metric": "annotated alignments/sec", "value": job / elapsed, "unit": "alignments/s", "n_gpus": n_joined, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32/u64 (f64 only under -y ratio)", "data": "synthetic",
            "config": {"workload": wname, "batch_alignments_per_gpu": B, "intervals": int(len(an.start)), "chromosomes": int(len(chr_len)), "order_elements": len(an.config),
                       "index_bytes": info, "sharding": sharding,
                       "input": "16-byte tuples resident in HBM when the timed region starts" + ("" if res_host is not None else " (--legs resident: no host-fed leg in this run)")
                                if res_dev is not None else
                                f"{args.tuple_bytes}-byte tuples in pinned host memory, hipMemcpyAsync inside the timed region (--legs host: NOT the metric of record)"},
            "stats": stats, "table_rows": int(len(masks)),
            # sha256 over the table sorted by element set (sets, sums' bit patterns) and the seven counters: equal digests = equal results.
            # --scaling strong: the same for every N (one stream); weak: N streams, comparable only at equal N
            "table_digest": table_digest(masks, sums, stats),
        }
        if merge_check is not None:
            out["merge_check"] = merge_check
        if strategy == "ratio" and multi and getattr(chain_ratio, "last_hop_ms", None):
            hm = chain_ratio.last_hop_ms
            out["ratio_chain"] = {"chunks": len(hm), "hop_ms_rank0": hm, "note": "rank 0, last leg: per row chunk, replay of this rank's parked batches + "
                                  "hand-over of the table to the next rank, device to device (no host hop)"}
        if res_host is not None:
            h_ms, h_n = res_host[1]["h2d"]
            avg_copy_s = h_ms / max(1, h_n) / 1e3
            link_b = float(args.tuple_bytes)
            link = link_b * B / avg_copy_s / 1e9 if avg_copy_s > 0 else 0.0
            out["h2d"] = {"bound": "pcie", "achieved": link, "peak": PCIE_PEAK_GBS, "unit": "GB/s", "frac": link / PCIE_PEAK_GBS,
                          "alg_bytes_per_alignment": B_ALG_H2D, "link_bytes_per_alignment": link_b, "avg_copy_ms": avg_copy_s * 1e3,
                          "end_to_end": job / res_host[0] / world * link_b / 1e9,
                          "note": "achieved = bytes per tuple on the link x batch / average hipMemcpyAsync duration (HIP events on the copy stream, "
                                  "timed region); end_to_end = per-GPU host-fed rate x those bytes: what the whole pipeline sustains on the link"}
            out["from_host"] = {"value": job / res_host[0], "unit": "alignments/s", "ms_per_step": res_host[0] / args.steps * 1e3,
                                "link_bytes_per_alignment": link_b,
                                "note": "same job, same table: the batches start in pinned host memory, every hipMemcpyAsync is inside the timed region "
                                        "(overlapped with the previous batch's kernels) - the PCIe-inclusive rate, bound by the link (h2d)"}
            if res_dev is None and args.tuple_bytes != 16:
                out["metric"] = "annotated alignments/sec, tuples packed to 12 B on the link (informational; the metric of record uses 16-B tuples)"
        if res_pk is not None:
            p_el = res_pk[0]
            p_ms, p_n = res_pk[1]["h2d"]
            p_copy_s = p_ms / max(1, p_n) / 1e3
            out["from_host_packed12"] = {"value": job / p_el, "unit": "alignments/s", "ms_per_step": p_el / args.steps * 1e3, "link_bytes_per_alignment": 12.0,
                                         "link_achieved_GBs": 12.0 * B / p_copy_s / 1e9 if p_copy_s > 0 else 0.0,
                                         "note": "same job, same table; the tuples cross the link in the 12-byte format (mm_pack12 where they are "
                                                 "produced, k_expand12 on the device) - not the metric of record, which is quoted on 16-byte tuples"}
        if res_dev is not None:
            k_el, k_ph = res_dev[0], res_dev[1]
            dom = max(("probe", "sort", "merge", "compact"), key=lambda p: phases_all[p][0])
            probe_ms, probe_n = k_ph["probe"]
            avg_probe_s = probe_ms / max(1, probe_n) / 1e3
            achieved = B_ALG_PROBE * B / avg_probe_s / 1e9 if avg_probe_s > 0 else 0.0
            traffic, traffic_source = pmc_traffic(args.workload, probe_path)
            # whole step against the HBM roof with SURVEY.md 8d's algorithmic bytes: 32 B per alignment (stream in + index probe), 49 B with
            # the multi-hit merge (-y default, NH > 1: + 16 B mask RMW + 8 B per group)
            b_step = 49.0 if (strategy == "default" and nh_max > 1) else 32.0
            step_gbs = job / k_el / world * b_step / 1e9
            out["kernel_only"] = {"value": job / k_el, "unit": "alignments/s", "ms_per_step": k_el / args.steps * 1e3,
                                  "note": "= value: the batches resident in HBM when the timed region starts (no copy); the roofline leg",
                                  "step_roofline": {"bound": "hbm", "alg_bytes_per_alignment": b_step, "achieved": step_gbs, "peak": 8000.0,
                                                    "unit": "GB/s", "frac": step_gbs / 8000.0}}
            out["roofline"] = {"bound": "hbm", "kernel": {"pipelined": "k_probe_fast + k_probe_list", "walk": "k_probe"}.get(probe_path, "k_probe / k_probe_fast (undecided)"),
                               "probe_path": probe_path, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                               "traffic": traffic, "traffic_source": traffic_source, "alg_bytes_per_alignment": B_ALG_PROBE,
                               "avg_launch_ms": avg_probe_s * 1e3, "dominant_phase": dom,
                               # (pipelined: the probe phase = k_probe_fast, the walk over 32-byte slots, + k_probe_list, the tuples it hands on)
                               "passed_on_frac": probe_stats["passed_on"] / max(1, sum(sizes)), "fast_slot_bytes": probe_stats["fast_slot_bytes"]}
            if traffic and B == (1 << 24):     # (the counters were collected on 2^24-tuple launches) the same kernel in the bytes that really moved
                out["roofline"]["traffic_achieved"] = traffic / avg_probe_s / 1e9
                out["roofline"]["traffic_frac"] = traffic / avg_probe_s / 1e9 / 8000.0
        out["phase_ms_per_step"] = {p: phases_all[p][0] / max(1, n_phase) for p in phases_all}
        out["phase_pass"] = f"separate untimed pass of {n_phase} resident steps with events around every phase"
        if want_cpu:
            port = cpu_baseline(args, an, wl, gen_len)
            try:
                ref = cpu_baseline_reference(args, an, wl, gen_len, index, cfg)
            except Exception as e:          # (a full /tmp, a killed worker ...: the line of record must not be lost with the baseline)
                ref = None
                port["reference_error"] = f"{type(e).__name__}: {e}"[:300]
            if ref is not None:
                ref["port_single_thread"] = {k: port[k] for k in ("value", "sample")}
                out["cpu_baseline"] = ref
            else:
                out["cpu_baseline"] = port
        if numa is not None:
            out["config"]["numa_node"] = numa
        print(json.dumps(out))
    ctr.close(); index.close()
    if multi:
        dist.destroy_process_group()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: the N ranks run as children of `python -m torch.distributed.run` (one process per
    GPU, RCCL over xGMI), started from a process that has not initialised the GPU; their exit status is this process's."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def table_digest(masks, sums, stats) -> str:
    """One hash of a job's result: rows sorted by element set (set, bit pattern of the sum) + the seven counters in Counter's order."""
    from mmannot_amd import _lib
    o = np.argsort(masks, kind="stable")
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(masks[o], dtype=np.uint64).tobytes())
    h.update(np.ascontiguousarray(sums[o], dtype=np.float64).view(np.uint64).tobytes())
    h.update(np.array([stats[k] for k in _lib.STAT_NAMES], dtype=np.uint64).tobytes())
    return h.hexdigest()[:32]


def mm_device_sha() -> str:
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "mmannot_amd", "csrc", "mm_device.hip"), "rb").read()).hexdigest()[:16]
    except OSError:
        return "unknown"


def pmc_traffic(workload, probe_path="walk"):
    """HBM bytes per probe launch (k_probe, or k_probe_fast + k_probe_list where the context chose the pipelined probe) from the tracked counter summary (profiles/pmc_latest.json, written by tools/prof_summary.py from
    separate rocprofv3 --pmc passes of `bench.py --legs resident`).  The figure comes from a FILE, not from this run: it is handed out
    only while the kernel source it was collected on is the one this run was built from (sha256 of mm_device.hip), and labelled."""
    pj = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        e = json.load(open(pj)).get(workload, {})
    except Exception:
        return None, "no counter summary (profiles/pmc_latest.json)"
    t = e.get("k_probe_fast_plus_list_hbm_bytes_per_launch" if probe_path == "pipelined" else "k_probe_hbm_bytes_per_launch")
    if t is None:
        return None, "no counter summary for this workload"
    have, now = e.get("mm_device_sha"), mm_device_sha()
    if have != now:
        return None, f"stale: {e.get('source')} was collected on mm_device.hip {have}, this run is {now}"
    return t, f"file {e.get('source')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --legs resident on mm_device.hip {have}), not this run"


def check_against_one_gpu(mm, torch, index, wl, gen, J, qfun, masks, sums, stats):
    """(rank 0, after the timed legs) every rank's share of the job once more on THIS GPU, one context, no collective - the merged
    table, counters and (ratio) sum bit patterns must be the same.  What it checks is the exchange step: dist.merge_counts_device /
    dist.chain_ratio over RCCL against plain addition / plain file order."""
    stream_qb, stream_qlo, stream_qbound = qfun
    strategy = wl["strategy"]
    ctr = mm.Counter(index, overlap=wl["overlap"], strandedness=wl["strand"], strategy=strategy)
    world, nb, B, Bg = J["world"], J["nb"], J["B"], J["Bg"]
    rows, tot = {}, {}

    def take():
        ctr.flush()
        m, s_ = ctr.getCounts()
        for k, v in zip(m.tolist(), s_.tolist()):
            rows[k] = rows.get(k, 0.0) + v
        for k, v in ctr.stats().items():
            tot[k] = tot.get(k, 0) + v

    if J["strong"] and J["by_name"]:          # the job = one stream: run it whole, with its own read-name ids
        qb = stream_qb(nb * Bg)
        if J["nh_max"] > 1 and strategy == "default":
            ctr.reserve(qb)
        for s_ in range(nb):
            a = gen(J["job_seed"], s_ * Bg, Bg)
            ctr.add_batch_device(a.data_ptr(), Bg, stream_qbound((s_ + 1) * Bg, qb), single_hit=J["single"], qid_lo=stream_qlo(s_ * Bg))
            ctr.sync()
        take()
        how = "the whole stream on one GPU"
    elif strategy == "ratio":                 # file order = rank order: one context, never cleared
        for r in range(world):
            for s_ in range(nb):
                first, cnt = J["ratio_block"](r, s_) if J["strong"] else (s_ * B, B)
                if cnt == 0:
                    continue
                a = gen(J["job_seed"] + (0 if J["strong"] else r), first, cnt)
                ctr.add_batch_device(a.data_ptr(), cnt, 1, single_hit=J["single"])
                ctr.sync()
        take()
        how = "every rank's block in rank order through one context"
    else:                                     # weak: N independent streams - tables and counters add up
        qb = stream_qb(nb * B)
        if J["nh_max"] > 1 and strategy == "default":
            ctr.reserve(qb)
        for r in range(world):
            ctr.clear()
            for s_ in range(nb):
                a = gen(J["job_seed"] + r, s_ * B, B)
                ctr.add_batch_device(a.data_ptr(), B, stream_qbound((s_ + 1) * B, qb), single_hit=J["single"], qid_lo=stream_qlo(s_ * B))
                ctr.sync()
            take()
        how = "every rank's stream on one GPU, tables and counters added on the host"
    ctr.close()
    got = dict(zip(masks.tolist(), np.ascontiguousarray(sums, dtype=np.float64).view(np.uint64).tolist()))
    want = {k: int(np.float64(v).view(np.uint64)) for k, v in rows.items()}
    ok = got == want and {k: int(v) for k, v in stats.items()} == tot
    return {"verified": bool(ok), "reference": how, "rows": len(want)}


def bind_to_gpu_numa_node(device_index: int):
    """Several ranks per host: every rank streams its batches from pinned host memory at the link's rate, and pinned pages are
    allocated on the NUMA node of the thread that asks for them - so the process is bound to the CPUs of its GPU's node before it
    allocates anything (8 x 55 GB/s out of one socket's memory, half of it across the socket link, is what an unbound node can end up
    with).  Best effort: returns the node, or None when the topology is not readable or the CPUs are not this process's to use."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def host_cores() -> int:
    """Host cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one (a GPU
    box hands each job a share of the host; more processes than that only time-slice)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], int(txt[1])
            else:
                quota, period = txt[0], int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and int(quota) > 0:
                n = min(n, max(1, int(quota) // period))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_baseline(args, an, wl, chr_len):
    """The CPU restatement of the path (oracle/mm_oracle.c, kind "port", one thread) on a prefix of rank 0's stream."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth_ref as sr
    from oracle import mm_oracle as mo
    n = 1 << args.cpu_sample_log2
    t = sr.synth_tuples(args.seed * 1000003, 0, n, wl["read_len"], wl["nh_max"], chr_len, wl["scatter"], wl["pair"])
    desc = mo.ConfigDesc.from_json(an.config)
    t0 = time.perf_counter()
    r = mo.run(desc, an.as_index_dict(), t, overlap=wl["overlap"], strandedness=wl["strand"], strategy=wl["strategy"])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "alignments/s", "cores": 1, "kind": "port",
            "sample": f"first 2^{args.cpu_sample_log2} alignments of rank 0's stream, oracle/mm_oracle.c single thread, {dt:.1f} s",
            "stats": r.stats}


def _write_shard(job):
    """(worker process) one BAM shard of rank 0's stream"""
    (path, seed, first, n, read_len, nh_max, chr_len, scatter, pair, chroms) = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth_ref as sr
    from oracle import mm_oracle as mo
    t = sr.synth_tuples(seed, first, n, read_len, nh_max, chr_len, scatter, pair)
    body = mo.bam_records_fixed(t["tid"], t["start"] - 1, np.where(t["strand"] == 1, 0, 16).astype(np.uint16), t["qid"],
                                np.full(n, read_len), t["nh"] if nh_max > 1 else None)
    mo.write_bam_bytes(path, chroms, body, level=1)
    return path


def cpu_baseline_reference(args, an, wl, chr_len, index, cfg):
    """The reference itself (oracle/_ref/mmannot_patched = v1.1 + the one-line strand fix that makes -s F deterministic,
    SURVEY.md F1), compiled in the build container, on a bounded sample of the same workload: the synthetic GTF and
    configuration as files, S BAM shards of rank 0's stream, S = every host core this process may use, one single-thread
    process per shard, all running together (the reference parallelises over input files only, and its `-t` is not thread
    safe).  The GTF-loading time (the same S processes on a one-read BAM) is subtracted.  Shard 0's count table must equal
    what the GPU path gives for the same tuples."""
    import multiprocessing
    import subprocess
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "mmannot_patched")
    if not os.path.exists(ref) or an.genes is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth_ref as sr
    import mmannot_amd as mm
    from mmannot_amd import synth
    from oracle import mm_oracle as mo
    read_len, nh_max, scatter, strategy = wl["read_len"], wl["nh_max"], wl["scatter"], wl["strategy"]
    flags = ["-s", wl["strand"]] + (["-l", "%g" % wl["overlap"]] if wl["overlap"] >= 0 else []) + (["-y", strategy] if strategy != "default" else [])
    S = min(host_cores(), 32)       # (a bounded sample also in memory: every reference process holds the annotation and its read-name maps)
    # bounded sample: S shards of ~8 s of reference work each (measured on the 16-core share of an MI355X box: 2^22 alignments take the
    # reference 2.1 s beside 2.5 s of GTF loading - the quotient would rest on the smaller half; 2^24 is 3.4x the loading time)
    n = 1 << (24 if S <= 16 else (23 if S <= 32 else 22))
    d = tempfile.mkdtemp(prefix="mmbench")
    import atexit
    import shutil
    atexit.register(shutil.rmtree, d, ignore_errors=True)      # (also when something below raises: the shards are gigabytes)
    gtf = os.path.join(d, "a.gtf"); open(gtf, "w").write(synth.gtf_text(an))
    cfgp = os.path.join(d, "c.cfg"); open(cfgp, "w").write(synth.config_text(an.shape))
    chroms = [(c, int(l)) for c, l in zip(an.chr_names, an.chr_len)]
    jobs = [(os.path.join(d, f"s{i}.bam"), args.seed * 1000003, i * n, n, read_len, nh_max, chr_len, scatter, wl["pair"], chroms) for i in range(S)]
    with multiprocessing.get_context("fork").Pool(min(S, 32)) as pool:   # (fork: the workers only touch numpy and zlib)
        shards = pool.map(_write_shard, jobs)
    first_tuples = sr.synth_tuples(args.seed * 1000003, 0, n, read_len, nh_max, chr_len, scatter, wl["pair"])
    one = os.path.join(d, "one.bam")
    mo.write_bam_bytes(one, chroms, mo.bam_records_fixed(np.array([0]), np.array([100]), np.array([0], dtype=np.uint16), np.array([0]), np.array([read_len]), None), level=1)

    def limits():   # every reference process: at most 8 GiB of address space (a runaway allocation must not take the box down)
        import resource
        resource.setrlimit(resource.RLIMIT_AS, (8 << 30, 8 << 30))

    def run(file_sets, tag):
        """One single-thread reference process per file set, all started together; wall time until the last one ends."""
        t0 = time.perf_counter()
        procs = [subprocess.Popen([ref, "-a", gtf, "-c", cfgp] + flags + ["-t", "1", "-o", os.path.join(d, f"{tag}{k}.tsv"), "-r"] + fs,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, preexec_fn=limits) for k, fs in enumerate(file_sets)]
        ok = True
        for p in procs:
            try:
                _, err = p.communicate(timeout=300)
                ok = ok and p.returncode == 0
            except subprocess.TimeoutExpired:
                p.kill(); p.communicate(); ok = False
        return (time.perf_counter() - t0) if ok else None
    # The reference's own `-t` shares unsynchronised state between its threads (e.g. the list of unknown
    # chromosomes, mmannot.cpp:1299) - undefined behaviour that can run away with memory - so "all cores" is S independent
    # single-thread processes, one shard each, which is also the faster way to use it.
    t_fixed1 = run([[one]], "f")
    t_one = run([[shards[0]]], "t") if t_fixed1 is not None else None
    t_fixed = run([[one]] * S, "g") if t_one is not None else None          # S concurrent loads of the GTF
    t_full = run([[sh] for sh in shards], "s") if t_fixed is not None else None
    if t_full is None:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
        return None
    # bench-time parity: GPU path on shard 0's tuples == the reference's table for shard 0
    ctr = mm.Counter(index, overlap=wl["overlap"], strandedness=wl["strand"], strategy=strategy)
    t = first_tuples
    # the reference resolves chromosomes by name: reads on a chromosome without intervals are 'unknown' to it
    has_iv = np.zeros(an.n_chromosomes + 1, dtype=bool); has_iv[np.unique(an.chr)] = True
    tid = np.where(has_iv[t["tid"]], t["tid"], -1)
    ctr.add_batch(mm.pack_alignments(tid, t["start"], t["end"], t["strand"], t["qid"], t["nh"]), None, int(t["qid"].max()) + 1)
    ctr.flush()
    tab = mm.TableCount(cfg, 1); tab.addCounter(ctr)
    parity = tab.dump(["s0"]) == open(os.path.join(d, "t0.tsv")).read()
    ctr.close()
    assert parity, "GPU table differs from the compiled reference on shard 0"
    rate = S * n / max(1e-9, t_full - t_fixed)
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return {"value": rate, "unit": "alignments/s", "cores": S, "kind": "reference",
            "sample": f"{S} BAM shards x 2^{n.bit_length() - 1} alignments of rank 0's stream, {S} concurrent single-thread processes of "
                      f"oracle/_ref/mmannot_patched (its own -t races) = {'every host core this process may use' if S == host_cores() else f'{S} of the {host_cores()} host cores this process may use'}: {t_full:.1f} s, {t_fixed:.1f} s of it "
                      f"GTF/config loading (measured the same way, subtracted); one shard alone: {n / max(1e-9, t_one - t_fixed1):.3g} alignments/s",
            "parity_table_vs_reference": bool(parity)}


if __name__ == "__main__":
    main()
