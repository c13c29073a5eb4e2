import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmannot_amd as mm
from mmannot_amd import _lib, synth
for wl in ("c2", "c3"):
    if wl == "c3":
        an = synth.synth_annotation("hs38", seed=3, scale=1.5); read_len, nh_max = 100, 1
    else:
        an = synth.synth_annotation("tair10", seed=3, scale=1.0); read_len, nh_max = 50, 1
    cfg = mm.Config.from_json(an.config)
    for shift in ((0, 11, 13) if wl == "c3" else (0, 8, 10)):
        index = mm.IntervalIndex(cfg, an.chr, an.start, an.end, an.type, an.strand, an.n_chromosomes, bin_shift=shift)
        B = 1 << 24
        dev = torch.device("cuda")
        a = torch.empty((B, 4), dtype=torch.int32, device=dev)
        chr_len = np.ascontiguousarray(an.chr_len, dtype=np.int64)
        p = _lib.SynthParams(5, 0, read_len, 1, len(chr_len), chr_len.ctypes.data, 0)
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().mm_synth_fill(a.data_ptr(), B, C.byref(p), st)); torch.cuda.synchronize()
        key = (a[:, 2].to(torch.int64) >> 1) * (1 << 32) + a[:, 0].to(torch.int64)
        order_full = torch.argsort(key)
        cum = np.concatenate([[0], np.cumsum(chr_len)])
        g = torch.from_numpy(cum[:-1]).to(dev)[(a[:, 2].to(torch.int64) >> 1) - 1] + a[:, 0].to(torch.int64)
        bucket = (g * 256 // int(cum[-1])).to(torch.int64)
        order_b256 = torch.argsort(bucket, stable=True)
        bucket4k = (g * 4096 // int(cum[-1])).to(torch.int64)
        order_b4k = torch.argsort(bucket4k, stable=True)
        variants = {"file": a, "bucket256": a[order_b256].contiguous(), "bucket4096": a[order_b4k].contiguous(), "sorted": a[order_full].contiguous()}
        for name, t in variants.items():
            ctr = mm.Counter(index, overlap=-1.0, strandedness="F", strategy="default", stream=st)
            for _ in range(2): ctr.add_batch_device(t.data_ptr(), B, 1, single_hit=True)
            ctr.sync(); ctr.set_profiling(True)
            for _ in range(5): ctr.add_batch_device(t.data_ptr(), B, 1, single_hit=True)
            ms, n = ctr.phase_ms()["probe"]
            ctr.flush(); s = ctr.stats()
            print(wl, "shift", index.info()["bin_shift"], name, "probe ms %.3f" % (ms / n), "G aln/s %.1f" % (B / (ms / n) / 1e6), s["nUnassigned"], flush=True)
            ctr.close()
        index.close()
