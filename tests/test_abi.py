"""CPU: the C-ABI library loads and exports every symbol include/mmannot_amd.h declares.
No compute call is made (no GPU here)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mmannot_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from mmannot_amd import _lib
    l = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(l, s), f"{s} declared in include/mmannot_amd.h but not exported"
        assert s in _lib.SYMBOLS, f"{s} has no ctypes signature in mmannot_amd/_lib.py"


def test_host_finalisation_helpers(oracle):
    """mm_round_to_uint / mm_sort_rows are host code: compare with the oracle's restatement."""
    from mmannot_amd import _lib
    l = _lib.lib()
    # `unsigned int v = round(double)` of the reference's x86-64 build (mmannot.cpp:1868): round half away from zero, cvttsd2si to
    # 64 bits ("integer indefinite" = 0x8000000000000000 for inf / nan / |v| >= 2^63), low 32 bits kept.  Known answers written out
    # by hand (KAT E4 pins the inf case through the reference itself); the oracle's restatement must agree as well.
    known = {0.0: 0, 0.49999999999999994: 0, 0.5: 1, 1.5: 2, 2.5: 3, 1e9 + 0.5: 1000000001, 4294967295.4: 4294967295, 4294967295.6: 0,
             4294967296.0: 0, 4294967297.0: 1, -0.4: 0, -1.0: 4294967295, -2.5: 4294967293, 9.2e18: 9200000000000000000 % (1 << 32),
             9.3e18: 0, -9.3e18: 0, float("inf"): 0, float("-inf"): 0, float("nan"): 0}
    for v, want in known.items():
        assert l.mm_round_to_uint(v) == want, v
        assert oracle.lib().mmo_round_to_uint(v) == want, v
    rng = np.random.default_rng(5)
    masks = np.unique(rng.integers(1, 1 << 22, 500, dtype=np.uint64) & rng.integers(1, 1 << 22, 500, dtype=np.uint64))
    masks = masks[masks != 0]
    want = sorted(masks.tolist(), key=lambda m: [i for i in range(64) if (m >> i) & 1])
    got = masks.copy()
    sums = got.astype(np.float64)
    l.mm_sort_rows(got.ctypes.data, sums.ctypes.data, len(got))
    assert got.tolist() == want and (sums == got.astype(np.float64)).all()
    for a, b in zip(want[:-1], want[1:]):
        assert oracle.lib().mmo_cmp_lists(a, b) < 0


def test_no_gpu_fails_loudly():
    """Without a device the compute entry points must fail with MM_ERR_HIP, never fall back."""
    import torch
    if torch.cuda.is_available():
        return
    import mmannot_amd as m
    cfg = m.Config([m.ConfigElement("a", "b", 0, 0)])
    try:
        m.IntervalIndex(cfg, [0], [1], [10], [0], [1], 1)
    except m.MMError as e:
        assert e.code == 2
    else:
        raise AssertionError("index creation succeeded without a GPU")


def test_cli_without_gpu_says_so():
    """The binary brings the device up in the background while it parses the GTF and opens the reads: without a device it must
    end with a message and exit code 1 - not decode into buffers it never got."""
    import os
    import subprocess
    import torch
    if torch.cuda.is_available():
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "tests", "golden", "kat", "E10_rescue")
    exe = os.path.join(root, "mmannot_amd", "mmannot-amd")
    for reads in ([os.path.join(d, "r.bam")], [os.path.join(d, "r.bam")] * 2):
        p = subprocess.run([exe, "-a", os.path.join(d, "a.gtf"), "-c", os.path.join(d, "c.cfg"), "-o", os.devnull, "-r"] + reads,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert p.returncode == 1
        assert "HIP device" in p.stderr.decode() or "hip" in p.stderr.decode().lower(), p.stderr.decode()[-300:]
