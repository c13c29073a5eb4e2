#!/usr/bin/env python3
"""Summarise rocprofv3 output directories (gpurun_out/prof/<tag>_{trace,fetch,write,l2}) into profiles/.

  python tools/prof_summary.py gpurun_out/prof c3 r01

writes profiles/<round>_<workload>_kernel_stats.csv (copy of rocprofv3 --stats), and
profiles/<round>_<workload>_pmc.json: per kernel, mean per launch of FETCH_SIZE / WRITE_SIZE (KiB as rocprofv3
reports them), TCC_HIT_sum / TCC_MISS_sum, and the derived HBM bytes per launch
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(the factor 2 is the gfx950 correction of MI355X_MICROARCH.md section HBM: FETCH_SIZE reports half the bytes
of 16-B-per-lane streaming reads; it is an upper-bound correction for the narrower gather loads).
Also refreshes profiles/pmc_latest.json, which bench.py reads for roofline.traffic."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.split("<")[0].strip()


def counters(d):
    out = defaultdict(lambda: defaultdict(list))
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    files = sorted(files, key=os.path.getmtime)[-1:]   # gpurun merges into what is already there: only the newest run counts
    for f in files:
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    base, wl, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs("profiles", exist_ok=True)
    ks = glob.glob(os.path.join(base, f"{wl}_trace", "*", "*kernel_stats.csv"))
    if ks:
        ks = sorted(ks, key=os.path.getmtime)[-1:]
        shutil.copy(ks[0], f"profiles/{rnd}_{wl}_kernel_stats.csv")
    res = {}
    for tag in ("fetch", "write", "l2"):
        for k, cs in counters(os.path.join(base, f"{wl}_{tag}")).items():
            for c, v in cs.items():
                res.setdefault(k, {})[c] = dict(mean=sum(v) / len(v), launches=len(v), min=min(v), max=max(v))
    for k, d in res.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024.0
            d["hbm_bytes_per_launch_uncorrected"] = (d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"]) * 1024.0
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            h, m = d["TCC_HIT_sum"]["mean"], d["TCC_MISS_sum"]["mean"]
            d["l2_hit_rate"] = h / (h + m) if h + m else None
    json.dump(res, open(f"profiles/{rnd}_{wl}_pmc.json", "w"), indent=1, sort_keys=True)
    latest = {}
    p = "profiles/pmc_latest.json"
    if os.path.exists(p):
        latest = json.load(open(p))
    # which kernel source the counters were collected on (tools/profile_all.sh notes it on the GPU box): bench.py hands the figure out
    # only while that is the source it runs
    shaf = os.path.join(base, f"{wl}_mm_device_sha.txt")
    if os.path.exists(shaf):
        sha = open(shaf).read().split()[0][:16]
    else:
        import hashlib
        sha = hashlib.sha256(open("mmannot_amd/csrc/mm_device.hip", "rb").read()).hexdigest()[:16]
    fast, lst = res.get("k_probe_fast", {}).get("hbm_bytes_per_launch"), res.get("k_probe_list", {}).get("hbm_bytes_per_launch")
    latest[wl] = {"source": f"profiles/{rnd}_{wl}_pmc.json", "mm_device_sha": sha,
                  "k_probe_hbm_bytes_per_launch": res.get("k_probe", {}).get("hbm_bytes_per_launch"),
                  # the pipelined probe is two kernels per batch: the slot pass and the walk of the tuples it hands on
                  "k_probe_fast_plus_list_hbm_bytes_per_launch": (fast + lst) if fast is not None and lst is not None else None}
    json.dump(latest, open(p, "w"), indent=1, sort_keys=True)
    for k in sorted(res):
        d = res[k]
        print(k, {c: round(d[c]["mean"], 1) for c in d if isinstance(d[c], dict)}, "hbm_bytes", d.get("hbm_bytes_per_launch"), "l2hit", d.get("l2_hit_rate"))


if __name__ == "__main__":
    main()
