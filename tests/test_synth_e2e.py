"""End-to-end on seeded synthetic GTF + BAM: the `mmannot-amd` binary (GPU) and the host code + CPU oracle (CPU) against
the compiled reference's table and stderr (tests/golden/synth_e2e, oracle/gen_synth_golden.py)."""
import json
import os
import subprocess

import numpy as np
import pytest

from golden_util import check_read_stats_file

import golden_util as gu

CASES = json.load(open(os.path.join(gu.GOLD, "synth_e2e", "cases.json")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "mmannot_amd", "mmannot-amd")


SINGLE = [k for k in CASES if "cases" in CASES[k]]


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    from oracle import gen_synth_golden as g
    out = {}
    for name, c in CASES.items():
        if "cases" not in c:
            continue
        d = str(tmp_path_factory.mktemp("synth_" + name))
        out[name] = (d,) + g.make_inputs(c["workload"], d)
    return out


@pytest.mark.parametrize("name", SINGLE)
def test_host_plus_oracle(oracle, inputs, name):
    import mmannot_amd.api as api
    d, gtf, cfg, bam = inputs[name]
    for c in CASES[name]["cases"]:
        if "-m" in c["args"] or "-M" in c["args"] or "random" in c["args"]:
            continue                     # host bookkeeping of the CLI: covered by test_cli
        o = gu.parse_args(c["args"])
        an = api.Annotation(cfg, gtf, o["d"], o["D"])
        dec = api.ReadsDecoder(bam, an, unique_only=(o["strategy"] == "unique"))
        aln, nh = dec.read_all()
        desc = oracle.ConfigDesc([oracle.Element(e.source, e.type, e.strand, e.line) for e in an.config().elements])
        ix = dict(chr=an.chr, start=an.start, end=an.end, type=an.type, strand=an.strand, n_chromosomes=len(an.chromosomes))
        rd = dict(tid=api.aln_tid(aln), start=aln["start"], end=aln["end"], strand=api.aln_strand(aln), nh=nh, qid=aln["qid"])
        r = oracle.run(desc, ix, rd, overlap=o["overlap"], strandedness=o["strandedness"], strategy=o["strategy"])
        assert oracle.format_table(desc, r.row_mask, r.row_count, "r") == c["table"], c["args"]
        assert oracle.format_stats(r.stats, "$D/r.bam") in c["stderr"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", SINGLE)
def test_cli(gpu_lib, inputs, name):
    d, gtf, cfg, bam = inputs[name]
    for c in CASES[name]["cases"]:
        out = os.path.join(d, "t.tsv")
        mf, mmf = os.path.join(d, "m_out.txt"), os.path.join(d, "mm_out.txt")
        args = [mf if a == "$M" else (mmf if a == "$MM" else a) for a in c["args"]]
        p = subprocess.run([CLI, "-a", gtf, "-r", bam, "-c", cfg, "-o", out] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                           env=dict(os.environ, MM_CLI_MARK_FLUSH="1"))
        assert p.returncode == c["rc"], p.stderr.decode()
        assert open(out).read() == c["table"], c["args"]
        assert p.stderr.decode().replace(d, "$D") == c["stderr"], c["args"]
        if c.get("mfile") is not None:
            check_read_stats_file(open(mf).read(), c["mfile"])
        if c.get("mmfile") is not None:
            assert open(mmf).read() == c["mmfile"]


@pytest.mark.gpu
def test_cli_two_input_files(gpu_lib, inputs, tmp_path):
    """-r a b -n x y: one column per input in command-line order, table on stdout."""
    from oracle import gen_synth_golden as g
    c = CASES["tair10_two_files"]
    d, gtf, cfg, bam = inputs["tair10"]
    d2 = str(tmp_path)
    _, _, bam2 = g.make_inputs(c["workload2"], d2)
    p = subprocess.run([CLI, "-a", gtf, "-c", cfg, "-s", "F", "-r", bam, bam2, "-n", "first", "second"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == c["rc"]
    assert p.stdout.decode() == c["stdout"]
    assert p.stderr.decode().replace(d2, "$E").replace(d, "$D") == c["stderr"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["-t", "2"], ["-t", "2", "--devices", "0,0"], ["-t", "5", "--device", "0"]])
def test_cli_files_in_flight(gpu_lib, inputs, tmp_path, extra):
    """-t n: n input files in flight on their own contexts (spread over --devices); table and messages are those of a -t 1 run,
    columns in command-line order whatever the completion order."""
    from oracle import gen_synth_golden as g
    c = CASES["tair10_two_files"]
    d, gtf, cfg, bam = inputs["tair10"]
    d2 = str(tmp_path)
    _, _, bam2 = g.make_inputs(c["workload2"], d2)
    p = subprocess.run([CLI, "-a", gtf, "-c", cfg, "-s", "F", "-r", bam, bam2, "-n", "first", "second"] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == c["rc"], p.stderr.decode()
    assert p.stdout.decode() == c["stdout"]
    assert p.stderr.decode().replace(d2, "$E").replace(d, "$D") == c["stderr"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", SINGLE)
@pytest.mark.parametrize("devices,batch", [("0,0", None), ("0,0,0", "7001"), ("0", "7001")])
def test_cli_one_input_on_several_devices(gpu_lib, inputs, name, devices, batch):
    """--devices a,b,.. with ONE -r file shards its reads: -y default / unique deal the tuples out by read name (mm_partition), -y ratio
    sends whole batches round-robin and chains the file-order double sums (mm_merge).  Same table and stderr block as one device, also
    when the file arrives in many small batches (MM_CLI_BATCH; with one device this exercises the asynchronous pipeline alone)."""
    d, gtf, cfg, bam = inputs[name]
    env = dict(os.environ)
    if batch:
        env["MM_CLI_BATCH"] = batch
    ran = 0
    for c in CASES[name]["cases"]:
        if "-m" in c["args"] or "-M" in c["args"] or "random" in c["args"]:
            continue                     # the sequential side channels stay on one device
        out = os.path.join(d, "t_multi.tsv")
        p = subprocess.run([CLI, "-a", gtf, "-r", bam, "-c", cfg, "-o", out, "--devices", devices] + list(c["args"]),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
        assert p.returncode == c["rc"], p.stderr.decode()
        assert open(out).read() == c["table"], c["args"]
        assert p.stderr.decode().replace(d, "$D") == c["stderr"], c["args"]
        ran += 1
    assert ran > 0


@pytest.mark.gpu
def test_cli_random_over_several_files(gpu_lib, inputs, tmp_path):
    """-y random with several -r files (taken one after the other): rand() runs on across the files and a read name already output
    in an earlier file is never output again (the reference's Counter keeps seen / chosenId / numberSeen through clear(),
    mmannot.cpp:1742-1747).  Every column equals the compiled reference's."""
    from oracle import gen_synth_golden as g
    c = CASES["tair10_two_files_random"]
    d, gtf, cfg, bam = inputs["tair10"]
    d2 = str(tmp_path)
    _, _, bam2 = g.make_inputs(c["workload2"], d2)
    for run in c["runs"]:
        files = [bam if ch == "a" else bam2 for ch in run["files"]]
        p = subprocess.run([CLI, "-a", gtf, "-c", cfg, "-s", "U", "-y", "random", "-r"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == run["rc"], p.stderr.decode()
        assert p.stdout.decode() == run["stdout"], run["files"]
        assert p.stderr.decode().replace(d2, "$E").replace(d, "$D") == run["stderr"]
