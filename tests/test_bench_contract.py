"""GPU: bench.py keeps its contract — exactly one JSON line on stdout with the agreed keys — for one rank and for two ranks
(torch.distributed.run, both ranks sharing cuda:0 over gloo: a rehearsal of the multi-GPU path on a one-GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "kernel_only", "h2d"}


def _json_lines(out):
    return [ln for ln in out.splitlines() if ln.startswith("{")]


def test_single_rank_contract(gpu_lib):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--batch-log2", "20"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = _json_lines(p.stdout.decode())
    assert len(lines) == 1 and p.stdout.decode().strip() == lines[0]
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["value"] > 0 and d["stats"]["nHits"] == 4 * (1 << 20)
    # `value` is the rate from pinned host tuples (copies inside the timed region); the HBM-resident rate is reported beside it
    h, k = d["h2d"], d["kernel_only"]
    assert h["bound"] == "pcie" and h["unit"] == "GB/s" and h["alg_bytes_per_alignment"] == 16.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12
    assert k["value"] > 0 and "pinned host" in d["config"]["input"] and "16-byte" in d["config"]["input"]
    assert 0 < k["step_roofline"]["frac"] < 1
    p = d["from_host_packed12"]      # the same job with 12 bytes per tuple on the link, reported beside the metric of record
    assert p["value"] > 0 and p["link_bytes_per_alignment"] == 12.0


@pytest.mark.timeout(900)
def test_two_ranks_share_one_gpu_gloo(gpu_lib):
    env = dict(os.environ, MM_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch-log2", "20"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = _json_lines(p.stdout.decode())
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["stats"]["nHits"] == 2 * 4 * (1 << 20) and "cpu_baseline" not in d


@pytest.mark.timeout(900)
def test_two_ranks_ratio_chain_gloo(gpu_lib):
    """-y ratio on two ranks: deferred accumulation + rank-to-rank chain (dist.chain_ratio), collectives warmed up before timing."""
    env = dict(os.environ, MM_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch-log2", "20",
                        "--workload", "c5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = _json_lines(p.stdout.decode())
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["stats"]["nHits"] == 2 * 4 * (1 << 20) and d["table_rows"] > 0


@pytest.mark.timeout(900)
def test_two_ranks_strong_scaling_gloo(gpu_lib):
    """--scaling strong: the job's K x 2^B tuples are split over the ranks instead of every rank bringing its own."""
    env = dict(os.environ, MM_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch-log2", "20",
                        "--scaling", "strong"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads(_json_lines(p.stdout.decode())[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["stats"]["nHits"] == 4 * (1 << 20) and d["config"]["batch_alignments_per_gpu"] == 1 << 19


@pytest.mark.timeout(900)
@pytest.mark.parametrize("workload", ["c3", "c5"])
def test_rccl_backend_group_of_one(gpu_lib, workload):
    """The default backend ("nccl" = RCCL) through the multi-rank code path - process group bound to the device, warm-up, the merge's
    all_gather / all_reduce / broadcast on device tensors, the ratio chain's end - with the one GPU this box has (MM_BENCH_FORCE_DIST)."""
    env = dict(os.environ, MM_BENCH_FORCE_DIST="1")
    env.pop("MM_BENCH_BACKEND", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch-log2", "20",
                        "--workload", workload, "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads(_json_lines(p.stdout.decode())[0])
    assert d["n_gpus"] == 1 and d["stats"]["nHits"] == 4 * (1 << 20) and d["table_rows"] > 0
