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
        "config", "roofline", "kernel_only", "h2d", "from_host"}


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
    assert isinstance(r["traffic_source"], str) and (r["traffic"] is None or "file" in r["traffic_source"])   # a counter figure by file says so
    assert len(d["table_digest"]) == 32 and "merge_check" not in d
    assert d["value"] > 0 and d["stats"]["nHits"] == 4 * (1 << 20)
    # `value` is the rate with the tuples resident in HBM when the timed region starts; the rate from pinned host tuples (copies inside
    # the timed region) is reported beside it
    h, k, fh = d["h2d"], d["kernel_only"], d["from_host"]
    assert h["bound"] == "pcie" and h["unit"] == "GB/s" and h["alg_bytes_per_alignment"] == 16.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12
    assert k["value"] == d["value"] and "resident in HBM" in d["config"]["input"] and "16-byte" in d["config"]["input"]
    assert 0 < fh["value"] <= d["value"] * 1.05 and abs(h["end_to_end"] - fh["value"] * 16.0 / 1e9) < 1e-6
    assert 0 < k["step_roofline"]["frac"] < 1
    p = d["from_host_packed12"]      # the same job with 12 bytes per tuple on the link, reported beside the metric of record
    assert p["value"] > 0 and p["link_bytes_per_alignment"] == 12.0


def _bench(args, env=None, launcher_ranks=None, port=29541):
    """bench.py as the driver calls it (plain `python bench.py ...`: with --gpus N > 1 it starts its ranks itself), or - launcher_ranks -
    under torch.distributed.run."""
    cmd = [sys.executable]
    if launcher_ranks:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher_ranks), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800, cwd=ROOT, env=env or os.environ)


def _one_line(p):
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = _json_lines(p.stdout.decode())
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


SMALL = ["--steps", "3", "--warmup", "1", "--batch-log2", "20", "--no-cpu-baseline"]


@pytest.mark.timeout(900)
def test_plain_gpus_2_starts_two_ranks_gloo(gpu_lib):
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (child process), they share cuda:0 over gloo;
    n_gpus is the number of ranks that joined; rank 0 re-runs both streams alone and finds the merged table (merge_check)."""
    env = dict(os.environ, MM_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    d = _one_line(_bench(["--gpus", "2"] + SMALL, env))
    assert d["n_gpus"] == 2 and d["stats"]["nHits"] == 2 * 4 * (1 << 20) and "cpu_baseline" not in d
    assert d["merge_check"]["verified"] is True and len(d["table_digest"]) == 32


@pytest.mark.timeout(900)
def test_world_size_mismatch_is_an_error(gpu_lib):
    """Two ranks from a launcher but --gpus 1 (or the reverse): an error, not a silent run at another size."""
    env = dict(os.environ, MM_BENCH_BACKEND="gloo")
    p = _bench(["--gpus", "1"] + SMALL, env, launcher_ranks=2, port=29551)
    assert p.returncode != 0 and b"WORLD_SIZE=2" in p.stderr and not _json_lines(p.stdout.decode())


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("workload", ["c3", "c4", "c5"])
def test_strong_scaling_same_digest_for_every_n(gpu_lib, workload):
    """--scaling strong deals ONE seeded stream out over the ranks (by read name, qid mod N; -y ratio: consecutive blocks), so the
    job's table is the same for every N: equal table_digest from 1 and 2 ranks (gloo, sharing cuda:0)."""
    env = dict(os.environ, MM_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    args = SMALL + ["--scaling", "strong", "--workload", workload, "--legs", "resident"]
    d1 = _one_line(_bench(["--gpus", "1"] + args, env))
    d2 = _one_line(_bench(["--gpus", "2"] + args, env))
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d1["scaling"] == d2["scaling"] == "strong"
    assert d1["stats"] == d2["stats"] and d1["stats"]["nHits"] == 4 * (1 << 20)
    assert d1["table_digest"] == d2["table_digest"] and d1["table_rows"] == d2["table_rows"] > 0
    assert d2["merge_check"]["verified"] is True
    # (about half each; not exactly: the generator's read-name ids and NH values are correlated, ids of one parity hold the odd NH)
    assert abs(d2["config"]["batch_alignments_per_gpu"] - (1 << 19)) < (1 << 19) * 0.1


@pytest.mark.timeout(900)
def test_two_ranks_under_a_launcher_ratio_chain_gloo(gpu_lib):
    """The way the driver starts N > 1 (torch.distributed.run): -y ratio on two ranks - deferred accumulation + rank-to-rank chain
    (dist.chain_ratio), collectives warmed up before timing - and the chain's result checked against one context in rank order."""
    env = dict(os.environ, MM_BENCH_BACKEND="gloo")
    d = _one_line(_bench(["--gpus", "2", "--workload", "c5"] + SMALL, env, launcher_ranks=2, port=29543))
    assert d["n_gpus"] == 2 and d["stats"]["nHits"] == 2 * 4 * (1 << 20) and d["table_rows"] > 0 and d["merge_check"]["verified"] is True


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
