"""N > 1 readiness without a multi-GPU box: `bench.py --gpus 2` launched exactly as the driver launches it
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`), two ranks sharing the
test box's one MI355X over gloo (RCCL refuses two ranks on one device; the transport is not what is tested): ONE JSON
line from rank 0, replicas in sync after the timed steps, the collective path named.  And the two single-rank variants
of the RCCL path (`--force-collective`): the all-reduce captured into the step's graph, and the forced fallback
(MMVAE_GRAPH_COLLECTIVE=0) that launches it after the graph."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(cmd, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-3000:]}"
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected on stdout, got {len(lines)}:\n{r.stdout[-2000:]}"
    return json.loads(lines[0])


def test_bench_two_ranks_one_gpu_gloo(hip_lib):
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "6", "--warmup",
                "2", "--backend", "gloo", "--no-cpu-baseline", "--no-extras"])
    assert res["n_gpus"] == 2 and res["steps"] == 6 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 256 and res["config"]["parallelism"] == "dp2"
    assert res["replicas_in_sync"] is True
    assert res["collective"]["path"] == "after graph" and res["collective"]["backend"] == "gloo"      # gloo cannot be captured
    assert res["collective"]["bytes"] >= 986890 * 4
    assert res["value"] > 0 and abs(res["value"] - 6 * 256 / (res["ms_per_step"] * 6e-3)) / res["value"] < 1e-3
    assert "roofline" not in res and "cpu_baseline" not in res          # N = 1 only


@pytest.mark.parametrize("in_graph", ["1", "0"])
def test_bench_rccl_collective_paths_single_rank(hip_lib, in_graph):
    """the RCCL all-reduce of the N > 1 step on a one-rank group: captured into the hipGraph (validated by a replay + a
    checksum at capture), and the forced fallback path -- both must train (finite loss) and say which path they took"""
    res = _run([sys.executable, "bench.py", "--force-collective", "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                "--no-extras"], env={"MMVAE_GRAPH_COLLECTIVE": in_graph})
    assert res["n_gpus"] == 1
    assert res["collective"]["path"] == ("in graph" if in_graph == "1" else "after graph")
    assert res["collective"]["backend"] == "nccl"
    assert res["final_loss"] == res["final_loss"] and res["final_loss"] > 0


def test_bench_bare_command_starts_its_own_ranks(hip_lib):
    """`python bench.py --gpus 2 ...` with NO torchrun around it: bench.py starts the ranks itself as a
    child `torch.distributed.run` before touching the GPU and relays rank 0's line"""
    env = {k: None for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e = {k: v for k, v in os.environ.items() if k not in env}
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--backend", "gloo",
           "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-3000:]}"
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["replicas_in_sync"] is True
    assert res["collective"]["nranks"] == 2 and res["n_collectives"] == 1
    assert res["collective"]["path"] in ("in graph", "after graph")
