"""The real data-parallel step on hardware with TWO ranks: two processes share the one MI355X of the test box (RCCL
refuses two ranks on one device, so the process group is gloo, which moves CUDA tensors through the host -- the
collective's transport is not what is tested).  Each rank runs the product path end to end: parallel.init_from_env,
MultimodalVAE, parallel.setup_replica, capture(world 2) = hipGraph without the optimiser, then per step: graph replay
on its OWN batch -> ONE all-reduce of the flat gradient buffer -> fused Adam with grad_scale 1/2.
Checked: replicas start identical (broadcast), the reduced buffer is the sum of the two local gradients, parameters stay
bit-identical across ranks after several steps, noise / dropout streams differ between the ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    try:
        from multimodal_vae_comparison_amd import parallel
        from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
        from multimodal_vae_comparison_amd.synthetic import cdsprites_batch, cdsprites_config
        r, _, w = parallel.init_from_env("gloo")
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        torch.manual_seed(10 + rank)                       # deliberately different replicas before the broadcast
        tr = MultimodalVAE(cdsprites_config("mopoe", 16, lr=1e-3), device=dev)
        tr.model.train()
        tr.configure_optimizers()
        before = tr.flat.data.clone()
        parallel.setup_replica(tr, r, w)
        start = tr.flat.data.clone()
        seeds = [int(tr.model._rng_state[0])] + [int(m.state[0]) for m in tr.model.modules()
                                                 if type(m).__name__ == "DropoutState"]
        batch = cdsprites_batch(32, 12, seed=50 + rank, device=dev)
        tr.capture(batch, world_size=w)
        assert not tr._adam_in_graph and tr.dp_world == 2 and tr.optimizer.grad_scale == 0.5
        # one step by hand: local gradients, their all-reduce, then the optimiser
        tr._graph.replay()
        torch.cuda.synchronize()
        local = tr.flat.grad.clone()
        gathered = [torch.empty_like(local) for _ in range(w)]
        dist.all_gather(gathered, local)
        parallel.allreduce_flat_gradients(tr.flat.grad, w)
        torch.cuda.synchronize()
        sum_err = float((tr.flat.grad - (gathered[0] + gathered[1])).abs().max())
        local_differ = float((gathered[0] - gathered[1]).abs().max())
        tr.optimizer.step()
        # ... and the product's own step function a few more times
        losses = [float(tr.fused_step(w)["loss"]) for _ in range(6)]
        torch.cuda.synchronize()
        peers = [torch.empty_like(tr.flat.data) for _ in range(w)]
        dist.all_gather(peers, tr.flat.data)
        q.put((rank, {"moved_by_broadcast": bool((before != start).any()) if rank else True,
                      "start_equal": None, "sum_err": sum_err, "local_differ": local_differ,
                      "replicas_equal": bool(torch.equal(peers[0], peers[1])),
                      "moved": float((tr.flat.data - start).abs().max()), "losses": losses, "seeds": seeds,
                      "start_sum": float(start.double().sum()), "steps": int(tr.optimizer.step_dev[0])}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise


def test_two_ranks_on_one_gpu_run_the_real_step(hip_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, d = q.get(timeout=600)
        res[rank] = d
    for p in procs:
        p.join(timeout=120)
    for r in (0, 1):
        assert "error" not in res[r], res[r].get("error")
    for p in procs:
        assert p.exitcode == 0
    a, b = res[0], res[1]
    assert a["start_sum"] == b["start_sum"], "broadcast: identical replicas at the start"
    assert b["moved_by_broadcast"], "rank 1 was built from another seed: the broadcast must have overwritten it"
    for d in (a, b):
        assert d["sum_err"] == 0.0, "the reduced buffer is exactly the sum of the two local gradients (gloo sums in order)"
        assert d["local_differ"] > 0.0, "the ranks saw different batches"
        assert d["replicas_equal"], "parameters stay bit-identical across the ranks"
        assert d["moved"] > 0.0 and d["steps"] == 7
        assert all(torch.isfinite(torch.tensor(d["losses"])))
    assert a["seeds"][0] != b["seeds"][0] and all(x != y for x, y in zip(a["seeds"], b["seeds"])), \
        "per-rank noise / dropout streams"
