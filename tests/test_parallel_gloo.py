"""N > 1 path on CPU: world_size-2 gloo.  The flat-buffer protocol (parameter packing order, one all-reduce, mean
folded into the optimiser scale) must make 2 ranks with local batches equal 1 rank that averages the two batches'
gradients.  The functions driven here are the ones bench.py and MultimodalVAE.fused_step call
(parallel.init_from_env / setup_replica / reduce_gradients_and_step); the local backward pass (a hipGraph replay on the
GPU) is replaced by the oracle's gradients and the Adam KERNEL by the oracle's Adam arithmetic behind the same
`step()` / `grad_scale` interface -- the collective plumbing around them is the product code."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

MODS = [{"enc": "CNN2", "dec": "CNN", "data_dim": [64, 64, 3], "ltype": "bce"},
        {"enc": "TxtTransformer", "dec": "TxtTransformer", "data_dim": [45, 27, 1], "ltype": "category_ce"}]
B, T, D = 3, 4, 8


class _Holder(torch.nn.Module):
    """parameters with the reference's key names, packed by FlatParams (CPU tensors here)"""

    def __init__(self, params):
        super().__init__()
        self.keys = list(params)
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in params.values()])

    def as_dict(self):
        return dict(zip(self.keys, self.ps))


def _local_grads(rank, flat_holder):
    from oracle import mmvae_oracle as orc
    from multimodal_vae_comparison_amd.synthetic import cdsprites_batch
    batch = cdsprites_batch(B, T, seed=100 + rank)
    g = torch.Generator().manual_seed(7 + rank)
    eps = [torch.randn(1, B, D, generator=g) for _ in range(2)]
    out = orc.mopoe_objective(flat_holder.as_dict(), MODS, batch, eps, D)
    out["loss"].backward()          # accumulates into the preset flat .grad views
    return float(out["loss"])


def _make(seed=0):
    from oracle import golden_weights as gw
    from oracle import mmvae_oracle as orc
    from multimodal_vae_comparison_amd.flat import FlatParams
    holder = _Holder(gw.make_params(orc.model_param_shapes(MODS, D), seed))
    return holder, FlatParams(holder)


class _OracleAdam:
    """FlatAdam's interface (step(), grad_scale, clears the flat gradient buffer) with the oracle's Adam arithmetic in
    place of the HIP kernel"""

    def __init__(self, flat, lr=1e-3):
        self.flat, self.lr, self.grad_scale, self.t = flat, lr, 1.0, 0
        self.state = {"w": (torch.zeros_like(flat.data), torch.zeros_like(flat.data), torch.zeros_like(flat.data))}
        self.seen_grad = None

    def step(self):
        from oracle import mmvae_oracle as orc
        self.t += 1
        self.seen_grad = (self.flat.grad * self.grad_scale).clone()
        w = {"w": self.flat.data}
        with torch.no_grad():
            orc.adam_amsgrad_step(w, {"w": self.seen_grad}, self.state, self.lr, self.t)
            self.flat.grad.zero_()


class _Trainer:
    """the attributes parallel.setup_replica touches on a MultimodalVAE"""

    def __init__(self, holder, flat):
        self.model, self.flat, self.optimizer = holder, flat, _OracleAdam(flat)
        self.dp_world, self.dp_force_collective = 1, False


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from multimodal_vae_comparison_amd import parallel
    r, _, w = parallel.init_from_env("gloo")
    holder, flat = _make(seed=rank)                 # deliberately different initial replicas
    tr = _Trainer(holder, flat)
    # a per-tower device generator state as the real model carries it: the rank must end up in its seed
    holder.register_buffer("_rng_state", torch.tensor([1234, 0, 0], dtype=torch.int32))
    parallel.setup_replica(tr, r, w)                # broadcast from rank 0, grad_scale = 1/world, per-rank noise seeds
    data0 = flat.data.detach().numpy().copy()
    _local_grads(r, holder)
    parallel.reduce_gradients_and_step(flat.grad, tr.optimizer, tr.dp_world, None, tr.dp_force_collective)
    # numpy copies are pickled by value (torch tensors travel as shared-memory handles that die with this process)
    q.put((rank, data0, tr.optimizer.seen_grad.numpy().copy(), flat.data.detach().numpy().copy(),
           int(holder._rng_state[0]), float(flat.grad.abs().max()), tr.dp_world, tr.optimizer.grad_scale))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_average():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:          # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, data, grad, after, seed, gmax, dpw, gscale = q.get(timeout=300)
        res[rank] = (torch.from_numpy(data), torch.from_numpy(grad), torch.from_numpy(after), seed, gmax, dpw, gscale)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: same parameters (seed 0), average of the two local-batch gradients, one Adam step
    holder, flat = _make(seed=0)
    before = flat.data.clone()
    _local_grads(0, holder)
    g0 = flat.grad.clone()
    flat.grad.zero_()
    _local_grads(1, holder)
    mean = 0.5 * (g0 + flat.grad)
    opt = _OracleAdam(flat)
    flat.grad.copy_(mean)
    opt.step()
    for rank in (0, 1):
        data, grad, after, seed, gmax, dpw, gscale = res[rank]
        assert torch.equal(data, before), "broadcast must leave identical replicas"
        err = float((grad - mean).abs().max() / mean.abs().max())
        assert err < 1e-6, err
        assert dpw == 2 and gscale == 0.5 and gmax == 0.0
        # step 1 of Adam moves every element by ~lr * sign(g): compare where the gradient is not rounding noise
        well = mean.abs() > 1e-4 * mean.abs().max()
        assert float((after - flat.data)[well].abs().max()) < 1e-6, "one optimiser step on the averaged gradient"
        assert float((after - before).abs().max()) > 5e-4
    assert torch.equal(res[0][1], res[1][1]), "all ranks hold the same reduced gradient"
    assert torch.equal(res[0][2], res[1][2]), "replicas stay identical after the step"
    assert res[0][3] == 1234 and res[1][3] != 1234, "rank 0 keeps its noise seed, rank 1 gets its own stream"


def test_rank_seed_mixing_changes_every_generator():
    """replicas built from one torch seed must not draw the same eps / dropout masks on different shards"""
    from multimodal_vae_comparison_amd import parallel
    from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
    from multimodal_vae_comparison_amd.synthetic import cdsprites_config
    torch.manual_seed(0)
    tr = MultimodalVAE(cdsprites_config("mopoe", 8), device="cpu")
    def seeds(m):
        out = [int(m._rng_state[0])]
        out += [int(mod.state[0]) for mod in m.modules() if type(mod).__name__ == "DropoutState"]
        return out
    s0 = seeds(tr.model)
    assert len(s0) >= 3
    parallel.decorrelate_replica_noise(tr.model, 0)
    assert seeds(tr.model) == s0
    parallel.decorrelate_replica_noise(tr.model, 3)
    s3 = seeds(tr.model)
    assert all(a != b for a, b in zip(s0, s3)) and all(0 <= v < 2 ** 31 for v in s3)


def test_flat_params_layout_is_aligned_and_grouped():
    holder, flat = _make()
    assert flat.data.numel() % 4 == 0 and flat.n_params == sum(p.numel() for p in holder.ps)
    for p in holder.ps:
        assert p.data_ptr() >= flat.data.data_ptr() and p.grad.data_ptr() >= flat.grad.data_ptr()
        assert (p.data_ptr() - flat.data.data_ptr()) == (p.grad.data_ptr() - flat.grad.data_ptr())


def _staged_worker(rank, world, port, q):
    """parallel.StagedGradReducer on two gloo ranks: buckets that go out as ranges become final == ONE all-reduce"""
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from multimodal_vae_comparison_amd import parallel
    parallel.init_from_env("gloo")
    try:
        g = torch.Generator().manual_seed(100 + rank)
        n = 50_000
        grad = torch.randn(n, generator=g)
        ref = grad.clone()
        dist.all_reduce(ref)
        # 7 "blocks" at the back of the buffer, last block first; [0, 9000) and [46000, n) belong to other towers
        cuts = [9000, 12000, 20000, 21000, 30000, 38000, 41000, 46000]
        order = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 2, -1, -1)]
        res = {}
        for bucket_bytes in (4, 4 * 9000, 4 * 20000, 1 << 30):
            buf = grad.clone()
            sr = parallel.StagedGradReducer(buf, order, world, bucket_bytes=bucket_bytes)
            for i in range(len(order)):
                sr.mark_final(i)
            scale = sr.finish()
            assert scale == 1.0 / world
            res[bucket_bytes] = (torch.equal(buf, ref), sr.n_collectives)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_staged_reducer_is_one_all_reduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 17
    ps = [ctx.Process(target=_staged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in out:
        for bucket, (same, ncoll) in res.items():
            assert same, (rank, bucket)
        assert res[4][1] == 7 + 2 and res[1 << 30][1] == 1, res      # one bucket per block + the two outer ranges | one
        assert 1 < res[4 * 20000][1] < res[4][1]
