"""Data parallelism: one process per GPU, a full replica each, ONE collective per step.

The reference has no distributed code (per-layer nn.DataParallel wrappers only).  Here every rank
computes the reference objective on its local batch (what Lightning-DDP would give the reference), the flat
gradient buffer (flat.py: 986 890 fp32 = 3.95 MB for the CdSprites+ model) is summed with a single
torch.distributed.all_reduce -- backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests -- and the
1/world_size average is folded into the fused Adam kernel (`FlatAdam.grad_scale`).

This module IS the multi-GPU path of the product: `bench.py` and `MultimodalVAE.fused_step` call these functions and
nothing else touches torch.distributed (tests/test_parallel_gloo.py drives the same functions on two gloo ranks)."""
import os

import torch
import torch.distributed as dist

_SEED_MIX = 0x9E3779B1      # odd 32-bit constant: rank r shifts every device generator seed by r * _SEED_MIX


def init_from_env(backend=None, device=None):
    """rank, local_rank, world_size from the torchrun environment; initialises the process group when world > 1.
    Call it before anything else touches the GPU (torchrun launches one process per GPU)."""
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device if device is not None else torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, local, world


def allreduce_flat_gradients(flat_grad, world_size, group=None, force=False):
    """sum the flat gradient buffer over the ranks (in place); returns the scale that turns the sum into the mean.
    force: launch the collective on a one-rank group too (bench.py --force-collective: the multi-GPU step structure
    timed on one GPU)"""
    if world_size > 1 or (force and dist.is_initialized()):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world_size


def broadcast_flat_parameters(flat_data, src=0, group=None):
    """make every replica start from rank `src`'s parameters (one broadcast of the flat buffer)"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_data, src=src, group=group)


def rank_seed(seed, rank):
    """device generator seed of replica `rank`: distinct streams per rank, rank 0 unchanged"""
    return (int(seed) + int(rank) * _SEED_MIX) & 0x7FFFFFFF


def decorrelate_replica_noise(model, rank):
    """Replicas are built from the same torch seed (identical parameters) -- which would also give every rank the same
    reparameterisation noise and the same dropout masks on its different data shard.  Shift the counter-based device
    generators (TorchMMVAE._rng_state, every tower's DropoutState) by the rank, as independent DDP workers would
    draw."""
    if rank == 0:
        return
    st = getattr(model, "_rng_state", None)
    if st is not None:
        st[0] = rank_seed(int(st[0]), rank)
    for mod in model.modules():
        ds = getattr(mod, "state", None)
        if type(mod).__name__ == "DropoutState" and ds is not None:
            ds[0] = rank_seed(int(ds[0]), rank)


def setup_replica(trainer, rank, world_size, group=None):
    """after construction + configure_optimizers(): identical parameters everywhere (broadcast from rank 0), the mean
    of the summed gradients folded into the Adam kernel, per-rank noise / dropout streams"""
    broadcast_flat_parameters(trainer.flat.data, 0, group)
    trainer.dp_world = int(world_size)
    trainer.dp_force_collective = dist.is_initialized() and world_size == 1
    if trainer.optimizer is not None:
        trainer.optimizer.grad_scale = 1.0 / world_size
    decorrelate_replica_noise(trainer.model, rank)


def reduce_gradients_and_step(flat_grad, optimizer, world_size, group=None, force=False):
    """the tail of one data-parallel step, after the local backward pass left its gradients in the flat buffer:
    ONE all-reduce (sum), then ONE optimiser step whose kernel applies `optimizer.grad_scale` = 1 / world_size"""
    scale = allreduce_flat_gradients(flat_grad, world_size, group, force)
    assert abs(optimizer.grad_scale - scale) < 1e-12, \
        f"optimizer.grad_scale {optimizer.grad_scale} != 1/world_size {scale}: call parallel.setup_replica() first"
    optimizer.step()
