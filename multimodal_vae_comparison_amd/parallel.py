"""Data parallelism: one process per GPU, a full replica each, ONE collective per step.

The reference has no distributed code (per-layer nn.DataParallel wrappers only).  Here every rank
computes the reference objective on its local batch (what Lightning-DDP would give the reference), the flat
gradient buffer (flat.py: 986 890 fp32 = 3.95 MB for the CdSprites+ model) is summed with a single
torch.distributed.all_reduce -- backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests -- and the
1/world_size average is folded into the fused Adam kernel (`FlatAdam.grad_scale`)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """rank, local_rank, world_size from the torchrun environment; initialises the process group when world > 1"""
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return rank, local, world


def allreduce_flat_gradients(flat_grad, world_size, group=None):
    """sum the flat gradient buffer over the ranks (in place); returns the scale that turns the sum into the mean"""
    if world_size > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world_size


def broadcast_flat_parameters(flat_data, src=0, group=None):
    """make every replica start from rank `src`'s parameters (one broadcast of the flat buffer)"""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_data, src=src, group=group)
