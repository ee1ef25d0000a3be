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


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the 102 MB model.  `encoder: CNN` (ResNet-50, 25.6 M parameters: what every shipped CdSprites+ / CUB / VILANRO
# config selects) would put ONE 102 MB all-reduce behind a ~4 ms backward pass: 2 (N-1)/N x 102 MB over 153 GB/s xGMI
# links is ~1.2 ms at 8 GPUs, none of it hidden.  The tower's backward walks its 16 bottlenecks from layer4 down, and
# layer4 alone is 15 M of the parameters: its gradients are final after 3 of the 16 blocks.
# ---------------------------------------------------------------------------------------------------------------------
class StagedGradReducer:
    """All-reduce of the flat gradient buffer in BUCKETS that go out while the backward pass is still running.

    `order`: [(start, end)] element ranges of the flat buffer in the order they become final during the backward pass
    (the ResNet bottlenecks, last block first -- flat.py lays parameters out in module order, so these ranges grow
    downwards contiguously).  `mark_final(i)` is called by the backward pass (rconv.BLOCK_DONE_HOOK) when range i can no
    longer change; whenever >= `bucket_bytes` of final, not yet reduced gradient have accumulated one all-reduce of that
    contiguous range is issued on the communication stream (RCCL: behind an event of the compute stream, so it runs beside
    the rest of the backward pass -- in a captured step the collective nodes sit on a side branch of the graph; gloo: at
    once).  `finish()` reduces what is left -- the ranges outside `order` included: the other towers, folded partials --
    and joins the streams; the optimiser step follows.  The sum of slices is the slice of the sum: the result is
    bit-identical to ONE all-reduce of the whole buffer for any bucket size (tests/test_parallel_gloo.py)."""

    def __init__(self, flat_grad, order, world_size, bucket_bytes=25 << 20, group=None, force=False):
        self.g, self.order, self.world, self.group = flat_grad, [(int(a), int(b)) for a, b in order], int(world_size), group
        self.bucket = int(bucket_bytes) // 4
        self.active = self.world > 1 or (force and dist.is_initialized())
        for (a0, b0), (a1, b1) in zip(self.order, self.order[1:]):
            assert b1 == a0 or b1 <= a0, "ranges must be listed from the back of the buffer to the front, without overlap"
        self.comm = torch.cuda.Stream(device=flat_grad.device) if flat_grad.is_cuda else None
        self.n_collectives = 0
        self.begin()

    def begin(self):
        """start of a backward pass"""
        self._final = 0                 # ranges order[:_final] are final
        self._sent_lo = None            # [sent_lo, sent_hi): already reduced
        self._sent_hi = None
        self.n_collectives = 0

    def _all_reduce(self, lo, hi):
        if hi <= lo:
            return
        self.n_collectives += 1
        if not self.active:
            return
        view = self.g[lo:hi]
        if self.comm is None:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            return
        self.comm.wait_stream(torch.cuda.current_stream(self.g.device))      # everything that wrote the range is queued
        with torch.cuda.stream(self.comm):
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

    def mark_final(self, i):
        """range i (and every range before it in `order`) is final"""
        self._final = max(self._final, i + 1)
        hi = self.order[0][1] if self._sent_lo is None else self._sent_lo
        lo = self.order[self._final - 1][0]
        if hi - lo >= self.bucket:
            self._all_reduce(lo, hi)
            if self._sent_hi is None:
                self._sent_hi = hi
            self._sent_lo = lo

    def finish(self):
        """end of the backward pass (after the fold of the deferred partials): everything not reduced yet, then join"""
        n = self.g.numel()
        if self._sent_lo is None:
            self._all_reduce(0, n)
        else:
            self._all_reduce(0, self._sent_lo)
            self._all_reduce(self._sent_hi, n)
        if self.comm is not None and self.active:
            torch.cuda.current_stream(self.g.device).wait_stream(self.comm)
        return 1.0 / self.world


def resnet_block_ranges(model, flat):
    """[(start, end)] of the flat buffer per ResNet bottleneck, LAST block first (the order their gradients become
    final), + the modules in that order; ([], []) for a model without a ResNet tower.  flat.py keeps the tower's
    convolution weights (25.5 of its 25.56 M parameters) in module order in one region and the BatchNorm vectors in
    another: a block's range is the span of its convolution weights, the 53 K BatchNorm parameters go out with the rest of
    the buffer at the end of the backward pass.

    `StagedGradReducer.mark_final(i)` reduces the whole span from range i up to what was already sent, so that span must
    hold nothing but finished blocks: the ranges are only returned when they tile ONE contiguous region --
    consecutive blocks touch up to the 16-byte group alignment -- which also rules out a model with two ResNet towers
    (the second tower's stem and the first tower's head layers would sit in the gap, their kernels possibly not even
    launched when the hook fires).  Anything else gets ([], []): one collective behind the backward pass."""
    from .models.resnet import Bottleneck
    blocks = [m for m in model.modules() if isinstance(m, Bottleneck)]
    base = flat.grad.data_ptr()
    out = []
    for b in blocks:
        ps = [p for p in b.parameters() if p.dim() == 4]
        lo = min((p.grad.data_ptr() - base) // 4 for p in ps)
        hi = max((p.grad.data_ptr() - base) // 4 + p.numel() for p in ps)
        out.append((lo, hi))
    return (out[::-1], blocks[::-1]) if ranges_tile_one_region(out) else ([], [])


def ranges_tile_one_region(ranges, align=4):
    """ascending [(lo, hi)] that touch each other up to `align` elements of padding (flat.py aligns groups to 16 bytes)"""
    if not ranges:
        return False
    return all(b0 <= a1 < b0 + align for (a0, b0), (a1, b1) in zip(ranges, ranges[1:]))
