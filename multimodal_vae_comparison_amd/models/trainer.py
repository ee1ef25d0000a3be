"""MultimodalVAE: the LightningModule surface of the reference (models/trainer.py:15-128) without the Lightning
dependency: `get_model`, `training_step(batch, idx) -> loss`, `configure_optimizers()`.  A Lightning Trainer can
drive it unchanged (nn.Module + the same hook names); `fused_step()` is the MI355X fast path used by bench.py:
forward + backward of the whole objective replayed from ONE hipGraph, optional RCCL all-reduce of the flat
gradient buffer, ONE fused Adam kernel."""
import ctypes
import os

import torch
import torch.nn as nn

from .. import flat as flatmod
from .. import hipops as H
from .. import parallel, rconv
from . import mmvae_models  # noqa: F401
from .mmvae_base import TorchMMVAE
from .vae import VAE

# feature_dims of the reference's dataset classes (models/datasets.py:207-209): the only part of the input
# pipeline the towers depend on
FEATURE_DIMS = {
    "cdspritesplus": {"image": [64, 64, 3], "text": [45, 27, 1]},                     # datasets.py:207-209
    "mnist_svhn": {"mnist": [28, 28, 1], "svhn": [32, 32, 3]},                        # datasets.py:418-420
}


class MultimodalVAE(nn.Module):
    def __init__(self, cfg, feature_dims=None, device="cuda"):
        super().__init__()
        from .config_cls import Config
        self.config = cfg if isinstance(cfg, Config) else Config(cfg)
        self.feature_dims = feature_dims or FEATURE_DIMS[self.config.dataset_name.lower()]
        self.optimizer = None
        self.flat = None
        self.model = None
        self.logged = {}            # name -> last logged scalar (the record `self.log` keeps when no Lightning Trainer is attached)
        self.get_model()
        self.to(device)
        self.flat = flatmod.FlatParams(self.model)
        self._graph = None
        self.dp_world = 1           # ranks the flat gradients are summed over (parallel.setup_replica sets it)
        self.dp_force_collective = False
        if self.config.pre_trained:
            # models/trainer.py:95-97: warm start from a Lightning checkpoint (reference key names).  The reference hands
            # the path to load_from_checkpoint; here the weights go INTO the flat buffer views of the model the config
            # describes -- a missing file or a mismatching layout is an error, never a silent training from scratch
            self.load_checkpoint(self.config.pre_trained, strict=True)

    def get_model(self):
        """models/trainer.py:91-115"""
        from .. import models
        c = self.config
        vaes = {}
        for i, m in enumerate(c.mods):
            vaes["mod_{}".format(i + 1)] = VAE(m["encoder"], m["decoder"], self.feature_dims[m["mod_type"]],
                                               c.n_latents, m["recon_loss"], m["private_latents"], obj_fn=c.obj,
                                               beta=c.beta, id_name="mod_{}".format(i + 1), prior_dist=m["prior"],
                                               post_dist=m["prior"], likelihood_dist=m["prior"],
                                               llik_scaling=m["llik_scaling"])
        if len(c.mods) > 1:
            obj_cfg = {"obj": c.obj, "beta": c.beta, "K": c.K}
            self.model = getattr(models, c.mixing.lower())(nn.ModuleDict(vaes), c.n_latents, obj_cfg, c.model_cfg)
            assert isinstance(self.model, TorchMMVAE)
        else:
            self.model = vaes["mod_1"]          # unimodal VAE scenario (models/trainer.py:112-113)
        return self.model

    def configure_optimizers(self):
        """models/trainer.py:75-89: Adam(lr, amsgrad=True) over the trainable parameters, or `optimizer: adabelief`
        (:82-86: adabelief_pytorch.AdaBelief(lr, eps=1e-16, betas=(0.9, 0.999), weight_decouple=True, rectify=False) -- a
        package the reference does not vendor and this image does not have: its published update restated as one flat
        kernel, parity unpinned, flat.FlatAdaBelief); any other name raises NotImplementedError as in the reference."""
        name = self.config.optimizer.lower()
        if name == "adabelief":
            self.optimizer = flatmod.FlatAdaBelief(self.flat, lr=float(self.config.lr))
        elif name == "adam":
            self.optimizer = flatmod.FlatAdam(self.flat, lr=float(self.config.lr))
        else:
            raise NotImplementedError(self.config.optimizer)
        return self.optimizer

    def log(self, name, value, batch_size=None, **kw):
        """Lightning's `self.log` when no Trainer is attached: keep the last value per name in `self.logged` (device
        scalars: no host synchronisation here) and hand it to `log_hook(name, value, batch_size)` when one is set."""
        self.logged[name] = value.detach() if torch.is_tensor(value) else value
        hook = getattr(self, "log_hook", None)
        if hook is not None:
            hook(name, self.logged[name], batch_size)

    def _log_losses(self, loss_d, prefix, tag):
        """the logging loop shared by training_step / validation_step / test_step (models/trainer.py:121-127,134-140,
        147-153): `<prefix>_<key>` = value.sum() for every key of the objective's dict, `Mod_<i>_<Tag>Loss` = sum of
        the i-th entry of `reconstruction_loss`"""
        for key in loss_d.keys():
            if key != "reconstruction_loss":
                self.log("{}_{}".format(prefix, key), loss_d[key].sum(), batch_size=self.config.batch_size)
            else:
                for i, p_l in enumerate(loss_d[key]):
                    self.log("Mod_{}_{}Loss".format(i, tag), p_l.sum(), batch_size=self.config.batch_size)

    def training_step(self, train_batch, batch_idx=0):
        """models/trainer.py:117-128"""
        loss_d = self.model.objective(train_batch)
        self.last_losses = loss_d
        self._log_losses(loss_d, "train", "Train")
        return loss_d["loss"]

    def validation_step(self, val_batch, batch_idx=0):
        """models/trainer.py:130-141: the same objective on the validation batch (Lightning puts the module in eval mode
        and disables gradients around it; a caller without Lightning does the same), logged as val_* / Mod_i_ValLoss"""
        loss_d = self.model.objective(val_batch)
        self._log_losses(loss_d, "val", "Val")
        return loss_d["loss"]

    def test_step(self, test_batch, batch_idx=0):
        """models/trainer.py:143-154: logged as test_* / Mod_i_TestLoss"""
        loss_d = self.model.objective(test_batch)
        self._log_losses(loss_d, "test", "Test")
        return loss_d["loss"]

    # ---- checkpoints -----------------------------------------------------------
    def save_checkpoint(self, path, epoch=0, global_step=0):
        """Lightning-style `.ckpt` with the reference's key names (`model.vaes.mod_k.enc...`, `model._pz_params.1`;
        main.py:46 `ModelCheckpoint`, eval/infer.py:26 `load_from_checkpoint`): `state_dict` of the wrapped mixer
        under the `model.` prefix, `optimizer_states` in torch.optim.Adam(amsgrad) layout, `hyper_parameters`."""
        sd = {"model." + k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
        ckpt = {"epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": "1.9.5",
                "state_dict": sd, "loops": {}, "callbacks": {}, "lr_schedulers": [],
                "optimizer_states": [self.optimizer.state_dict()] if self.optimizer is not None else [],
                "hyper_parameters": {"cfg": self.config, "feature_dims": self.feature_dims}}
        if self.optimizer is not None:
            for st in ckpt["optimizer_states"][0]["state"].values():
                for k, v in list(st.items()):        # Adam(amsgrad): exp_avg / exp_avg_sq / max_exp_avg_sq; AdaBelief:
                    if torch.is_tensor(v):            # exp_avg / exp_avg_var -- whatever the optimiser keeps
                        st[k] = v.cpu()
        torch.save(ckpt, path)

    def load_checkpoint(self, path_or_ckpt, strict=True):
        """Load a checkpoint written by the reference's Lightning trainer (or by save_checkpoint): parameters are copied
        INTO the flat buffer views (the flat layout, the captured graph and the optimiser stay valid); optimiser state
        too when present.  Buffers the reference stores but this package recomputes (`pe`) are ignored."""
        ckpt = torch.load(path_or_ckpt, map_location="cpu", weights_only=False) if isinstance(path_or_ckpt, str) else path_or_ckpt
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
        own = self.model.state_dict()
        missing = [k for k, _ in self.model.named_parameters() if k not in sd]
        if strict and missing:
            raise KeyError(f"checkpoint lacks parameters: {missing[:5]}{' ...' if len(missing) > 5 else ''}")
        # torch's load_state_dict semantics: a shape mismatch is an error (another n_latents / tower config must not
        # load silently and leave those parameters at their random initialisation)
        bad = [f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}"
               for k, v in sd.items() if k in own and own[k].shape != v.shape]
        bad += [f"unexpected key {k}" for k in sd if k not in own]
        if bad and strict:
            raise RuntimeError("Error(s) in loading state_dict:\n\t" + "\n\t".join(bad))
        with torch.no_grad():
            for k, v in sd.items():
                if k in own and own[k].shape == v.shape:
                    own[k].copy_(v)                       # state_dict() tensors alias the flat views
        if self.optimizer is not None and ckpt.get("optimizer_states"):
            self.optimizer.load_state_dict(ckpt["optimizer_states"][0])
        return ckpt.get("epoch", 0), ckpt.get("global_step", 0)

    def zero_grad(self, set_to_none=False):
        """nn.Module.zero_grad defaults to set_to_none=True, which would detach every parameter from its flat gradient
        view (the weight-gradient kernels accumulate straight into those views): clear the flat buffer instead"""
        if self.optimizer is not None:
            self.optimizer.zero_grad()
        elif self.flat is not None:
            self.flat.rebind_grads()
            self.flat.zero_grad()

    # ---- MI355X fast path ------------------------------------------------------------------------
    def capture(self, batch, world_size=1, optimizer_in_graph=None, input_ring=None):
        """Capture objective + backward (+ the Adam step when there is no collective between them, world_size 1) for
        `batch`'s shapes into a hipGraph.  `batch` tensors become the static input buffers: copy new data into them
        (`load_batch`) before each replay.  optimizer_in_graph=False keeps the optimiser step out of a one-rank graph
        (fused_step then launches it after the replay: the replayed gradients stay readable in between).
        input_ring: a list of `pack_compact_pinned()` host batches of one layout -- the INPUT STEP becomes part of the
        graph (GraphInputRing): replay j trains on ring[j % len(ring)], with no runtime call per step but the graph
        launch itself."""
        assert self.optimizer is not None, "call configure_optimizers() first"
        # the collective and the 1/world mean come from parallel.setup_replica (dp_world, optimizer.grad_scale), not from
        # this argument: a caller that asks for a multi-rank step without having set the replica up would otherwise train
        # diverging replicas without a word
        assert world_size == 1 or world_size == self.dp_world or self.dp_force_collective, \
            f"capture(world_size={world_size}) but the trainer is set up for {self.dp_world} rank(s): call " \
            f"parallel.setup_replica(trainer, rank, world_size) first"
        self._static_batch = batch
        self._one = torch.ones((), device=self.flat.data.device)     # loss.backward() seed: no fill kernel per step
        from .. import ops
        import weakref
        ptr = self._one.data_ptr()
        ops.LincombRows.unit_seed_ptr = ptr                          # ... and the ELBO assembly's backward is free

        def _clear(ptr=ptr):        # the address may be recycled once this trainer is gone
            if ops.LincombRows.unit_seed_ptr == ptr:
                ops.LincombRows.unit_seed_ptr = None
        weakref.finalize(self, _clear)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):                       # warm-up: sizes the shared workspace, loads code objects
                self._fwd_bwd(batch)
                self._finish_step()
            self.flat.zero_grad()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._input_ring = GraphInputRing(self, input_ring) if input_ring else None
        self._graph = torch.cuda.CUDAGraph()
        dump = os.environ.get("MMVAE_GRAPH_DUMP")          # a DOT file of the captured step (node count: bench.py)
        if dump:
            self._graph.enable_debug_mode()
        self._graph2 = None
        self._adam_in_graph = world_size == 1 if optimizer_in_graph is None else bool(optimizer_in_graph) and world_size == 1
        # MMVAE_GRAPH_COLLECTIVE=1 (data parallel): the all-reduce of the flat gradients and the Adam launch are captured
        # INTO the step's graph (RCCL collectives are stream-capturable): the whole N > 1 step is one graph launch, with
        # no host-side launches between the backward pass, the collective and the optimiser
        # (measured with a one-rank RCCL group, --force-collective: 0.457 -> 0.442 ms/step = the one-GPU time).  gloo
        # collectives run on the host and cannot be captured; a capture failure falls back to the launches after the graph.
        self._collective_in_graph = (world_size > 1 and torch.distributed.is_initialized()
                                     and torch.distributed.get_backend() == "nccl"
                                     and os.environ.get("MMVAE_GRAPH_COLLECTIVE", "1") == "1")
        # data parallel, optional (MMVAE_DP_OVERLAP=1): cut the backward at the fusion.  The decoders' (and the prior's)
        # gradients -- the second range of the flat buffer -- are final after the first graph, so their all-reduce runs
        # under the second graph (the encoders' backward) instead of after the whole step.  Bit-identical training
        # (tools/probe/dp_overlap_check.py), but OFF by default: on one rank, where the collectives move nothing, the
        # second graph launch, the second collective and the second fold cost 75 us per step (0.458 -> 0.532 ms), more
        # than a 2 MB ring all-reduce over xGMI takes -- it needs a cheaper cut before it pays.
        overlap = (world_size > 1 and hasattr(self.model, "backward_encoders") and 0 < self.flat.split < self.flat.grad.numel()
                   and os.environ.get("MMVAE_DP_OVERLAP", "0") == "1")
        if overlap:
            with torch.cuda.stream(s):                # one warm-up of the two-phase path
                self.model.objective_backward(batch, cut=True)
                self._finish_step()
                self.model.backward_encoders()
                self.flat.zero_grad()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            with torch.cuda.graph(self._graph):
                out = self.model.objective_backward(batch, cut=True)
                self._finish_step()
            self._graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph2, pool=self._graph.pool()):
                self.model.backward_encoders()
        else:
            def record(with_collective):
                kw = {"capture_error_mode": "thread_local"} if with_collective else {}
                calls0 = ops.CALLS[0]
                with torch.cuda.graph(self._graph, **kw):
                    ops.Marks.mark("step start")
                    if self._input_ring is not None:
                        # head: the staged compact batch -> the step's static inputs; the NEXT batch is pulled under
                        # this step, behind the side (text) tower's last backward launch
                        ring = self._input_ring
                        ring.expand()
                        ring.done = False
                        ops.GradReducer.side_tail = ops.GradReducer.pre_join = ring.pull_next
                        ring.split_pct = int(os.environ.get("MMVAE_RING_SPLIT", "50"))
                        ring.split = ring.split_pct > 0 and hasattr(self.model, "objective_backward")
                        ops.GradReducer.side_head = ring.pull_first_half if ring.split else None
                    # one GPU: the optimiser follows the backward at once, so the end-of-backward fold is left to it
                    ops.GradReducer.defer_next = self._adam_in_graph and self.optimizer.supports_fold
                    # ... and its second range (decoders, prior) does not wait for the end of the step: MoPOE.
                    # objective_backward runs it behind the fusion's backward (MMVAE_EARLY_ADAM=0: one launch at the end)
                    self.model.early_adam = None
                    if ops.GradReducer.defer_next and 0 < self.flat.split < self.flat.grad.numel() and \
                            os.environ.get("MMVAE_EARLY_ADAM", "2") != "0":
                        ops.GradReducer.early_at = int(os.environ.get("MMVAE_EARLY_ADAM", "2"))
                        lo, hi, g0 = self.flat.split, self.flat.grad.numel(), self.flat.grad.data_ptr()
                        self.model.early_adam = (lambda table, lo=lo, hi=hi: self.optimizer.step_early(lo, hi, table),
                                                 g0 + 4 * lo, g0 + 4 * hi)
                    stager = self._stager if with_collective else None
                    if stager is not None:
                        stager.begin()
                        rconv.BLOCK_DONE_HOOK = lambda blk: stager.mark_final(self._stager_index[id(blk.mod)])
                    try:
                        res = self._fwd_bwd(batch)
                    finally:
                        rconv.BLOCK_DONE_HOOK = None
                    ops.GradReducer.defer_next = False
                    self.model.early_adam = None
                    ops.GradReducer.pre_join = ops.GradReducer.side_tail = ops.GradReducer.side_head = None
                    if self._input_ring is not None:
                        self._input_ring.pull_next(self.flat.data.device, [])     # (no hook point ran: pull here)
                    ops.Marks.mark("backward done")
                    if self._adam_in_graph:
                        self.optimizer.step()
                        ops.Marks.mark("adam done")
                    self._finish_step()
                    if with_collective and stager is not None:
                        # the ResNet tower's buckets went out while its backward pass ran (rconv.BLOCK_DONE_HOOK); the
                        # rest of the flat buffer now, then the optimiser step on the joined stream
                        stager.finish()
                        self.optimizer.step()
                    elif with_collective:
                        parallel.reduce_gradients_and_step(self.flat.grad, self.optimizer, self.dp_world, None,
                                                           self.dp_force_collective)
                self.abi_calls_in_graph = ops.CALLS[0] - calls0      # C-ABI calls of one captured step (~ graph kernel nodes)
                return res
            # the 102 MB model (encoder: CNN): bucketed all-reduce beside the ResNet tower's backward pass instead of one
            # collective behind it (parallel.StagedGradReducer; only when the collective is part of the captured step)
            self._stager = None
            if self._collective_in_graph and os.environ.get("MMVAE_DP_STAGED", "1") == "1":
                ranges, mods = parallel.resnet_block_ranges(self.model, self.flat)
                if ranges:
                    self._stager = parallel.StagedGradReducer(self.flat.grad, ranges, self.dp_world,
                                                              force=self.dp_force_collective)
                    self._stager_index = {id(m): i for i, m in enumerate(mods)}
            if self._collective_in_graph:
                # EVERY rank reaches the verdict all-reduce (a rank whose capture raised used to skip it and
                # leave the others hanging in it), and a failed capture leaves no hook set on GradReducer
                captured = None
                try:
                    out = record(True)
                except RuntimeError as e:          # the runtime refused to capture the collective
                    captured = f"all-reduce not capturable here ({e})"
                finally:
                    ops.GradReducer.defer_next = False
                    ops.GradReducer.pre_join = ops.GradReducer.side_tail = None
                why = self._validate_graph_collective(captured)
                if why is None and self._input_ring is not None:
                    self._input_ring.reprime()      # the validation replay consumed a slot
                if why is not None:                # launch it after the graph instead (the round-1 step structure)
                    import warnings
                    warnings.warn(f"{why}; using the post-graph tail")
                    self._collective_in_graph = False
                    torch.cuda.synchronize()
                    self._graph = torch.cuda.CUDAGraph()
                    out = record(False)
            else:
                out = record(False)
        self._static_out = out
        if dump:
            try:
                self._graph.debug_dump(dump)
            except Exception as e:                   # (diagnostics only)
                print(f"MMVAE_GRAPH_DUMP: {e}")
        self.flat.zero_grad()
        return out

    def _validate_graph_collective(self, capture_error=None):
        """A captured RCCL all-reduce has to prove itself before the step relies on it (the path had only ever
        run on a one-rank group): ONE replay of the freshly captured graph -- backward, in-graph all-reduce, Adam --
        and then the replicas must still hold bit-identical parameters (min == max of a checksum over the ranks, the
        check bench.py reports as `replicas_in_sync`).  Parameters, optimiser state, noise counters and the step count
        are restored afterwards, so training starts from the same state either way.  Returns None when the graph is
        good, else the reason; every rank reaches the same verdict (the failure flag is reduced with MAX)."""
        dist = torch.distributed
        opt = self.optimizer
        # every tensor the replay advances: parameters, optimiser state and step count, and the model's buffers -- noise
        # and dropout counters (`_rng_state`, DropoutState.state), BatchNorm running statistics / num_batches_tracked of a
        # ResNet tower (those were left advanced, so the first batch was counted twice)
        state = tuple(t for t in (self.flat.data, opt.m, opt.v, getattr(opt, "vmax", None), opt.step_dev) if t is not None)
        state += tuple(b for b in self.model.buffers() if b.is_cuda)
        keep = [t.clone() for t in state]
        bad, why = 0.0, None
        # Vote BEFORE anyone replays: a rank whose capture raised has no graph, and a replay on the other
        # ranks would put gradient-bucket all-reduces on the communicator against this rank's one-element verdict
        # reduce (mismatched collectives: a hang or garbage).  So the capture outcome is reduced with MAX first; only
        # when every rank captured does any rank replay.
        failed = torch.tensor([0.0 if capture_error is None else 1.0], device=self.flat.data.device)
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if float(failed.item()) != 0.0:
            return capture_error or "another rank could not capture the all-reduce"
        try:
            self._graph.replay()
            torch.cuda.synchronize()
        except RuntimeError as e:
            bad, why = 1.0, f"replay of the graph with the captured all-reduce failed ({e})"
        cs = self.flat.data.double().sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        flag = torch.tensor([bad + float(not bool(torch.isfinite(cs).all())) + float(lo.item() != hi.item())],
                            device=cs.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        for t, k in zip(state, keep):
            t.copy_(k)
        self.flat.zero_grad()
        torch.cuda.synchronize()
        if float(flag.item()) != 0.0:
            return why or "replicas diverged after one replay of the graph with the captured all-reduce"
        return None

    def _fwd_bwd(self, batch):
        """forward + backward of the objective; mixers whose loss is linear in per-tower terms provide
        `objective_backward`, which seeds every tower's backward without joining the towers first"""
        if hasattr(self.model, "objective_backward"):
            return self.model.objective_backward(batch)
        out = self.model.objective(batch)
        out["loss"].backward(self._one)
        return out

    def _finish_step(self):
        if hasattr(self.model, "finish_step"):
            self.model.finish_step()

    def load_batch(self, batch):
        for k, v in batch.items():
            for kk in ("data", "masks"):
                if v[kk] is not None:
                    self._static_batch[k][kk].copy_(v[kk], non_blocking=True)

    def load_batch_compact(self, compact):
        """Like load_batch, from the compact host representation: per modality either
        `{"u8": uint8 tensor}` (image bytes, expanded to fp32 / 255 on the device) or `{"tokens": (B,T) int32,
        "lengths": (B,) int32}` (expanded to the one-hot tensor and the mask), a quarter of the PCIe traffic of the
        reference's fp32 batch; anything else is treated as load_batch would."""
        from .. import ops
        for k, v in compact.items():
            dst = self._static_batch[k]
            if "u8" in v:
                src = v["u8"].to(dst["data"].device, non_blocking=True)
                ops.expand_image_u8(src.contiguous(), dst["data"])
            elif "tokens" in v:
                dev = dst["data"].device
                tok = v["tokens"].to(dev, non_blocking=True).contiguous()
                ln = v["lengths"].to(dev, non_blocking=True).contiguous()
                m = dst["masks"]
                mu8 = m.view(torch.uint8) if m is not None and m.dtype == torch.bool else m
                ops.expand_text_tokens(tok, ln, dst["data"], mu8)
            else:
                self.load_batch({k: v})

    @staticmethod
    def pack_compact_pinned(compact):
        """Re-house a compact host batch in ONE pinned byte buffer (every tensor becomes a 16-byte aligned view of it):
        `prefetch_compact` then moves the whole batch with a single H2D copy.  (tools/probe/input_pipeline_time.py, cfg2:
        replay 422 us/step, + expansion 436, + three copies 483, + one packed copy 474.)"""
        items = [(k, name, t.contiguous()) for k, v in compact.items() for name, t in v.items()]
        offs, total = [], 0
        for _, _, t in items:
            offs.append(total)
            total += (t.numel() * t.element_size() + 15) // 16 * 16
        buf = torch.empty(total, dtype=torch.uint8).pin_memory()
        out = {}
        for (k, name, t), o in zip(items, offs):
            view = buf[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
            view.copy_(t)
            out.setdefault(k, {})[name] = view
        out["_packed"] = buf
        return out

    def input_pipe(self, packed):
        """The native form of prefetch_compact / commit_prefetched for batches made by `pack_compact_pinned` (all of
        one layout, given by `packed`): InputPipe.prefetch(batch0) once, then InputPipe.step(next_batch) before every
        replay -- ONE library call per step (csrc/input_pipe.hip)."""
        return InputPipe(self, packed)

    def prefetch_compact(self, compact):
        """Start the host -> device copy of the NEXT batch (compact host format, pinned tensors) on a copy stream into
        staging buffers; it runs under the current step.  `commit_prefetched()` then expands the staged bytes into the
        captured step's static inputs (two HBM-bound launches, ~5 us at batch 128) right before the next replay.
        A batch made by `pack_compact_pinned` travels as one copy."""
        dev = self.flat.data.device
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
            self._staging, self._staged_evt = {}, torch.cuda.Event()
        cs = self._copy_stream
        cs.wait_stream(torch.cuda.current_stream(dev))     # the previous commit has finished reading the staging buffers
        packed = compact.get("_packed")
        with torch.cuda.stream(cs):
            if packed is not None:
                # the device views below are cut from the FIRST packed batch's layout: a later batch of the same size
                # but another layout (offsets, shapes, dtypes) must rebuild them, not be expanded through stale offsets
                sig = _packed_signature(compact)
                dbuf = getattr(self, "_staging_packed", None)
                if dbuf is None or dbuf.numel() != packed.numel() or getattr(self, "_staging_sig", None) != sig:
                    dbuf = self._staging_packed = torch.empty(packed.numel(), dtype=torch.uint8, device=dev)
                    self._staging = {}
                    self._staging_sig = sig
                dbuf.copy_(packed, non_blocking=True)
                if not self._staging:       # device views with the host buffer's offsets, built once
                    base = packed.data_ptr()
                    for k, v in compact.items():
                        if k == "_packed":
                            continue
                        for name, t in v.items():
                            o = t.data_ptr() - base
                            self._staging.setdefault(k, {})[name] = \
                                dbuf[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
            else:
                for k, v in compact.items():
                    slot = self._staging.setdefault(k, {})
                    for name, t in v.items():
                        if name not in slot or slot[name].shape != t.shape or slot[name].dtype != t.dtype:
                            slot[name] = torch.empty(t.shape, dtype=t.dtype, device=dev)
                        slot[name].copy_(t, non_blocking=True)
            self._staged_evt.record(cs)

    def commit_prefetched(self):
        """expand the staged compact batch into the static input buffers (on the current stream, after the copy)"""
        from .. import ops
        torch.cuda.current_stream(self.flat.data.device).wait_event(self._staged_evt)
        for k, slot in self._staging.items():
            dst = self._static_batch[k]
            if "u8" in slot:
                ops.expand_image_u8(slot["u8"], dst["data"])
            elif "tokens" in slot:
                m = dst["masks"]
                mu8 = m.view(torch.uint8) if m is not None and m.dtype == torch.bool else m
                ops.expand_text_tokens(slot["tokens"], slot["lengths"], dst["data"], mu8)

    def fused_step(self, world_size=1):
        """one optimisation step on the static batch: graph replay -> (all-reduce) -> fused Adam"""
        assert world_size == 1 or world_size == self.dp_world or self.dp_force_collective, \
            f"fused_step(world_size={world_size}) on a trainer set up for {self.dp_world} rank(s)"
        out = self._static_out
        self._graph.replay()
        if self._adam_in_graph:
            assert world_size == 1, "captured with the optimiser step inside the graph"
            return out
        if self._collective_in_graph:
            return out
        if self._graph2 is not None:
            # decoders' half of the flat gradients on the wire while the encoders' backward runs
            g, k = self.flat.grad, self.flat.split
            w_dec = torch.distributed.all_reduce(g[k:], async_op=True) if world_size > 1 else None
            self._graph2.replay()
            w_enc = torch.distributed.all_reduce(g[:k], async_op=True) if world_size > 1 else None
            for w in (w_dec, w_enc):
                if w is not None:
                    w.wait()                                   # stream-level wait, the host does not block
            self.optimizer.step()
            return self._static_out
        # ONE RCCL collective over the flat gradient buffer, then the Adam kernel with the 1/world mean folded in
        parallel.reduce_gradients_and_step(self.flat.grad, self.optimizer, self.dp_world, None, self.dp_force_collective)
        return out



class GraphInputRing:
    """The input step INSIDE the captured training step (round 3).  A ring of pinned compact host
    batches (`MultimodalVAE.pack_compact_pinned`, one layout) is registered once; the graph then holds
      head:  the two expansion launches, staging buffer -> the step's static fp32 inputs (mmvae_expand_*);
      tail:  the pull of the NEXT ring slot over the host link into the staging buffer (mmvae_input_ring_pull: source =
             slot `count % n` of a device-resident table of the pinned batches, the launch advances `count`), queued on the
             side (text) tower's stream behind its last backward launch (GradReducer.side_tail; at the end-of-backward
             join for models without that point).
    Replay j trains on ring[j % n]; the host makes NO runtime call per step besides the graph launch (the native pipe is
    one library call = ~7 runtime calls around every launch).  A loader refills slots behind the device: `consumed()`.
    Measured (cfg2, same box): bare replay 0.410, this 0.450-0.455, native pipe 0.485 ms/step.  Three other layouts were
    measured and dropped (tools/probe/ring_time.py): the expansion launches beside the optimiser on the side stream (two
    more cross-stream edges: 0.470), the pull in two halves at the text stream's two idle windows (no gain), and pull +
    expansion as ONE node writing the other of two input-buffer sets with the step captured twice, ping / pong (0.473:
    alternating two graph executables costs 20 us per step by itself).
    Round 6: the pull needs ~70 us for its 1.6 MB (device-initiated reads over the host link) -- more than any idle window
    of either stream -- so it is queued in TWO parts: the first MMVAE_RING_SPLIT per cent (50) behind the backward pass of the
    decoder that does not share the fusion's stream (ops.StreamIdlePoint: that stream idles until the fusion's backward),
    the rest at the tail as before: 0.4247 -> 0.4160 ms (25 / 35 / 50 / 65 %: 0.420 / 0.417 / 0.416 / 0.415; the WHOLE pull
    at the early point: 0.44 - 0.51, it runs into the fusion's backward)."""

    def __init__(self, trainer, ring):
        assert len(ring) >= 1
        sig = _packed_signature(ring[0])
        for b in ring:
            assert b["_packed"].is_pinned() and _packed_signature(b) == sig, "ring slots: pinned, one layout"
        self.ring = list(ring)                       # keeps the pinned buffers alive
        dev = trainer.flat.data.device
        self.dev = dev
        buf = ring[0]["_packed"]
        self.bytes = buf.numel()
        assert self.bytes % 16 == 0 and all(b["_packed"].data_ptr() % 16 == 0 for b in ring), "16-byte aligned slots"
        self.staging = torch.empty(self.bytes, dtype=torch.uint8, device=dev)
        self.table = torch.tensor([b["_packed"].data_ptr() for b in ring], dtype=torch.int64, device=dev)
        self.ctr = torch.zeros(2, dtype=torch.int32, device=dev)
        base = buf.data_ptr()
        self.mods = []
        for k, v in ring[0].items():
            if k == "_packed":
                continue
            dst = trainer._static_batch[k]
            if "u8" in v:
                o = v["u8"].data_ptr() - base
                self.mods.append(("u8", self.staging[o:o + v["u8"].numel()], dst["data"]))
            elif "tokens" in v:
                o, ol = v["tokens"].data_ptr() - base, v["lengths"].data_ptr() - base
                tok = self.staging[o:o + v["tokens"].numel() * 4].view(torch.int32).view(v["tokens"].shape)
                ln = self.staging[ol:ol + v["lengths"].numel() * 4].view(torch.int32)
                m = dst["masks"]
                self.mods.append(("tokens", tok, ln, dst["data"], m.view(torch.uint8) if m is not None and m.dtype == torch.bool else m))
            else:
                raise ValueError(f"modality {k}: the in-graph input step moves 'u8' images and 'tokens' text")
        self._keep = trainer._static_batch
        self.reprime()

    def reprime(self):
        """counter 0, slot 0 in the staging buffer (count becomes 1): the state the recorded step starts from"""
        self.ctr.zero_()
        self.done = False
        self.pull_next(self.dev, [])
        torch.cuda.synchronize()

    def expand(self):
        from .. import ops
        for m in self.mods:
            if m[0] == "u8":
                ops.expand_image_u8(m[1], m[2])
            else:
                ops.expand_text_tokens(m[1], m[2], m[3], m[4])

    split = False
    split_pct = 50
    half_done = False

    def _first_bytes(self):
        return self.bytes * self.split_pct // 100 // 16 * 16

    def pull_first_half(self, device):
        """the first half of the next slot, without advancing the slot counter (ops.StreamIdlePoint: on the stream of the
        decoder that finishes its backward pass early)"""
        if self.done or self.half_done:
            return
        self.half_done = True
        h = self._first_bytes()
        rc = H.lib().mmvae_input_ring_pull(self.table.data_ptr(), len(self.ring), self.ctr.data_ptr(),
                                           self.staging.data_ptr(), 0, h, 0, H.stream())
        if rc:
            raise RuntimeError(f"mmvae_input_ring_pull: {rc}")

    def pull_next(self, device, side_streams):
        """queue the pull of the next slot (once per recorded step): on the current stream -- the side tower's when its
        last backward launch calls this --, or on the last of `side_streams`; only the second half when pull_first_half ran"""
        if self.done:
            return
        self.done = True
        st = side_streams[-1] if side_streams else None
        off = self._first_bytes() if self.half_done else 0
        self.half_done = False
        with torch.cuda.stream(st):
            rc = H.lib().mmvae_input_ring_pull(self.table.data_ptr(), len(self.ring), self.ctr.data_ptr(),
                                               self.staging.data_ptr(), off, self.bytes - off, 1, H.stream())
        if rc:
            raise RuntimeError(f"mmvae_input_ring_pull: {rc}")

    def consumed(self):
        """number of slots the device has pulled so far (host synchronisation: a loader's flow control)"""
        return int(self.ctr[0].item())


def _packed_signature(packed):
    """layout of a pack_compact_pinned() batch: (key, name, byte offset, shape, dtype) of every tensor in the buffer"""
    base = packed["_packed"].data_ptr()
    return tuple((k, name, t.data_ptr() - base, tuple(t.shape), str(t.dtype))
                 for k, v in packed.items() if k != "_packed" for name, t in v.items())


class InputPipe:
    """Host side of csrc/input_pipe.hip: the packed pinned batch -> (copy stream) staging -> static inputs of the
    captured step.  `step(next)` = wait for the staged batch, expand it, start copying `next` (or nothing for None)."""

    def __init__(self, trainer, packed):
        self._H = H
        buf = packed["_packed"]
        assert buf.is_pinned(), "pack_compact_pinned() batches only"
        dev = trainer.flat.data.device
        self.bytes = buf.numel()
        self.staging = torch.empty(self.bytes, dtype=torch.uint8, device=dev)
        base = buf.data_ptr()
        mods = []
        for k, v in packed.items():
            if k == "_packed":
                continue
            dst = trainer._static_batch[k]
            m = H.InputMod()
            if "u8" in v:
                m.kind, m.src_off, m.dst, m.n = H.INPUT_IMAGE_U8, v["u8"].data_ptr() - base, dst["data"].data_ptr(), v["u8"].numel()
                assert dst["data"].numel() == v["u8"].numel() and dst["data"].is_contiguous()
            elif "tokens" in v:
                Bt, Tt = v["tokens"].shape
                assert v["tokens"].dtype == torch.int32 and v["lengths"].dtype == torch.int32
                assert tuple(dst["data"].shape[:2]) == (Bt, Tt) and dst["data"].is_contiguous()
                m.kind, m.src_off, m.len_off = H.INPUT_TEXT_TOKENS, v["tokens"].data_ptr() - base, v["lengths"].data_ptr() - base
                m.dst, m.B, m.T, m.V = dst["data"].data_ptr(), Bt, Tt, dst["data"].shape[2]
                mk = dst["masks"]
                m.mask = mk.data_ptr() if mk is not None else None      # bool and uint8 masks are both one byte per token
            else:
                raise ValueError(f"modality {k}: the native pipe moves 'u8' images and 'tokens' text")
            mods.append(m)
        assert 0 < len(mods) <= H.INPUT_MAX_MODS
        self._mods = (H.InputMod * len(mods))(*mods)
        self._sig = _packed_signature(packed)   # every later batch must have exactly this layout
        self._checked = {}
        self._checked_max = 16                  # ring sizes in use are 2-4; a dropped entry is simply re-validated
        self._keep = trainer._static_batch      # the raw pointers above point into these tensors
        h = ctypes.c_void_p()
        rc = H.lib().mmvae_input_pipe_create(ctypes.byref(h), self.staging.data_ptr(), self.bytes)
        if rc:
            raise RuntimeError(f"mmvae_input_pipe_create: {rc}")
        self._h = h

    def _ptr(self, packed):
        buf = packed["_packed"]
        ptr = buf.data_ptr()
        # (a ring of pinned batches: each buffer's layout is checked once -- and the checked buffer is HELD, so that its
        # address cannot come back as another allocation with other offsets: ADVICE r3)
        if self._checked.get(ptr) is not buf:
            assert buf.numel() == self.bytes and buf.is_pinned()
            if _packed_signature(packed) != self._sig:
                raise ValueError("InputPipe: this packed batch has another layout (offsets / shapes / dtypes) than the "
                                 "one the pipe was created for")
            self._checked[ptr] = buf
            while len(self._checked) > self._checked_max:      # a fresh pinned batch per step must not pin host memory
                self._checked.pop(next(iter(self._checked)))   # without bound: oldest validated buffer out
        return ptr

    def prefetch(self, packed):
        rc = self._H.lib().mmvae_input_pipe_prefetch(self._h, self._ptr(packed))
        if rc:
            raise RuntimeError(f"mmvae_input_pipe_prefetch: {rc}")

    def step(self, next_packed=None):
        rc = self._H.lib().mmvae_input_pipe_commit(self._h, self._mods, len(self._mods),
                                                   self._ptr(next_packed) if next_packed is not None else None,
                                                   self._H.stream())
        if rc:
            raise RuntimeError(f"mmvae_input_pipe_commit: {rc}")

    def close(self):
        if self._h is not None:
            self._H.lib().mmvae_input_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
