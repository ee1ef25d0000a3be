"""One flat fp32 parameter buffer + one flat gradient buffer + fused Adam(amsgrad) (host plumbing).

Why: the model has ~180 small tensors (0.99 M parameters).  Laid out contiguously they need ONE Adam kernel,
ONE RCCL all-reduce (3.95 MB) and no per-tensor AccumulateGrad kernels: the weight-gradient kernels
accumulate straight into views of `grad` (ops.py).  Layers that want several tensors adjacent (weight|bias,
the mu|logvar head pair) declare `flat_groups()`; everything is 16-byte aligned.
"""
import torch
import torch.nn as nn

from . import ops


def _channels_last(p):
    """a (Cout, Cin, kh, kw) parameter whose memory is (Cout, kh, kw, Cin) (the ResNet tower's k x k weights)"""
    return p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last)


class FlatParams:
    @staticmethod
    def view_of(buf, o, p):
        """the slice of a flat buffer that belongs to parameter p, with p's shape AND memory layout"""
        if _channels_last(p):
            co, ci, kh, kw = p.shape
            return buf[o:o + p.numel()].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return buf[o:o + p.numel()].view(p.shape)

    @staticmethod
    def flatten_like(t, p):
        """a tensor of p's shape as the flat run of elements in p's memory order"""
        return t.permute(0, 2, 3, 1).reshape(-1) if _channels_last(p) else t.reshape(-1)

    def __init__(self, module: nn.Module):
        params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        by_id = {id(p): (n, p) for n, p in params}
        order, seen = [], set()
        for m in module.modules():
            groups = m.flat_groups() if hasattr(m, "flat_groups") else []
            for g in groups:
                g = [p for p in g if p is not None and id(p) in by_id and id(p) not in seen]
                if g:
                    order.append(g)
                    seen.update(id(p) for p in g)
        for n, p in params:
            if id(p) not in seen:
                order.append([p])
                seen.add(id(p))
        # encoder parameters first, everything else (decoders, prior) behind them: the second range's gradients are
        # final when the fusion backward starts, so data parallelism can all-reduce it under the encoders' backward
        # (trainer.capture, world_size > 1).  `split` = first element of the second range.
        is_enc = lambda g: ".enc." in ("." + by_id[id(g[0])][0])
        order = [g for g in order if is_enc(g)] + [g for g in order if not is_enc(g)]
        offs, total = [], 0
        self.split = None
        for g in order:
            total = (total + 3) // 4 * 4           # 16-byte alignment of every group
            if self.split is None and not is_enc(g):
                self.split = total
            for p in g:
                offs.append((p, total))
                total += p.numel()
        total = (total + 3) // 4 * 4
        if self.split is None:
            self.split = total
        dev = params[0][1].device
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.n_params = sum(p.numel() for _, p in params)
        self.index = {}
        for p, o in offs:
            self.data[o:o + p.numel()].copy_(self.flatten_like(p.data, p))
            view = self.view_of(self.data, o, p)
            grad = self.view_of(self.grad, o, p)
            p.data = view
            p.grad = grad
            self.index[by_id[id(p)][0]] = (o, tuple(p.shape))
        self.params = [p for p, _ in offs]
        self.offset_of = {id(p): o for p, o in offs}
        self.params_in_model_order = [p for _, p in params]

    def rebind_grads(self):
        """(re)attach the flat gradient views (after an optimizer.zero_grad(set_to_none=True))"""
        for name_off, p in zip(self.index.values(), self.params):
            o, shape = name_off
            p.grad = self.view_of(self.grad, o, p)

    def zero_grad(self):
        ops.fill(self.grad, 0.0)


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, amsgrad=True) (reference: models/trainer.py:79-81)
    as one kernel over the flat buffer.  The step count lives on the device so that a captured hipGraph
    replays with the correct bias correction."""

    supports_fold = True       # FlatAdam.step() can take the end-of-backward fold (GradReducer.deferred) into its launch

    def __init__(self, flat: FlatParams, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        super().__init__(flat.params, dict(lr=lr, betas=betas, eps=eps, amsgrad=True))
        self.flat = flat
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.vmax = torch.zeros_like(flat.data)
        self.step_dev = torch.zeros(6, dtype=torch.int32, device=flat.data.device)   # {count, ticket, b1^count, b2^count}
        self.grad_scale = grad_scale
        self._steps_since_check = 0
        self._early = None        # (lo, hi) of a range this step's step_early() has already updated

    def check_grad_views(self):
        """every parameter's .grad must still be its slice of the flat gradient buffer (a zero_grad(set_to_none=True),
        a model.to() or .half() detaches them: autograd would then fill fresh tensors while this optimiser keeps
        reading an all-zero flat buffer and silently updates nothing)"""
        base = self.flat.grad.data_ptr()
        for p in self.flat.params:
            g = p.grad
            o = self.flat.offset_of[id(p)]
            if g is None or g.data_ptr() != base + 4 * o or p.data_ptr() != self.flat.data.data_ptr() + 4 * o:
                raise RuntimeError("FlatAdam: a parameter (or its .grad) is no longer a view of the flat buffers; use "
                                   "FlatAdam.zero_grad() / MultimodalVAE.zero_grad() (not set_to_none) and do not move "
                                   "or cast the model after construction")

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._steps_since_check == 0:          # first step and every 256th (180 pointer reads: not per step)
            self.check_grad_views()
        self._steps_since_check = (self._steps_since_check + 1) & 255
        g = self.param_groups[0]
        d, ops.GradReducer.deferred = ops.GradReducer.deferred, None
        early, self._early = self._early, None
        if early is not None:
            # the second range [lo, n) of this step was updated behind the decoders' backward (step_early): the closing
            # launch covers [0, lo) with the segments that are left
            lo, hi = early
            assert hi == self.flat.data.numel() and 0 < lo < hi
            ops.adam_fold_range(self.flat.data, self.flat.grad, self.m, self.v, self.vmax, 0, lo, True, float(g["lr"]),
                                g["betas"][0], g["betas"][1], g["eps"], self.step_dev, self.grad_scale, True,
                                d["table"] if d is not None else None, tail=d["tail"] if d is not None else None)
            return loss
        if d is not None:
            # the backward pass left its split partials unfolded for us (GradReducer.defer_next): fold + update at once
            ops.adam_fold_flat(self.flat.data, self.flat.grad, self.m, self.v, self.vmax, float(g["lr"]), g["betas"][0],
                               g["betas"][1], g["eps"], self.step_dev, self.grad_scale, True, d)
            return loss
        # step = -1: the kernel takes step_dev[0] + 1 and stores it itself (one launch per step)
        ops.adam_amsgrad_flat(self.flat.data, self.flat.grad, self.m, self.v, self.vmax, float(g["lr"]),
                              g["betas"][0], g["betas"][1], g["eps"], -1, self.step_dev, self.grad_scale, True)
        return loss

    @torch.no_grad()
    def step_early(self, lo, hi, table):
        """elements [lo, hi) of this step NOW, on the current stream, without closing the step (ops.GradReducer.early_step:
        the decoders' + prior's gradients are final when the fusion's backward has run); step() then covers the rest"""
        assert self._early is None and self.supports_fold
        g = self.param_groups[0]
        ops.adam_fold_range(self.flat.data, self.flat.grad, self.m, self.v, self.vmax, lo, hi, False, float(g["lr"]),
                            g["betas"][0], g["betas"][1], g["eps"], self.step_dev, self.grad_scale, True, table)
        self._early = (lo, hi)

    def zero_grad(self, set_to_none=False):
        # step() already clears the buffer inside the Adam kernel; an explicit zero_grad() (Lightning calls it
        # every step) keeps torch semantics: clear, and keep the flat views attached.
        self.flat.rebind_grads()
        self.flat.zero_grad()

    # ---- torch.optim.Adam-compatible state (what a Lightning checkpoint stores under "optimizer_states") ----------
    def state_dict(self):
        """torch.optim.Adam(amsgrad=True).state_dict() layout over the model's parameters in `named_parameters()` order
        (the order the reference's `Adam(filter(requires_grad, self.parameters()))` enumerates them in)"""
        step = int(self.step_dev[0].item())
        state = {}
        for i, p in enumerate(self.flat.params_in_model_order):
            o = self.flat.offset_of[id(p)]
            sl = slice(o, o + p.numel())
            vo = lambda buf: FlatParams.view_of(buf, o, p).clone()
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": vo(self.m), "exp_avg_sq": vo(self.v),
                        "max_exp_avg_sq": vo(self.vmax)}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": True,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(state)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """torch.optim.Adam only creates state for parameters that have received a gradient: a reference checkpoint of
        `mixing: moe, obj: elbo` has no entry for the model-level prior `_pz_params.1` (requires_grad, never touched by
        MOE.objective; mmvae_base.py:37).  So the PARAMETER LIST (`param_groups[0]["params"]`) must match the model;
        state entries may be missing (their moments stay zero), present ones must match in size."""
        params = self.flat.params_in_model_order
        steps = set()
        listed = sd["param_groups"][0].get("params")
        if listed is not None and len(listed) != len(params):
            raise RuntimeError(f"optimizer param group lists {len(listed)} parameters, the model has {len(params)} "
                               f"trainable parameters")
        extra = [i for i in sd["state"] if not (isinstance(i, int) and 0 <= i < len(params))]
        if extra:
            raise RuntimeError(f"optimizer state has entries for unknown parameter indices {extra[:5]}")
        self.m.zero_()
        self.v.zero_()
        self.vmax.zero_()
        for i, p in enumerate(params):
            st = sd["state"].get(i)
            if st is None:
                continue
            if st["exp_avg"].numel() != p.numel():
                raise RuntimeError(f"optimizer state {i}: exp_avg has {st['exp_avg'].numel()} elements, parameter "
                                   f"{tuple(p.shape)} has {p.numel()}")
            o = self.flat.offset_of[id(p)]
            sl = slice(o, o + p.numel())
            fl = lambda t: FlatParams.flatten_like(t.reshape(p.shape), p)
            self.m[sl].copy_(fl(st["exp_avg"]))
            self.v[sl].copy_(fl(st["exp_avg_sq"]))
            if "max_exp_avg_sq" in st:
                self.vmax[sl].copy_(fl(st["max_exp_avg_sq"]))
            steps.add(int(float(st["step"])))
        assert len(steps) <= 1, f"per-parameter step counts differ: {steps}"
        self.step_dev.zero_()                     # the running beta powers are rebuilt by the kernel (pow()) once
        if steps:
            self.step_dev[0] = steps.pop()
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = g["lr"]
        self.param_groups[0]["betas"] = tuple(g["betas"])
        self.param_groups[0]["eps"] = g["eps"]


class FlatAdaBelief(FlatAdam):
    """adabelief_pytorch.AdaBelief(lr, eps=1e-16, betas=(0.9, 0.999), weight_decouple=True, rectify=False) -- what the
    reference's `optimizer: adabelief` builds (models/trainer.py:82-86) -- as one kernel over the flat buffers.  The package
    is not vendored by the reference and absent in this image: the update is restated from the paper (Zhuang et al. 2020,
    Algorithm 2) and the package's update order (csrc/optim.hip: adabelief_update1), PARITY UNPINNED.  State names as the
    package's: exp_avg, exp_avg_var."""

    supports_fold = False      # the end-of-backward fold is fused into the Adam launch only

    def __init__(self, flat: FlatParams, lr=1e-4, betas=(0.9, 0.999), eps=1e-16, grad_scale=1.0):
        super().__init__(flat, lr=lr, betas=betas, eps=eps, grad_scale=grad_scale)
        self.param_groups[0]["amsgrad"] = False
        self.vmax = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._steps_since_check == 0:
            self.check_grad_views()
        self._steps_since_check = (self._steps_since_check + 1) & 255
        assert ops.GradReducer.deferred is None, "FlatAdaBelief does not take a deferred fold"
        g = self.param_groups[0]
        ops.adabelief_flat(self.flat.data, self.flat.grad, self.m, self.v, float(g["lr"]), g["betas"][0], g["betas"][1],
                           g["eps"], -1, self.step_dev, self.grad_scale, True)
        return loss

    def state_dict(self):
        step = int(self.step_dev[0].item())
        state = {}
        for i, p in enumerate(self.flat.params_in_model_order):
            o = self.flat.offset_of[id(p)]
            state[i] = {"step": step, "exp_avg": FlatParams.view_of(self.m, o, p).clone(),
                        "exp_avg_var": FlatParams.view_of(self.v, o, p).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "buffer": [[None, None, None] for _ in range(10)], "params": list(range(len(state)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        params = self.flat.params_in_model_order
        listed = sd["param_groups"][0].get("params")
        if listed is not None and len(listed) != len(params):
            raise RuntimeError(f"optimizer param group lists {len(listed)} parameters, the model has {len(params)}")
        self.m.zero_()
        self.v.zero_()
        steps = set()
        for i, p in enumerate(params):
            st = sd["state"].get(i)
            if st is None:
                continue
            if st["exp_avg"].numel() != p.numel():
                raise RuntimeError(f"optimizer state {i}: exp_avg has {st['exp_avg'].numel()} elements, parameter has {p.numel()}")
            o = self.flat.offset_of[id(p)]
            sl = slice(o, o + p.numel())
            self.m[sl].copy_(FlatParams.flatten_like(st["exp_avg"].reshape(p.shape), p))
            self.v[sl].copy_(FlatParams.flatten_like(st["exp_avg_var"].reshape(p.shape), p))
            steps.add(int(float(st["step"])))
        assert len(steps) <= 1, f"per-parameter step counts differ: {steps}"
        self.step_dev.zero_()
        if steps:
            self.step_dev[0] = steps.pop()
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = g["lr"]
        self.param_groups[0]["betas"] = tuple(g["betas"])
        self.param_groups[0]["eps"] = g["eps"]
