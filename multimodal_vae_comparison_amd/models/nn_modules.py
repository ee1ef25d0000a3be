"""Building blocks shared by the towers (host side).  Only what the hot path needs from the reference's
models/nn_modules.py: PositionalEncoding (nn_modules.py:418-438)."""
import math

import torch
import torch.nn as nn

from .. import hipops as H
from .. import ops


class ModuleWrap(nn.Module):
    """Keeps the reference's state_dict key layout: it wraps single layers in nn.DataParallel
    (models/decoders.py:58-69, encoders.py:825-826), which adds a `.module.` segment to every key.  This shim
    adds the same segment without any scatter/gather (one process drives one GPU here)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def positional_table(d_model, max_len=1000):
    """The `pe` buffer of the reference's PositionalEncoding, (max_len, 1, d_model); nn_modules.py:422-429."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).transpose(0, 1).contiguous()


class PositionalEncoding(nn.Module):
    """Holds the sin/cos table (host-computed once, as the reference does); the add itself is fused into the
    consumers (ops.embed_pe for the text encoder, a cached (T,B,D) query tensor for the text decoder).
    Dropout p is kept for train mode (applied by the consumer)."""

    def __init__(self, d_model, dropout=0.1, max_len=1000):
        super().__init__()
        self.p = dropout
        self.d_model = d_model
        self.register_buffer("pe", positional_table(d_model, max_len))


class DropoutState(nn.Module):
    """Per-tower counter-based dropout state (include/mmvae_hip.h: mmvae_dropout_t): int32 buffer
    [seed, counter, slot_0 .. slot_15].  `begin()` (once per tower forward in train mode) bumps the counter on
    the device and parks its value in the slot of this call, so that the backward kernels of the same call -- which
    may run after later calls of the same tower (PoE) -- regenerate identical masks."""
    _next_seed = [0x1234567]

    def __init__(self):
        super().__init__()
        st = torch.zeros(2 + H.DROPOUT_SLOTS, dtype=torch.int32)
        DropoutState._next_seed[0] = (DropoutState._next_seed[0] * 1103515245 + 12345) & 0x7FFFFFFF
        st[0] = (torch.initial_seed() ^ DropoutState._next_seed[0]) & 0x7FFFFFFF
        self.register_buffer("state", st, persistent=False)
        self.call = 0
        self.pre = False      # call 0 of this step was advanced by another tower's launch: begin() launches nothing
        self.lead = None      # encoder state: the decoder states whose call 0 rides on this state's first advance
        self.prefix = ""

    def reset_calls(self):
        self.call = 0
        self.pre = False
        self.lead = None

    @staticmethod
    def link(enc_states, dec_states):
        """Start of a training step: the first encoder tower that draws a mask also advances call 0 of every DECODER
        tower's state in the same launch (one graph node less per decoder, nothing at the head of the decoders' chains).
        Safe across streams because every mixer joins all encoder streams at the fusion before any decoder starts;
        the other encoders, which may run beside the leading one, advance their own."""
        dec_states = [s for s in dec_states if s.state.is_cuda]
        if not enc_states or not dec_states or len(dec_states) + 1 > H.DROPOUT_ADVANCE_MAX:
            return
        for s in enc_states:
            s.lead = dec_states      # whichever encoder comes first takes them (the list is emptied then)

    def begin(self):
        slot = self.call % H.DROPOUT_SLOTS
        self.call += 1
        if self.pre and self.call == 1:
            self.pre = False
        elif self.call == 1 and self.lead:
            followers = [s for s in self.lead if s is not self and s.call == 0 and not s.pre]
            del self.lead[:]
            ops.dropout_advance_many([self.state] + [s.state for s in followers])
            for s in followers:
                s.pre = True
        else:
            ops.dropout_advance(self.state, slot)
        return slot, self.call - 1

    def spec(self, slot, call, site, p, name):
        return ops.DropSpec(self.state, slot, site, p, f"{self.prefix}.{name}#{call}")


class HipLinear(nn.Module):
    """nn.Linear parameters (torch default init) driven by the MFMA GEMM; `in_act` is applied to the input."""

    def __init__(self, in_features, out_features, in_act=H.ACT_NONE):
        super().__init__()
        ref = nn.Linear(in_features, out_features)
        self.weight, self.bias = ref.weight, ref.bias
        self.in_features, self.out_features, self.in_act = in_features, out_features, in_act

    def flat_groups(self):
        return [[self.weight, self.bias]]

    def forward(self, x, in_act=None, out_ep=H.EP_NONE, lazy=False, res_sink=None):
        """res_sink: ops.ResidualGrad of a residual connection around the block this layer opens (its gradient joins this
        layer's data-gradient launch instead of an addition of its own)"""
        return ops.linear(x, self.weight, self.bias, self.in_act if in_act is None else in_act, self.weight.grad,
                          self.bias.grad, out_ep, res_sink=res_sink, lazy=lazy)


class HipLayerNorm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))

    def flat_groups(self):
        return [[self.weight, self.bias]]

    def forward(self, x, residual=None, drop=None, res_sink=None):
        """LayerNorm(dropout(x) + residual); res_sink: ops.ResidualGrad shared with the op that made x from `residual`"""
        gg, gb = self.weight.grad, self.bias.grad
        if gg is not None and gb is not None and gb.data_ptr() != gg.data_ptr() + 4 * gg.numel():
            gg = gb = None      # not laid out adjacently: let autograd accumulate
        return ops.layernorm_residual(x, residual, self.weight, self.bias, gg, gb, drop, res_sink)


class HipSelfAttention(nn.Module):
    """nn.MultiheadAttention parameter layout (in_proj_weight/in_proj_bias/out_proj.{weight,bias}), xavier-uniform
    in_proj like torch; forward = packed QKV GEMM -> masked softmax attention kernel -> out_proj GEMM."""

    def __init__(self, d, nhead):
        super().__init__()
        ref = nn.MultiheadAttention(d, nhead)
        self.in_proj_weight, self.in_proj_bias = ref.in_proj_weight, ref.in_proj_bias
        self.out_proj = HipLinear(d, d)
        with torch.no_grad():
            self.out_proj.weight.copy_(ref.out_proj.weight)
            self.out_proj.bias.copy_(ref.out_proj.bias)
        self.d, self.nhead = d, nhead

    def flat_groups(self):
        return [[self.in_proj_weight, self.in_proj_bias]]

    def lazy_out(self, x):
        """may the caller's LayerNorm compute out_proj in its own launch (ops.PROJ32_LN: d_model 32 on the device)?"""
        return ops.PROJ32_LN and self.d == 32 and x.is_cuda and x.dtype == torch.float32

    def forward(self, x, mask_u8, drop=None, res_sink=None, lazy_out=False):
        """mask_u8: (N, L) validity bytes (1 = real token); padded keys are ignored; `drop`: attention-weight dropout;
        res_sink: ops.ResidualGrad of the residual connection around this block (its gradient joins the in-projection's);
        lazy_out: the out-projection is left to the LayerNorm that consumes the result (ops.linear(..., lazy=True))"""
        qkv = ops.linear(x, self.in_proj_weight, self.in_proj_bias, H.ACT_NONE, self.in_proj_weight.grad,
                         self.in_proj_bias.grad, res_sink=res_sink)
        a = ops.attention(qkv, mask_u8, self.nhead, mask_is_valid=True, drop=drop)
        return self.out_proj(a, lazy=lazy_out)

    def value_path(self, mem, L=None, drop=None, lazy_out=False):
        """Cross-attention over a length-1 memory: softmax over one key == 1, so the output is
        out_proj(v_proj(mem)) for every query (q/k projections receive exactly zero gradient).  In train mode the
        attention-weight dropout acts on that single weight per (query, head): the value row is replicated over the
        L queries with a per-(sample, head, query) mask before out_proj."""
        d = self.d
        gw, gb = self.in_proj_weight.grad, self.in_proj_bias.grad
        v = ops.linear(mem, self.in_proj_weight[2 * d:], self.in_proj_bias[2 * d:], H.ACT_NONE,
                       gw[2 * d:] if gw is not None else None, gb[2 * d:] if gb is not None else None)
        if drop is not None:
            v = ops.head_bcast_dropout(v, L, self.nhead, drop)        # (L, N, d)
            return self.out_proj(v, lazy=lazy_out)
        return self.out_proj(v)
