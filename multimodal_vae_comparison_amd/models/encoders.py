"""Encoder towers behind the reference's plugin names (`Enc_<Name>(latent_dim, data_dim, latent_private,
enc_mu_logvar)`, looked up by string in models/vae.py; reference: models/encoders.py).

Same constructor signatures, attributes (latent_dim, data_dim, out_dim) and state_dict key names as the
reference; the arithmetic runs on the gfx950 kernels of the C-ABI library (no torch.nn.functional compute).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import hipops as H
from .. import ops
from .NetworkTypes import NetworkRoles, NetworkTypes
from .nn_modules import DropoutState, HipLayerNorm, HipLinear, HipSelfAttention, ModuleWrap, PositionalEncoding


class VaeComponent(nn.Module):
    """models/encoders.py:15-54"""

    def __init__(self, latent_dim, data_dim, latent_private=None, enc_mu_logvar=True,
                 net_type=NetworkTypes.UNSPECIFIED, net_role=NetworkRoles.UNSPECIFIED):
        super().__init__()
        self.net_role = net_role
        self.latent_dim = latent_dim
        self.enc_mu_logvar = enc_mu_logvar
        self.latent_private = latent_private
        self.out_dim = latent_dim + latent_private if latent_private is not None else latent_dim
        self.data_dim = data_dim
        self.net_type = net_type
        self.mu_layer = None
        self.logvar_layer = None
        self.raw_heads = False      # set by a mixer's objective(): return [mu | raw logvar head], softmax in the PoE kernel

    def init_final_layers(self, in_feats):
        self.mu_layer = HipLinear(in_feats, self.out_dim)
        if self.enc_mu_logvar:
            self.logvar_layer = HipLinear(in_feats, self.out_dim)

    # the two heads are laid out as one (2*out_dim, F) weight so that one GEMM produces [mu | u]
    def _heads(self):
        mu = self.mu_layer.module if isinstance(self.mu_layer, ModuleWrap) else self.mu_layer
        lv = self.logvar_layer.module if isinstance(self.logvar_layer, ModuleWrap) else self.logvar_layer
        return mu, lv

    def flat_groups(self):
        mu, lv = self._heads()
        if mu is None or lv is None:
            return []
        return [[mu.weight, lv.weight], [mu.bias, lv.bias]]

    def process_output(self, data, in_act=H.ACT_NONE):
        """mu = W_mu h + b;  "logvar" = softmax(W_lv h + b, -1) + 1e-6   (models/encoders.py:49-54).
        Returns views of one packed (B, 2*out_dim) tensor; `mu._base` is that tensor (used by the fused
        latent kernel)."""
        mu_l, lv_l = self._heads()
        if not self.enc_mu_logvar:
            return mu_l(data, in_act)
        packed = self.packed_heads() if data.requires_grad else None
        if packed is not None:   # flat layout (flat.py): one (2D, F) GEMM, gradients accumulated in place
            h = ops.linear(data, packed[0], packed[1], in_act, packed[2], packed[3])
        else:
            h = torch.cat([mu_l(data, in_act), lv_l(data, in_act)], dim=-1)
        return self.finish_heads(h)

    def finish_heads(self, h):
        """packed head outputs (B, 2D) -> (mu, "logvar") views (softmax unless the fused latent kernel applies it)"""
        D = self.out_dim
        if not self.raw_heads:
            h = ops.head_softmax(h)
        return h[:, :D], h[:, D:]

    def packed_heads(self):
        """(w (2D, F), b (2D), gw, gb): the mu and logvar heads as ONE linear layer when their parameters and preset
        gradient views are adjacent in the flat buffers (flat.py lays them out that way), else None"""
        if not self.enc_mu_logvar:
            return None
        mu_l, lv_l = self._heads()
        D = self.out_dim
        w_mu, w_lv, b_mu, b_lv = mu_l.weight, lv_l.weight, mu_l.bias, lv_l.bias
        gw_mu, gw_lv, gb_mu, gb_lv = w_mu.grad, w_lv.grad, b_mu.grad, b_lv.grad
        if not (w_lv.data_ptr() == w_mu.data_ptr() + 4 * w_mu.numel()
                and b_lv.data_ptr() == b_mu.data_ptr() + 4 * b_mu.numel()
                and None not in (gw_mu, gw_lv, gb_mu, gb_lv)
                and gw_lv.data_ptr() == gw_mu.data_ptr() + 4 * w_mu.numel()
                and gb_lv.data_ptr() == gb_mu.data_ptr() + 4 * b_mu.numel()):
            return None
        F = w_mu.shape[1]
        return (torch.as_strided(w_mu.detach(), (2 * D, F), (F, 1)), torch.as_strided(b_mu.detach(), (2 * D,), (1,)),
                torch.as_strided(gw_mu, (2 * D, F), (F, 1)), torch.as_strided(gb_mu, (2 * D,), (1,)))


class VaeEncoder(VaeComponent):
    def __init__(self, latent_dim, data_dim, latent_private, enc_mu_logvar, net_type: NetworkTypes):
        super().__init__(latent_dim, data_dim, latent_private, enc_mu_logvar, net_type, NetworkRoles.ENCODER)


class HipConv2d(nn.Module):
    """nn.Conv2d(k=4, s=2, p=1) parameters (torch default init) on the MFMA implicit-GEMM kernels."""

    def __init__(self, cin, cout, in_act=H.ACT_NONE):
        super().__init__()
        ref = nn.Conv2d(cin, cout, 4, stride=2, padding=1)
        self.weight, self.bias = ref.weight, ref.bias
        self.in_act = in_act

    def flat_groups(self):
        return [[self.weight, self.bias]]

    def forward(self, x):
        return ops.conv2d_k4s2(x, self.weight, self.bias, self.in_act, self.weight.grad, self.bias.grad)


class Enc_CNN2(VaeEncoder):
    """models/encoders.py:163-223: 4x [Conv2d(k4,s2,p1) + SiLU] 3->32->32->32->32, flatten, Linear 512->512, heads.
    The SiLU of layer l is applied by layer l+1 while it stages its input (ops.py conventions)."""

    def __init__(self, latent_dim, data_dim, latent_private, enc_mu_logvar):
        data_dim = (3, 64, 64)
        super().__init__(latent_dim, data_dim, latent_private, enc_mu_logvar, net_type=NetworkTypes.CNN)
        hid = 32
        self.hidden_dim = 512
        self.reshape = (hid, 4, 4)
        self.conv1 = HipConv2d(3, hid, H.ACT_NONE)
        self.conv2 = HipConv2d(hid, hid, H.ACT_SILU)
        self.conv3 = HipConv2d(hid, hid, H.ACT_SILU)
        self.conv4 = HipConv2d(hid, hid, H.ACT_SILU)
        self.lin1 = HipLinear(int(np.prod(self.reshape)), self.hidden_dim, H.ACT_SILU)
        self.init_final_layers(self.hidden_dim)

    def forward(self, x):
        if isinstance(x, dict):
            x = x["data"]
        bs = x.size(0)
        u = self.conv4(self.conv3(self.conv2(self.conv1(x.float()))))
        o5 = self.lin1(u.view(bs, -1))
        return self.process_output(o5)


class HipConv(nn.Module):
    """nn.Conv2d / nn.ConvTranspose2d parameters (torch default init) for any (channels, k, stride, pad): ops.conv2d /
    ops.convT2d pick the MFMA kernel when the shape is one of theirs, the generic kernel otherwise."""

    def __init__(self, cin, cout, k, stride, pad, in_act=H.ACT_NONE, transposed=False, out_ep=H.EP_NONE):
        super().__init__()
        ref = (nn.ConvTranspose2d if transposed else nn.Conv2d)(cin, cout, k, stride=stride, padding=pad)
        self.weight, self.bias = ref.weight, ref.bias
        self.cfg = (stride, pad, in_act, transposed, out_ep)

    def flat_groups(self):
        return [[self.weight, self.bias]]

    def forward(self, x, ep_bwd=True):
        stride, pad, in_act, transposed, out_ep = self.cfg
        if transposed:
            return ops.convT2d(x, self.weight, self.bias, stride, pad, in_act, out_ep, self.weight.grad, self.bias.grad,
                               ep_bwd)
        return ops.conv2d(x, self.weight, self.bias, stride, pad, in_act, self.weight.grad, self.bias.grad)


class _HiddenHeads:
    """the MNIST / SVHN towers name their heads hidden_mu / hidden_logvar (models/encoders.py:246-247,455-456)"""

    def _heads(self):
        return self.hidden_mu, self.hidden_logvar


class Enc_MNIST(_HiddenHeads, VaeEncoder):
    """models/encoders.py:226-265: 784 -> 400 -> 400 (ReLU) MLP, heads.  state_dict keys enc.{0,1}.0.*, hidden_*."""

    def __init__(self, latent_dim, data_dim, latent_private, enc_mu_logvar):
        super().__init__(latent_dim, data_dim, latent_private, enc_mu_logvar, net_type=NetworkTypes.FNN)
        self.net_type = "CNN"
        self.hidden_dim = 400
        self.enc = nn.ModuleList([nn.ModuleList([HipLinear(784, self.hidden_dim)]),
                                  nn.ModuleList([HipLinear(self.hidden_dim, self.hidden_dim, H.ACT_RELU)])])
        self.hidden_mu = HipLinear(self.hidden_dim, self.out_dim)
        self.hidden_logvar = HipLinear(self.hidden_dim, self.out_dim)

    def forward(self, x):
        x = x["data"] if isinstance(x, dict) else x
        h = x.reshape(x.shape[0], -1).float()
        h = self.enc[1][0](self.enc[0][0](h))            # the ReLUs are applied by the consumers
        return self.process_output(h, H.ACT_RELU)


class Enc_SVHN(_HiddenHeads, VaeEncoder):
    """models/encoders.py:434-478: Conv2d k4 3->32->64->64 (s2 p1), 64->128 (s2 p0), ReLU, heads on the 128 features"""

    def __init__(self, latent_dim, data_dim, latent_private, enc_mu_logvar):
        super().__init__(latent_dim, data_dim, latent_private, enc_mu_logvar, net_type=NetworkTypes.CNN)
        self.net_type = "CNN"
        self.conv1 = HipConv(3, 32, 4, 2, 1)
        self.conv2 = HipConv(32, 64, 4, 2, 1, H.ACT_RELU)
        self.conv3 = HipConv(64, 64, 4, 2, 1, H.ACT_RELU)
        self.conv4 = HipConv(64, 128, 4, 2, 0, H.ACT_RELU)
        self.hidden_mu = HipLinear(128, self.out_dim)
        self.hidden_logvar = HipLinear(128, self.out_dim)

    def forward(self, x):
        x = x["data"] if isinstance(x, dict) else x
        h = self.conv4(self.conv3(self.conv2(self.conv1(x.float()))))
        return self.process_output(h.reshape(h.shape[0], -1), H.ACT_RELU)


class Enc_CNN(VaeEncoder):
    """models/encoders.py:86-127: `encoder: CNN` = ResNet-50 -> SiLU -> heads on the 1000 logits.  The tower itself is
    models/resnet.py (NHWC GEMM formulation on the MFMA kernels); ImageNet weights come from a user file
    (MMVAE_RESNET50_WEIGHTS), else torchvision's random initialisation -- the reference downloads them."""

    def __init__(self, latent_dim, data_dim, latent_private, enc_mu_logvar):
        from .resnet import ResNet50, maybe_load_pretrained
        data_dim = (3, 64, 64)
        super().__init__(latent_dim, data_dim, latent_private, enc_mu_logvar, net_type=NetworkTypes.CNN)
        self.hidden_dim = 1000
        self.reshape = (32, 4, 4)
        self.resnet = ResNet50()
        self.pretrained = maybe_load_pretrained(self.resnet)
        self.init_final_layers(self.hidden_dim)

    def forward(self, x):
        if isinstance(x, dict):
            x = x["data"]
        return self.process_output(self.resnet(x), H.ACT_SILU)


class _GruParams(nn.Module):
    """the parameters of torch.nn.GRU(H, H, 1, bidirectional=True) under torch's names and in torch's order, initialised
    as nn.GRU.reset_parameters does (U(-1/sqrt(H), 1/sqrt(H))); gate order r, z, n in the stacked (3H, .) tensors"""

    def __init__(self, hidden):
        super().__init__()
        k = 1.0 / (hidden ** 0.5)
        for sfx in ("", "_reverse"):
            for name, shape in (("weight_ih_l0", (3 * hidden, hidden)), ("weight_hh_l0", (3 * hidden, hidden)),
                                ("bias_ih_l0", (3 * hidden,)), ("bias_hh_l0", (3 * hidden,))):
                self.register_parameter(name + sfx, nn.Parameter(torch.empty(shape).uniform_(-k, k)))


class Enc_TxtRNN(VaeEncoder):
    """models/encoders.py:840-869: Embedding(njoints*nfeats, 512) -> bidirectional one-layer GRU(512) -> `output[-1]` ->
    sum of the two directions -> Linear(512, 2 D') -> chunk -> softmax + eta.

    A DEFINED path, PARITY UNPINNED against the reference: its own forward crashes on the batch format the data module
    produces (the (B,T,V) one-hot cast to long and embedded gives nn.GRU a 5-D tensor) and there is no
    Dec_TxtRNN (pair it with `decoder: TxtTransformer`).  What its lines spell for a batch of token ids is restated in
    oracle/mmvae_oracle.py: enc_txt_rnn (checked against torch.nn.GRU itself, tests/test_oracle_txtrnn.py): ids = argmax
    of the one-hot rows (a padding row is token 0 -- the reference never reads `masks`), sequence-first layout,
    `output[-1]` = the forward direction's state after all T steps + the reverse direction's state after its FIRST step.
    Parameters carry torch's names (embed.weight, gru.weight_ih_l0[_reverse], ..., o2p.*) so a reference state_dict
    loads.  Kernels: csrc/gru.hip."""

    def __init__(self, latent_dim, data_dim, latent_private, enc_mu_logvar, hidden_size=512, n_layers=1, bidirectional=True):
        super().__init__(latent_dim, data_dim, latent_private, enc_mu_logvar, net_type=NetworkTypes.TXTTRANSFORMER)
        assert n_layers == 1 and bidirectional, "Enc_TxtRNN: the reference's configuration (one layer, bidirectional)"
        self.input_size = data_dim[-1] * data_dim[-2]
        self.hidden_size, self.n_layers, self.bidirectional = hidden_size, n_layers, bidirectional
        self.embed = nn.Embedding(self.input_size, hidden_size)
        self.gru = _GruParams(hidden_size)
        self.o2p = HipLinear(hidden_size, self.out_dim * 2)

    def forward(self, batch):
        x = batch["data"] if isinstance(batch, dict) else batch
        g = self.gru
        ids, oh = ops.gru_token_ids(x)                                      # (T,B) int32, (T*B, V)
        T, B = ids.shape
        E = self.embed.weight
        # the embedding folded into the input projections: pt (3H, V) = W_ih E^T (gx_t = pt[:, id_t] + b_ih)
        pt = ops.linear(g.weight_ih_l0, E, None, H.ACT_NONE, E.grad, None)
        pt_r = ops.linear(g.weight_ih_l0_reverse, E, None, H.ACT_NONE, E.grad, None)
        h_fwd = ops.gru_forward(pt, g.bias_ih_l0, g.weight_hh_l0, g.bias_hh_l0, ids, oh, g.weight_hh_l0.grad,
                                g.bias_hh_l0.grad)
        out = ops.gru_cell0(pt_r, g.bias_ih_l0_reverse, g.bias_hh_l0_reverse, ids[T - 1], oh[(T - 1) * B:], h_fwd)
        return self.finish_heads(self.o2p(out))                             # [mu | logvar head] = o2p's two halves


FUSED_TXT_LAYERS = True      # module switches (tests compare both forms)
FUSED_HEADS = True     # text encoder: heads inside the last layer's launch
FUSED_FFN = True         # d_model-32 layers outside the fused-layer shapes


class HipTransformerEncoderLayer(nn.Module):
    """torch.nn.TransformerEncoderLayer (post-norm, gelu) parameter layout on the HIP kernels."""

    def __init__(self, d, nhead, ff):
        super().__init__()
        self.self_attn = HipSelfAttention(d, nhead)
        self.linear1 = HipLinear(d, ff)
        self.linear2 = HipLinear(ff, d, H.ACT_GELU)
        self.norm1 = HipLayerNorm(d)
        self.norm2 = HipLayerNorm(d)

    def fused_params(self):
        a = self.self_attn
        return {"in_w": a.in_proj_weight, "in_b": a.in_proj_bias, "out_w": a.out_proj.weight, "out_b": a.out_proj.bias,
                "l1_w": self.linear1.weight, "l1_b": self.linear1.bias, "l2_w": self.linear2.weight,
                "l2_b": self.linear2.bias, "n1_g": self.norm1.weight, "n1_b": self.norm1.bias,
                "n2_g": self.norm2.weight, "n2_b": self.norm2.bias}

    def forward(self, x, mask_u8, ds=None, time_mean=False, heads=None):
        """`ds`: dict of DropSpec for train mode (attn, drop1, ffn, drop2) or None.  `time_mean`: return the mean of
        the output over the frames, (N, d) -- the pooling that follows the LAST encoder layer, folded into the fused
        layer's launch when that path is taken.  `heads` (VaeComponent.packed_heads()): additionally apply the packed
        posterior heads to the pooled feature, returning (N, 2D')."""
        L, _, d = x.shape
        ff, nh = self.linear1.out_features, self.self_attn.nhead
        if FUSED_TXT_LAYERS and ops.txt_layer_supported(L, d, ff, nh, False):
            # one workgroup per sequence runs the whole layer (csrc/txtlayer.hip); same arithmetic and dropout masks
            p = self.fused_params()
            return ops.txt_layer(x, None, mask_u8, ops.TxtLayerMeta(d, ff, nh, False, ds, time_mean, heads), p,
                                 {k: v.grad for k, v in p.items()})
        if heads is not None:
            z = ops.mean_over_time(self.forward(x, mask_u8, ds))
            return ops.linear(z, heads[0], heads[1], H.ACT_NONE, heads[2], heads[3])
        if time_mean:
            return ops.mean_over_time(self.forward(x, mask_u8, ds))
        fused_ffn = FUSED_FFN and x.is_cuda and ops.ffn32_supported(d, ff)
        # the gradients of the two residual connections join their sub-layers' first backward kernels (ops.ResidualGrad)
        s1 = ops.residual_sink(x)
        if ds is None:
            x = self.norm1(self.self_attn(x, mask_u8, res_sink=s1, lazy_out=self.self_attn.lazy_out(x)), x, res_sink=s1)
            if fused_ffn:
                s2 = ops.residual_sink(x)
                return self.norm2(self._ffn(x, None, s2, (x, self.norm2.weight, self.norm2.bias, None)), x, res_sink=s2)
            s2 = ops.residual_sink(x)
            return self.norm2(self.linear2(self.linear1(x, res_sink=s2)), x, res_sink=s2)
        x = self.norm1(self.self_attn(x, mask_u8, ds["attn"], res_sink=s1, lazy_out=self.self_attn.lazy_out(x)), x, ds["drop1"],
                       res_sink=s1)
        if fused_ffn:      # the (L*N, ff) hidden activation never leaves the registers (csrc/ffn.hip)
            s2 = ops.residual_sink(x)
            return self.norm2(self._ffn(x, ds["ffn"], s2, (x, self.norm2.weight, self.norm2.bias, ds["drop2"])), x, ds["drop2"],
                              res_sink=s2)
        s2 = ops.residual_sink(x)
        h = ops.dropout_act(self.linear1(x, res_sink=s2), H.ACT_GELU, ds["ffn"])           # dropout(gelu(.)) materialised
        return self.norm2(self.linear2(h, in_act=H.ACT_NONE), x, ds["drop2"], res_sink=s2)

    def _ffn(self, x, drop, res_sink=None, ln=None):
        return ops.ffn32(x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, drop,
                         stack_ffn_image(self), res_sink, ln)


def stack_ffn_image(layer):
    """The split-bf16 weight image of this layer's feed-forward block (ops.ffn32's `wsplit`): the FIRST layer of a stack
    prepares the images of all its layers in one launch (ops.ffn32_prep_many), the others pick theirs up -- every layer
    splitting its own weights is one 5 us launch per layer on the tower's chain.  None: ops.Ffn32 makes its own."""
    stack, idx = getattr(layer, "_ffn_stack", (None, 0))
    if stack is None or not ops.FFN32_SPLIT_BF16 or len(stack) > 16:
        return None
    if idx == 0:
        imgs = ops.ffn32_prep_many([(l.linear1.weight, l.linear2.weight) for l in stack])
        # tagged with the weight generation (every optimiser launch of this package bumps ops.WEIGHT_GEN) and the index of
        # the layer that may take the next image: a layer called on its own, after a forward that stopped early, or after
        # an optimiser step gets None and splits its own weights (the list used to be trusted blindly)
        stack[0]._ffn_images = {"imgs": imgs, "gen": ops.WEIGHT_GEN[0], "next": 0} if imgs is not None else None
    tag = getattr(stack[0], "_ffn_images", None)
    if tag is None or tag["gen"] != ops.WEIGHT_GEN[0] or tag["next"] != idx:
        stack[0]._ffn_images = None
        return None
    tag["next"] = idx + 1
    img = tag["imgs"][idx]
    if idx == len(stack) - 1:
        stack[0]._ffn_images = None
    return img


def link_ffn_stack(layers):
    for i, l in enumerate(layers):
        object.__setattr__(l, "_ffn_stack", (list(layers), i))      # (not a submodule registration)


class HipTransformerStack(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        link_ffn_stack(self.layers)


class Enc_TxtTransformer(VaeEncoder):
    """models/encoders.py:790-837 incl. the PositionalEncoding quirk."""

    def __init__(self, latent_dim, data_dim, latent_private, enc_mu_logvar, ff_size=128, num_layers=1, num_heads=2,
                 dropout=0.1, activation="gelu"):
        super().__init__(latent_dim, data_dim, latent_private, enc_mu_logvar, net_type=NetworkTypes.TXTTRANSFORMER)
        assert activation == "gelu" and num_layers >= 1
        self.net_type = "TxtTransformer"
        self.njoints, self.nfeats = data_dim[-1], data_dim[-2]
        self.ff_size, self.num_layers, self.num_heads, self.dropout = ff_size, num_layers, num_heads, dropout
        self.hidden_dim = self.out_dim
        self.input_feats = self.njoints * self.nfeats
        self.embedding_size = 2
        self.embedding = nn.Embedding(self.input_feats, self.embedding_size)
        self.sequence_pos_encoder = PositionalEncoding(self.embedding_size, self.dropout)
        d = self.input_feats * self.embedding_size
        self.seqTransEncoder = HipTransformerStack([HipTransformerEncoderLayer(d, num_heads, ff_size)
                                                    for _ in range(num_layers)])
        self.mu_layer = ModuleWrap(HipLinear(d, self.out_dim))
        self.logvar_layer = ModuleWrap(HipLinear(d, self.out_dim))
        self.drop_state = DropoutState()

    takes_repeat = True      # forward accepts batch["repeat"] (POE.objective)

    @staticmethod
    def repeat_ok(batch):
        """repeated passes can share one call only in the batch-indexed positional branch (B != T, B != 1: the other branch
        relabels memory and mixes the samples of a pass)"""
        bs, nframes = batch["data"].shape[0], batch["data"].shape[1]
        return bs != nframes and bs != 1

    def forward(self, batch):
        x, mask = batch["data"], batch["masks"]
        bs, nframes, _ = x.shape
        if bs > self.sequence_pos_encoder.pe.shape[0]:
            raise RuntimeError("batch larger than the positional table (reference: pe[:B], nn_modules.py:419)")
        if mask is None:
            mask = torch.ones(bs, nframes, dtype=torch.bool, device=x.device)
        mode = 1 if (bs == nframes or bs == 1) else 0
        # `repeat` = R (POE.objective): R passes over this batch as ONE call of R * bs sequences, row k * bs + b -- every
        # sample keeps the positional term of its position b in the batch (ops.embed_pe), every pass its own dropout masks
        rep = int(batch.get("repeat", 1)) if mode == 0 else 1
        if rep > 1:
            mask = mask.repeat(rep, 1)
        mask_u8 = ops.as_u8(mask)            # validity bytes, read in place by the attention kernel
        w = self.embedding.weight
        p = self.dropout
        if self.training and p > 0:       # nn.Dropout sites of the reference: PE + 4 per layer
            slot, call = self.drop_state.begin()
            sp = lambda site, name: self.drop_state.spec(slot, call, site, p, name)
            d_pe = sp(0, "pe")
            ds = [{"attn": sp(1 + 4 * i, f"l{i}.attn"), "drop1": sp(2 + 4 * i, f"l{i}.drop1"),
                   "ffn": sp(3 + 4 * i, f"l{i}.ffn"), "drop2": sp(4 + 4 * i, f"l{i}.drop2")}
                  for i in range(len(self.seqTransEncoder.layers))]
        else:
            d_pe, ds = None, [None] * len(self.seqTransEncoder.layers)
        h = ops.embed_pe(x, w, self.sequence_pos_encoder.pe.view(-1, 2), mode, w.grad, d_pe, rep)   # (T, R * B, 2V)
        last = len(self.seqTransEncoder.layers) - 1
        heads = self.packed_heads() if (FUSED_HEADS and torch.is_grad_enabled()) else None
        for i, (layer, d) in enumerate(zip(self.seqTransEncoder.layers, ds)):
            # z = h.mean(0) -- and the posterior heads when they are one packed layer -- folded into the last layer
            h = layer(h, mask_u8, d, time_mean=(i == last), heads=heads if i == last else None)
        return self.finish_heads(h) if heads is not None else self.process_output(h)


class Enc_Transformer(VaeEncoder):
    """models/encoders.py:656-729 (ACTOR-style encoder for action sequences (B, T, joints, feats)):
    Linear(joints*feats -> D') + time positional encoding + dropout, `num_layers` post-norm encoder layers
    (d = D', ff 1024, 2 heads, gelu), mean over time, heads Linear(D' -> D')."""

    def __init__(self, latent_dim, data_dim, latent_private, enc_mu_logvar, ff_size=1024, num_layers=8, num_heads=2,
                 dropout=0.1, activation="gelu"):
        super().__init__(latent_dim, data_dim, latent_private, enc_mu_logvar, net_type=NetworkTypes.TRANSFORMER)
        assert activation == "gelu" and num_layers >= 1
        self.net_type = "Transformer"
        self.njoints, self.nfeats = data_dim[1], (data_dim[2] if len(data_dim) > 2 else 1)
        self.ff_size, self.num_layers, self.num_heads, self.dropout = ff_size, num_layers, num_heads, dropout
        self.activation = activation
        self.input_feats = self.njoints * self.nfeats
        d = self.out_dim
        self.mu_layer = ModuleWrap(HipLinear(d, d))
        self.logvar_layer = ModuleWrap(HipLinear(d, d))
        self.skel_Embedding = ModuleWrap(HipLinear(self.input_feats, d))
        self.sequence_pos_encoder = PositionalEncoding(d, self.dropout)
        self.seqTransEncoder = HipTransformerStack([HipTransformerEncoderLayer(d, num_heads, ff_size)
                                                    for _ in range(num_layers)])
        self.drop_state = DropoutState()

    def forward(self, batch):
        x, mask = batch["data"], batch["masks"]
        if x.dim() == 3:
            x = x.unsqueeze(-1)
        bs, nframes = x.shape[0], x.shape[1]
        if mask is None:
            mask = torch.ones(bs, nframes, dtype=torch.bool, device=x.device)
        mask_u8 = ops.as_u8(mask)
        d = self.out_dim
        x = x.permute(1, 0, 2, 3).reshape(nframes, bs, self.input_feats).float().contiguous()
        nl = len(self.seqTransEncoder.layers)
        if self.training and self.dropout > 0:    # nn.Dropout sites: PE + 4 per layer
            slot, call = self.drop_state.begin()
            sp = lambda site, name: self.drop_state.spec(slot, call, site, self.dropout, name)
            d_pe = sp(0, "pe")
            ds = [{"attn": sp(1 + 4 * i, f"l{i}.attn"), "drop1": sp(2 + 4 * i, f"l{i}.drop1"),
                   "ffn": sp(3 + 4 * i, f"l{i}.ffn"), "drop2": sp(4 + 4 * i, f"l{i}.drop2")} for i in range(nl)]
        else:
            d_pe, ds = None, [None] * nl
        h = self.skel_Embedding(x)
        pe = self.sequence_pos_encoder.pe[:nframes].reshape(nframes, d)
        h = ops.add_pe_dropout(h, pe, nframes, bs, d, d_pe)
        for i, (layer, dd) in enumerate(zip(self.seqTransEncoder.layers, ds)):
            h = layer(h, mask_u8, dd, time_mean=(i == nl - 1))
        return self.process_output(h)
