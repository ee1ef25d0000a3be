"""Decoder towers behind the reference's plugin names (`Dec_<Name>(latent_dim, data_dim, latent_private)`;
reference: models/decoders.py).  Same signatures / attributes / state_dict keys (including the `.module.`
segment of the reference's per-layer nn.DataParallel wrappers); arithmetic on the gfx950 kernels."""
import numpy as np
import torch
import torch.nn as nn

from .. import hipops as H
from .. import ops
from . import encoders
from .NetworkTypes import NetworkRoles, NetworkTypes
from .encoders import VaeComponent
from .nn_modules import DropoutState, HipLayerNorm, HipLinear, HipSelfAttention, ModuleWrap, PositionalEncoding


class VaeDecoder(VaeComponent):
    def __init__(self, latent_dim, data_dim, latent_private, net_type: NetworkTypes):
        # the reference passes net_type in the enc_mu_logvar slot (decoders.py:32); no numeric effect
        super().__init__(latent_dim, data_dim, latent_private, net_type, NetworkRoles.DECODER)


class HipConvT2d(nn.Module):
    """nn.ConvTranspose2d(k=4, s=2, p=1) parameters (torch default init) on the MFMA kernels."""

    def __init__(self, cin, cout, in_act=H.ACT_NONE, out_ep=H.EP_NONE):
        super().__init__()
        ref = nn.ConvTranspose2d(cin, cout, 4, stride=2, padding=1)
        self.weight, self.bias = ref.weight, ref.bias
        self.in_act, self.out_ep = in_act, out_ep

    def flat_groups(self):
        return [[self.weight, self.bias]]

    def forward(self, x, ep_bwd=True):
        return ops.convT2d_k4s2(x, self.weight, self.bias, self.in_act, self.out_ep, self.weight.grad, self.bias.grad,
                                ep_bwd)


class Dec_CNN(VaeDecoder):
    """models/decoders.py:35-98: 3x [Linear + ReLU] D'->512->512->512, view (K*B,32,4,4), 3x [ConvT + ReLU],
    ConvT 32->3, sigmoid, clamp[1e-6, 1-1e-6]; output labelled (..., 64,64,3) by view (no permute).
    Every ReLU is applied by the consuming layer; sigmoid+clamp is the last kernel's epilogue."""

    def __init__(self, latent_dim, data_dim, latent_private):
        super().__init__(latent_dim, data_dim, latent_private, net_type=NetworkTypes.CNN)
        hid, hidden_dim = 32, 512
        self.reshape = (hid, 4, 4)
        self.n_chan = 3
        self.lin1 = ModuleWrap(HipLinear(self.out_dim, hidden_dim, H.ACT_NONE))
        self.lin2 = ModuleWrap(HipLinear(hidden_dim, hidden_dim, H.ACT_RELU))
        self.lin3 = ModuleWrap(HipLinear(hidden_dim, int(np.prod(self.reshape)), H.ACT_RELU))
        self.convT_64 = ModuleWrap(HipConvT2d(hid, hid, H.ACT_RELU))
        self.convT1 = ModuleWrap(HipConvT2d(hid, hid, H.ACT_RELU))
        self.convT2 = ModuleWrap(HipConvT2d(hid, hid, H.ACT_RELU))
        self.convT3 = ModuleWrap(HipConvT2d(hid, self.n_chan, H.ACT_RELU, H.EP_SIGMOID_CLAMP))
        self.register_buffer("_scale", torch.tensor(0.75), persistent=False)   # the decoders' fixed likelihood scale

    def forward(self, z):
        z_in = z
        z = z["latents"]
        if z.dim() == 2:
            z = z.unsqueeze(0)
        K, bs = z.shape[0], z.shape[1]
        u = self.lin3(self.lin2(self.lin1(z.reshape(K * bs, -1))))
        u = u.view(bs * K, *self.reshape)
        # (K*B,3,64,64) clamped sigmoid.  `raw` holds the same values, but its gradient is the LOGITS' gradient; `d` goes
        # through SigmoidClampOut.  A bce loss takes the fused closed-form path via `out._bce_src` (objectives.py).
        u3 = self.convT2(self.convT1(self.convT_64(u)))
        tgt = z_in.get("bce_target") if isinstance(z_in, dict) else None
        last = self.convT3.module
        if tgt is not None and ops.convT3_bce_supported(u3, last.weight, tgt):
            # training objective with a bce likelihood (the mixer hands the target over and has announced the term's ELBO
            # weight, ops.ConstSeed): last layer + sigmoid + clamp + reconstruction row sums in ONE launch, x_hat is never
            # stored.  The "output" is the (K*B,) row sums, marked for recon_rowsum (objectives.py)
            rows = ops.convT3_bce(u3, last.weight, last.bias, last.in_act, last.weight.grad, last.bias.grad, tgt)
            rows._bce_rows = True
            return rows, self._scale
        raw = last(u3, ep_bwd=False)
        d = ops.sigmoid_clamp_out(raw)
        d = d.view(*z.size()[:-1], *self.data_dim)                          # decoders.py:96 (view, no permute)
        out = d.squeeze().reshape(-1, *self.data_dim)
        out._bce_src = raw
        return out, self._scale


_ONES = {}


def _ones_mask(bs, T, device):
    """the `masks=None` decoder mask (every step valid), constant per shape: built once (read-only for every consumer)"""
    key = (bs, T, str(device))
    m = _ONES.get(key)
    if m is None:
        if len(_ONES) >= 16:
            _ONES.clear()
        m = _ONES[key] = torch.ones(bs, T, dtype=torch.bool, device=device)
    return m


class HipTransformerDecoderLayer(nn.Module):
    """torch.nn.TransformerDecoderLayer (post-norm, gelu) parameter layout.  Memory length 1 (K = 1 latent
    sample, decoders.py:718): the cross-attention softmax is identically 1."""

    def __init__(self, d, nhead, ff):
        super().__init__()
        self.self_attn = HipSelfAttention(d, nhead)
        self.multihead_attn = HipSelfAttention(d, nhead)
        self.linear1 = HipLinear(d, ff)
        self.linear2 = HipLinear(ff, d, H.ACT_GELU)
        self.norm1 = HipLayerNorm(d)
        self.norm2 = HipLayerNorm(d)
        self.norm3 = HipLayerNorm(d)

    def fused_params(self):
        a, c = self.self_attn, self.multihead_attn
        return {"in_w": a.in_proj_weight, "in_b": a.in_proj_bias, "out_w": a.out_proj.weight, "out_b": a.out_proj.bias,
                "l1_w": self.linear1.weight, "l1_b": self.linear1.bias, "l2_w": self.linear2.weight,
                "l2_b": self.linear2.bias, "n1_g": self.norm1.weight, "n1_b": self.norm1.bias,
                "n2_g": self.norm2.weight, "n2_b": self.norm2.bias, "n3_g": self.norm3.weight, "n3_b": self.norm3.bias,
                "x_in_w": c.in_proj_weight, "x_in_b": c.in_proj_bias, "x_out_w": c.out_proj.weight,
                "x_out_b": c.out_proj.bias}

    def forward(self, x, mem, mask_u8, ds=None):
        L, _, d = x.shape
        ff, nh = self.linear1.out_features, self.self_attn.nhead
        if encoders.FUSED_TXT_LAYERS and ops.txt_layer_supported(L, d, ff, nh, True):
            p = self.fused_params()
            return ops.txt_layer(x, mem, mask_u8, ops.TxtLayerMeta(d, ff, nh, True, ds), p,
                                 {k: v.grad for k, v in p.items()})
        fused_ffn = encoders.FUSED_FFN and x.is_cuda and ops.ffn32_supported(d, ff)
        ffn = lambda t, drop, sink, ldrop=None: ops.ffn32(t, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                                          self.linear2.bias, drop, encoders.stack_ffn_image(self), sink,
                                                          (t, self.norm3.weight, self.norm3.bias, ldrop))
        # residual gradients of the self-attention and feed-forward blocks join those blocks' first backward kernels
        s1 = ops.residual_sink(x)
        if ds is None:
            x = self.norm1(self.self_attn(x, mask_u8, res_sink=s1, lazy_out=self.self_attn.lazy_out(x)), x, res_sink=s1)
            x = self.norm2(x, self.multihead_attn.value_path(mem))     # (N,d) residual broadcast over time
            if fused_ffn:
                s3 = ops.residual_sink(x)
                return self.norm3(ffn(x, None, s3), x, res_sink=s3)
            s3 = ops.residual_sink(x)      # (the launch-per-op form: the residual's gradient joins linear1's data gradient)
            return self.norm3(self.linear2(self.linear1(x, res_sink=s3)), x, res_sink=s3)
        L = x.shape[0]
        x = self.norm1(self.self_attn(x, mask_u8, ds["attn"], res_sink=s1, lazy_out=self.self_attn.lazy_out(x)), x, ds["drop1"],
                       res_sink=s1)
        ca = self.multihead_attn.value_path(mem, L, ds["xattn"],          # (L,N,d): weight dropout varies with l
                                            lazy_out=self.multihead_attn.lazy_out(x))
        x = self.norm2(ca, x, ds["drop2"])
        if fused_ffn:
            s3 = ops.residual_sink(x)
            return self.norm3(ffn(x, ds["ffn"], s3, ds["drop3"]), x, ds["drop3"], res_sink=s3)
        s3 = ops.residual_sink(x)
        h = ops.dropout_act(self.linear1(x, res_sink=s3), H.ACT_GELU, ds["ffn"])
        return self.norm3(self.linear2(h, in_act=H.ACT_NONE), x, ds["drop3"], res_sink=s3)


class HipTransformerDecoderStack(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        encoders.link_ffn_stack(self.layers)


class Dec_TxtTransformer(VaeDecoder):
    """models/decoders.py:668-723.

    K = 1 latent sample: the reference's behaviour (memory of length 1).  K > 1: the reference attends over the K
    samples as a length-K memory and returns ONE (B,T,V) output that its own loss code then cannot reshape
    (objectives.py:120) -- no objective runs through it.  DEFINED EXTENSION here (parity unpinned; restated
    in oracle/mmvae_oracle.py: dec_txt_transformer keep_k): every sample is decoded on its own, (K,B) flattened into
    the batch axis exactly as Dec_CNN does (decoders.py:73-76): output (K*B, T, V), row k*B + b."""

    takes_keep_steps = True      # forward() honours batch["keep_steps"] (POE.objective)

    def __init__(self, latent_dim, data_dim, latent_private, ff_size=128, num_layers=1, num_heads=2, dropout=0.1,
                 activation="gelu"):
        super().__init__(latent_dim, data_dim, latent_private, net_type=NetworkTypes.TXTTRANSFORMER)
        assert activation == "gelu"
        self.net_type = "Transformer"
        self.njoints = data_dim[1]
        self.nfeats = data_dim[2] if len(data_dim) > 2 else 1
        self.data_dim = data_dim
        self.latent_dim = latent_dim
        self.ff_size, self.num_layers, self.num_heads, self.dropout = ff_size, num_layers, num_heads, dropout
        self.input_feats = self.njoints * self.nfeats
        self.seqTransDecoder = HipTransformerDecoderStack(
            [HipTransformerDecoderLayer(self.out_dim, num_heads, ff_size) for _ in range(num_layers)])
        self.finallayer = ModuleWrap(HipLinear(self.out_dim, self.input_feats))
        self.sequence_pos_encoder = ModuleWrap(PositionalEncoding(self.out_dim, self.dropout))
        self._tq_cache = {}
        self.register_buffer("_scale", torch.tensor(0.75), persistent=False)
        self.drop_state = DropoutState()

    def _timequeries(self, T, bs, D, device):
        """PositionalEncoding(zeros(T,bs,D)) = pe[:T] broadcast over the batch (decoders.py:716-717); constant for
        a given (T, bs), so it is materialised once."""
        key = (T, bs, D, str(device))
        tq = self._tq_cache.get(key)
        if tq is None:
            pe = self.sequence_pos_encoder.module.pe[:T].to(device)          # (T,1,D)
            tq = pe.expand(T, bs, D).contiguous()
            if len(self._tq_cache) >= 8:      # (a PoE step decodes at two (T, bs) shapes: keep both, bound the rest)
                self._tq_cache.clear()
            self._tq_cache[key] = tq
        return tq

    def forward(self, batch):
        z = batch["latents"]
        z = z.unsqueeze(0) if z.dim() == 2 else z
        mask = batch["masks"]
        K, bs, D = z.shape
        if mask is None:
            mask = _ones_mask(bs, self.data_dim[0], z.device)
        mask = mask.to(z.device)
        if K != 1:          # K-preserving extension (class docstring): K * B independent sequences
            mask = mask.repeat(K, 1)
            bs = K * bs
        T = mask.shape[1]
        mask_u8 = ops.as_u8(mask)
        x = self._timequeries(T, bs, D, z.device)
        mem = z.reshape(bs, D)                     # a view (z[0] would cost a select-backward fill + copy)
        p = self.dropout
        nl = len(self.seqTransDecoder.layers)
        if self.training and p > 0:       # nn.Dropout sites of the reference: PE + 6 per layer
            slot, call = batch.get("drop_begun") or self.drop_state.begin()      # (begun by the caller: POE._decoder_lanes)
            sp = lambda site, name: self.drop_state.spec(slot, call, site, p, name)
            x = ops.dropout_act(x, H.ACT_NONE, sp(0, "pe"))
            ds = [{"attn": sp(1 + 6 * i, f"l{i}.attn"), "drop1": sp(2 + 6 * i, f"l{i}.drop1"),
                   "xattn": sp(3 + 6 * i, f"l{i}.xattn"), "drop2": sp(4 + 6 * i, f"l{i}.drop2"),
                   "ffn": sp(5 + 6 * i, f"l{i}.ffn"), "drop3": sp(6 + 6 * i, f"l{i}.drop3")} for i in range(nl)]
        else:
            ds = [None] * nl
        for layer, d in zip(self.seqTransDecoder.layers, ds):
            x = layer(x, mem, mask_u8, d)
        out = self.finallayer(x)                                              # (T, bs, V)
        # (bs, T, V), zero at padding; "keep_steps": the caller compares the first steps only (a PoE subset without this
        # modality decodes at full length against a target that carries a shorter mask: objectives.py:30-52)
        keep = batch.get("keep_steps")
        out = ops.permute_mask(out, mask_u8, keep if (keep is not None and keep < T) else None)
        return out, self._scale


class Dec_Transformer(VaeDecoder):
    """models/decoders.py:541-616: time queries PE(zeros) + dropout, `num_layers` post-norm decoder layers over the
    length-1 memory z (d = D', ff 1024, 2 heads), Linear(D' -> joints*feats), padded steps zeroed, (B, T, joints, feats)."""

    def __init__(self, latent_dim, data_dim, latent_private, ff_size=1024, num_layers=4, num_heads=2, dropout=0.1,
                 activation="gelu"):
        super().__init__(latent_dim, data_dim, latent_private, net_type=NetworkTypes.TRANSFORMER)
        assert activation == "gelu"
        self.net_type = "Transformer"
        self.njoints = data_dim[1]
        self.nfeats = data_dim[2] if len(data_dim) > 2 else 1
        self.data_dim = data_dim
        self.latent_dim = latent_dim
        self.ff_size, self.num_layers, self.num_heads, self.dropout = ff_size, num_layers, num_heads, dropout
        self.activation = activation
        self.input_feats = self.njoints * self.nfeats
        self.sequence_pos_encoder = ModuleWrap(PositionalEncoding(self.out_dim, self.dropout))
        self.seqTransDecoder = HipTransformerDecoderStack(
            [HipTransformerDecoderLayer(self.out_dim, num_heads, ff_size) for _ in range(num_layers)])
        self.finallayer = ModuleWrap(HipLinear(self.out_dim, self.input_feats))
        self.register_buffer("_scale", torch.tensor(0.75), persistent=False)
        self.drop_state = DropoutState()

    def forward(self, batch):
        z, mask = batch["latents"], batch["masks"]
        D = self.out_dim
        z = z.reshape(-1, D)                       # memory of length 1: (K*B, D')
        bs = z.shape[0]
        if mask is not None:
            if bs > mask.shape[0]:
                mask = mask.repeat(int(bs / mask.shape[0]), 1)
            mask = mask.to(z.device)
        else:
            mask = _ones_mask(bs, self.data_dim[0], z.device)
        T = mask.shape[1]
        mask_u8 = ops.as_u8(mask)
        nl = len(self.seqTransDecoder.layers)
        if self.training and self.dropout > 0:    # nn.Dropout sites: PE + 6 per layer
            slot, call = batch.get("drop_begun") or self.drop_state.begin()      # (begun by the caller: POE._decoder_lanes)
            sp = lambda site, name: self.drop_state.spec(slot, call, site, self.dropout, name)
            d_pe = sp(0, "pe")
            ds = [{"attn": sp(1 + 6 * i, f"l{i}.attn"), "drop1": sp(2 + 6 * i, f"l{i}.drop1"),
                   "xattn": sp(3 + 6 * i, f"l{i}.xattn"), "drop2": sp(4 + 6 * i, f"l{i}.drop2"),
                   "ffn": sp(5 + 6 * i, f"l{i}.ffn"), "drop3": sp(6 + 6 * i, f"l{i}.drop3")} for i in range(nl)]
        else:
            d_pe, ds = None, [None] * nl
        pe = self.sequence_pos_encoder.module.pe[:T].reshape(T, D)
        x = ops.add_pe_dropout(None, pe, T, bs, D, d_pe)
        for layer, d in zip(self.seqTransDecoder.layers, ds):
            x = layer(x, z, mask_u8, d)
        out = self.finallayer(x)                                              # (T, bs, joints*feats)
        out = ops.permute_mask(out, mask_u8)                                  # (bs, T, .), padded steps zero
        return out.view(bs, T, self.njoints, self.nfeats), self._scale


class Dec_MNIST(VaeDecoder):
    """models/decoders.py:230-270: D' -> 400 -> 400 (ReLU) -> 784, sigmoid, reshaped to data_dim and permuted to NCHW"""

    def __init__(self, latent_dim, data_dim, latent_private):
        super().__init__(latent_dim, data_dim, latent_private, net_type=NetworkTypes.FNN)
        self.data_dim = data_dim
        self.net_type = "CNN"
        self.hidden_dim = 400
        self.dec = nn.ModuleList([nn.ModuleList([HipLinear(self.out_dim, self.hidden_dim)]),
                                  nn.ModuleList([HipLinear(self.hidden_dim, self.hidden_dim, H.ACT_RELU)])])
        self.fc3 = HipLinear(self.hidden_dim, 784, H.ACT_RELU)
        self.register_buffer("_scale", torch.tensor(0.75), persistent=False)

    def forward(self, z):
        z = z["latents"]
        lead = z.shape[:-1]
        # sigmoid in the last GEMM's epilogue; `raw` holds the same values but takes the LOGITS' gradient (a lprob loss
        # feeds it directly through `_lprob_src`, anything else goes through SigmoidOut)
        raw = self.fc3(self.dec[1][0](self.dec[0][0](z.reshape(-1, z.shape[-1]))), out_ep=H.EP_SIGMOID)
        x_hat = ops.sigmoid_out(raw)
        d = x_hat.reshape(*lead, *self.data_dim)
        if d.dim() == 5:
            d = d.squeeze(0)
        d = d.permute(0, 3, 1, 2) if d.dim() == 4 else d.permute(0, 1, 4, 2, 3)
        if self.data_dim[-1] == 1:        # a size-1 channel axis: the permute does not reorder memory
            d._lprob_src = (raw, 0)
        return d, self._scale


class Dec_SVHN(VaeDecoder):
    """models/decoders.py:101-147: Linear(D', 128), ReLU, ConvT k4 128->64 (s1 p0), 64->64, 64->32, 32->3 (s2 p1),
    sigmoid, output permuted to (B, 32, 32, 3)"""

    def __init__(self, latent_dim, data_dim, latent_private):
        super().__init__(latent_dim, data_dim, latent_private, net_type=NetworkTypes.CNN)
        self.data_dim = data_dim
        self.net_type = "CNN"
        HC = encoders.HipConv
        self.linear = HipLinear(self.out_dim, 128)
        self.conv1 = HC(128, 64, 4, 1, 0, H.ACT_RELU, transposed=True)
        self.conv2 = HC(64, 64, 4, 2, 1, H.ACT_RELU, transposed=True)
        self.conv3 = HC(64, 32, 4, 2, 1, H.ACT_RELU, transposed=True)
        self.conv4 = HC(32, 3, 4, 2, 1, H.ACT_RELU, transposed=True, out_ep=H.EP_SIGMOID)
        self.register_buffer("_scale", torch.tensor(0.75), persistent=False)

    def forward(self, z):
        zs = z["latents"]
        bs = zs.shape[:2] if (zs.dim() == 3 and zs.shape[0] > 1) else None
        zs = zs.reshape(-1, zs.shape[-1])
        h = self.linear(zs).reshape(-1, 128, 1, 1)
        # sigmoid in conv4's epilogue; `raw` (NCHW) takes the logits' gradient.  The reference returns the output
        # permuted to (B,32,32,3); a lprob loss reads `raw` with that pairing instead of the materialised permutation
        raw = self.conv4(self.conv3(self.conv2(self.conv1(h))), ep_bwd=False)
        d = ops.sigmoid_out(raw).permute(0, 2, 3, 1)
        if bs:
            d = d.reshape(*bs, *d.shape[1:])
        d._lprob_src = (raw, raw.shape[1])
        return d, self._scale
