"""CPU oracle: a from-scratch PyTorch-CPU fp32 restatement of the reference's per-step training path.

TEST INFRASTRUCTURE ONLY.  Nothing under multimodal_vae_comparison_amd/ may import this module; only
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, and only as the checker /
the timed CPU baseline.  The product path has no CPU fallback.

Parity pinning: every function below is checked against golden vectors produced by importing the
reference itself (gabinsane/multimodal-vae-comparison @ 2025-05-23) under tests/golden/ref_harness.py
in the build container (tests/golden/make_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).
The reference's own tests hold no numeric vectors for this path.

Third-party arithmetic restated here (not vendored by the reference): PyTorch's nn.Conv2d /
nn.ConvTranspose2d / nn.Linear / nn.MultiheadAttention / nn.TransformerEncoderLayer /
nn.TransformerDecoderLayer (post-norm, exact-erf GELU, LayerNorm eps 1e-5) / nn.Embedding /
torch.distributions.Normal + kl_divergence.  The reference pins "PyTorch 1.12.1" in prose only
(/root/reference/README.md:48); the fixtures were generated with torch 2.10.0.

All file:line citations are relative to /root/reference/multimodal_compare/.

Precision: every cast in this file goes to torch.get_default_dtype(), so the whole restatement runs in fp64 under
`torch.set_default_dtype(torch.float64)` with fp64 parameters / inputs / noise (tests use that to measure how
ill-conditioned a quantity is: |oracle32 - oracle64| bounds what ANY fp32 implementation can be held to).

Conventions
-----------
* `params`: dict {reference state_dict key: tensor}.  Key names are the reference's, including the
  `.module.` segment that its nn.DataParallel wrappers add (models/decoders.py:58-69).
* `eps`: list of standard-normal tensors consumed in the order the reference draws them
  (one `Normal.rsample` each); the reference's own generator is never used here.
* dropout: the reference trains with p=0.1 dropout (PositionalEncoding + every transformer sub-layer).
  `train` selects how the sites behave: False / None = off (parity mode == reference .eval()); True / "rng" =
  torch CPU dropout at the same sites (only so that the timed CPU baseline does the work of a reference training
  step); a dict {site name: multiplicative mask} = explicit masks (train-mode parity against the HIP path, whose
  counter-based masks are extracted with mmvae_dropout_mask).  Site names: "<prefix>.<enc|dec>.<site>#<call>"
  with sites pe, l0.attn, l0.drop1, l0.ffn, l0.drop2 (encoder) and pe, l0.attn, l0.drop1, l0.xattn, l0.drop2,
  l0.ffn, l0.drop3 (decoder); <call> counts the forward calls of that tower within one objective.
"""
import itertools
import math

import torch
import torch.nn.functional as F

ETA = 1e-6          # utils.py:254  Constants.eta
DROPOUT_P = 0.1     # models/encoders.py:790, models/decoders.py:669 (ctor defaults)
LN_EPS = 1e-5       # torch.nn.LayerNorm default
ENC_TRANSFORMER_LAYERS, DEC_TRANSFORMER_LAYERS, TRANSFORMER_FF = 8, 4, 1024   # encoders.py:659, decoders.py:542
TXTRNN_HIDDEN = 512    # Enc_TxtRNN ctor default hidden_size, encoders.py:841


# ----------------------------------------------------------------------------------------------
# parameter inventory (reference key names)
# ----------------------------------------------------------------------------------------------
def tower_param_shapes(prefix, enc, dec, data_dim, n_latents, private=None):
    """Shapes of every learnable tensor of one VAE, keyed like the reference's state_dict.

    models/vae.py:121-170 (VAE ctor), models/encoders.py:163-200,790-826, models/decoders.py:35-69,668-706.
    """
    Dp = n_latents + (private or 0)
    s = {}
    if enc == "CNN":
        s.update({f"{prefix}.enc.resnet.{k}": v for k, v in resnet50_param_shapes().items()})
        for h in ("mu_layer", "logvar_layer"):
            s[f"{prefix}.enc.{h}.weight"] = (Dp, 1000)
            s[f"{prefix}.enc.{h}.bias"] = (Dp,)
    elif enc == "CNN2":
        s[f"{prefix}.enc.conv1.weight"] = (32, 3, 4, 4)
        s[f"{prefix}.enc.conv1.bias"] = (32,)
        for i in (2, 3, 4):
            s[f"{prefix}.enc.conv{i}.weight"] = (32, 32, 4, 4)
            s[f"{prefix}.enc.conv{i}.bias"] = (32,)
        s[f"{prefix}.enc.lin1.weight"] = (512, 512)
        s[f"{prefix}.enc.lin1.bias"] = (512,)
        s[f"{prefix}.enc.mu_layer.weight"] = (Dp, 512)
        s[f"{prefix}.enc.mu_layer.bias"] = (Dp,)
        s[f"{prefix}.enc.logvar_layer.weight"] = (Dp, 512)
        s[f"{prefix}.enc.logvar_layer.bias"] = (Dp,)
    elif enc == "TxtTransformer":
        feats = data_dim[-1] * data_dim[-2]        # njoints*nfeats, encoders.py:805-815
        d = feats * 2
        L = f"{prefix}.enc.seqTransEncoder.layers.0"
        s[f"{prefix}.enc.embedding.weight"] = (feats, 2)
        s[f"{L}.self_attn.in_proj_weight"] = (3 * d, d)
        s[f"{L}.self_attn.in_proj_bias"] = (3 * d,)
        s[f"{L}.self_attn.out_proj.weight"] = (d, d)
        s[f"{L}.self_attn.out_proj.bias"] = (d,)
        s[f"{L}.linear1.weight"] = (128, d)
        s[f"{L}.linear1.bias"] = (128,)
        s[f"{L}.linear2.weight"] = (d, 128)
        s[f"{L}.linear2.bias"] = (d,)
        for n in ("norm1", "norm2"):
            s[f"{L}.{n}.weight"] = (d,)
            s[f"{L}.{n}.bias"] = (d,)
        s[f"{prefix}.enc.mu_layer.module.weight"] = (Dp, d)
        s[f"{prefix}.enc.mu_layer.module.bias"] = (Dp,)
        s[f"{prefix}.enc.logvar_layer.module.weight"] = (Dp, d)
        s[f"{prefix}.enc.logvar_layer.module.bias"] = (Dp,)
    elif enc == "TxtRNN":
        # Enc_TxtRNN ctor, models/encoders.py:840-852: Embedding(njoints*nfeats, 512), bidirectional one-layer
        # GRU(512, 512) (torch's parameter names), Linear(512, 2 D'); no mu_layer / logvar_layer
        feats, H = data_dim[-1] * data_dim[-2], TXTRNN_HIDDEN
        s[f"{prefix}.enc.embed.weight"] = (feats, H)
        for sfx in ("", "_reverse"):
            s[f"{prefix}.enc.gru.weight_ih_l0{sfx}"] = (3 * H, H)
            s[f"{prefix}.enc.gru.weight_hh_l0{sfx}"] = (3 * H, H)
            s[f"{prefix}.enc.gru.bias_ih_l0{sfx}"] = (3 * H,)
            s[f"{prefix}.enc.gru.bias_hh_l0{sfx}"] = (3 * H,)
        s[f"{prefix}.enc.o2p.weight"] = (2 * Dp, H)
        s[f"{prefix}.enc.o2p.bias"] = (2 * Dp,)
    elif enc == "MNIST":
        # Enc_MNIST ctor, models/encoders.py:226-250: 784 -> 400 -> 400 (ReLU), heads hidden_mu / hidden_logvar
        s[f"{prefix}.enc.enc.0.0.weight"] = (400, 784)
        s[f"{prefix}.enc.enc.0.0.bias"] = (400,)
        s[f"{prefix}.enc.enc.1.0.weight"] = (400, 400)
        s[f"{prefix}.enc.enc.1.0.bias"] = (400,)
        for h in ("hidden_mu", "hidden_logvar"):
            s[f"{prefix}.enc.{h}.weight"] = (Dp, 400)
            s[f"{prefix}.enc.{h}.bias"] = (Dp,)
    elif enc == "SVHN":
        # Enc_SVHN ctor, models/encoders.py:434-456: 4 convs k4 (s2 p1, s2 p1, s2 p1, s2 p0), ReLU
        for i, (co, ci) in enumerate(((32, 3), (64, 32), (64, 64), (128, 64))):
            s[f"{prefix}.enc.conv{i + 1}.weight"] = (co, ci, 4, 4)
            s[f"{prefix}.enc.conv{i + 1}.bias"] = (co,)
        for h in ("hidden_mu", "hidden_logvar"):
            s[f"{prefix}.enc.{h}.weight"] = (Dp, 128)
            s[f"{prefix}.enc.{h}.bias"] = (Dp,)
    elif enc == "Transformer":
        # Enc_Transformer ctor, models/encoders.py:659-700: d_model = out_dim, 8 layers, ff 1024, 2 heads
        feats = data_dim[1] * (data_dim[2] if len(data_dim) > 2 else 1)
        d = Dp
        s[f"{prefix}.enc.mu_layer.module.weight"] = (Dp, Dp)
        s[f"{prefix}.enc.mu_layer.module.bias"] = (Dp,)
        s[f"{prefix}.enc.logvar_layer.module.weight"] = (Dp, Dp)
        s[f"{prefix}.enc.logvar_layer.module.bias"] = (Dp,)
        s[f"{prefix}.enc.skel_Embedding.module.weight"] = (d, feats)
        s[f"{prefix}.enc.skel_Embedding.module.bias"] = (d,)
        for li in range(ENC_TRANSFORMER_LAYERS):
            L = f"{prefix}.enc.seqTransEncoder.layers.{li}"
            s[f"{L}.self_attn.in_proj_weight"] = (3 * d, d)
            s[f"{L}.self_attn.in_proj_bias"] = (3 * d,)
            s[f"{L}.self_attn.out_proj.weight"] = (d, d)
            s[f"{L}.self_attn.out_proj.bias"] = (d,)
            s[f"{L}.linear1.weight"] = (TRANSFORMER_FF, d)
            s[f"{L}.linear1.bias"] = (TRANSFORMER_FF,)
            s[f"{L}.linear2.weight"] = (d, TRANSFORMER_FF)
            s[f"{L}.linear2.bias"] = (d,)
            for n in ("norm1", "norm2"):
                s[f"{L}.{n}.weight"] = (d,)
                s[f"{L}.{n}.bias"] = (d,)
    else:
        raise NotImplementedError(enc)
    if dec == "CNN":
        s[f"{prefix}.dec.lin1.module.weight"] = (512, Dp)
        s[f"{prefix}.dec.lin1.module.bias"] = (512,)
        for n in ("lin2", "lin3"):
            s[f"{prefix}.dec.{n}.module.weight"] = (512, 512)
            s[f"{prefix}.dec.{n}.module.bias"] = (512,)
        for n in ("convT_64", "convT1", "convT2"):
            s[f"{prefix}.dec.{n}.module.weight"] = (32, 32, 4, 4)
            s[f"{prefix}.dec.{n}.module.bias"] = (32,)
        s[f"{prefix}.dec.convT3.module.weight"] = (32, 3, 4, 4)
        s[f"{prefix}.dec.convT3.module.bias"] = (3,)
    elif dec == "TxtTransformer":
        d = Dp
        feats = data_dim[1] * (data_dim[2] if len(data_dim) > 2 else 1)   # decoders.py:683-694
        L = f"{prefix}.dec.seqTransDecoder.layers.0"
        for a in ("self_attn", "multihead_attn"):
            s[f"{L}.{a}.in_proj_weight"] = (3 * d, d)
            s[f"{L}.{a}.in_proj_bias"] = (3 * d,)
            s[f"{L}.{a}.out_proj.weight"] = (d, d)
            s[f"{L}.{a}.out_proj.bias"] = (d,)
        s[f"{L}.linear1.weight"] = (128, d)
        s[f"{L}.linear1.bias"] = (128,)
        s[f"{L}.linear2.weight"] = (d, 128)
        s[f"{L}.linear2.bias"] = (d,)
        for n in ("norm1", "norm2", "norm3"):
            s[f"{L}.{n}.weight"] = (d,)
            s[f"{L}.{n}.bias"] = (d,)
        s[f"{prefix}.dec.finallayer.module.weight"] = (feats, d)
        s[f"{prefix}.dec.finallayer.module.bias"] = (feats,)
    elif dec == "MNIST":
        # Dec_MNIST ctor, models/decoders.py:230-250
        s[f"{prefix}.dec.dec.0.0.weight"] = (400, Dp)
        s[f"{prefix}.dec.dec.0.0.bias"] = (400,)
        s[f"{prefix}.dec.dec.1.0.weight"] = (400, 400)
        s[f"{prefix}.dec.dec.1.0.bias"] = (400,)
        s[f"{prefix}.dec.fc3.weight"] = (784, 400)
        s[f"{prefix}.dec.fc3.bias"] = (784,)
    elif dec == "SVHN":
        # Dec_SVHN ctor, models/decoders.py:101-116: Linear(D', 128), ConvT k4 s1 p0, then 3 x ConvT k4 s2 p1
        s[f"{prefix}.dec.linear.weight"] = (128, Dp)
        s[f"{prefix}.dec.linear.bias"] = (128,)
        for i, (ci, co) in enumerate(((128, 64), (64, 64), (64, 32), (32, 3))):
            s[f"{prefix}.dec.conv{i + 1}.weight"] = (ci, co, 4, 4)
            s[f"{prefix}.dec.conv{i + 1}.bias"] = (co,)
    elif dec == "Transformer":
        # Dec_Transformer ctor, models/decoders.py:542-588: 4 layers, ff 1024, 2 heads
        d = Dp
        feats = data_dim[1] * (data_dim[2] if len(data_dim) > 2 else 1)
        for li in range(DEC_TRANSFORMER_LAYERS):
            L = f"{prefix}.dec.seqTransDecoder.layers.{li}"
            for a in ("self_attn", "multihead_attn"):
                s[f"{L}.{a}.in_proj_weight"] = (3 * d, d)
                s[f"{L}.{a}.in_proj_bias"] = (3 * d,)
                s[f"{L}.{a}.out_proj.weight"] = (d, d)
                s[f"{L}.{a}.out_proj.bias"] = (d,)
            s[f"{L}.linear1.weight"] = (TRANSFORMER_FF, d)
            s[f"{L}.linear1.bias"] = (TRANSFORMER_FF,)
            s[f"{L}.linear2.weight"] = (d, TRANSFORMER_FF)
            s[f"{L}.linear2.bias"] = (d,)
            for n in ("norm1", "norm2", "norm3"):
                s[f"{L}.{n}.weight"] = (d,)
                s[f"{L}.{n}.bias"] = (d,)
        s[f"{prefix}.dec.finallayer.module.weight"] = (feats, d)
        s[f"{prefix}.dec.finallayer.module.bias"] = (feats,)
    else:
        raise NotImplementedError(dec)
    return s


RESNET50_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))     # (planes, blocks, first stride); expansion 4


def resnet50_param_shapes():
    """learnable tensors of the published ResNet-50 (He et al. 2016, v1.5: the stride sits on the 3x3 convolution)
    under torchvision's state_dict names -- what `Enc_CNN.resnet` holds in the reference (models/encoders.py:108)"""
    s = {"conv1.weight": (64, 3, 7, 7), "bn1.weight": (64,), "bn1.bias": (64,)}
    inplanes = 64
    for li, (planes, blocks, _) in enumerate(RESNET50_LAYERS):
        for b in range(blocks):
            pre = f"layer{li + 1}.{b}"
            s[f"{pre}.conv1.weight"] = (planes, inplanes, 1, 1)
            s[f"{pre}.conv2.weight"] = (planes, planes, 3, 3)
            s[f"{pre}.conv3.weight"] = (4 * planes, planes, 1, 1)
            for j, c in ((1, planes), (2, planes), (3, 4 * planes)):
                s[f"{pre}.bn{j}.weight"], s[f"{pre}.bn{j}.bias"] = (c,), (c,)
            if b == 0:
                s[f"{pre}.downsample.0.weight"] = (4 * planes, inplanes, 1, 1)
                s[f"{pre}.downsample.1.weight"], s[f"{pre}.downsample.1.bias"] = (4 * planes,), (4 * planes,)
            inplanes = 4 * planes
    s["fc.weight"], s["fc.bias"] = (1000, 2048), (1000,)
    return s


def model_param_shapes(mods, n_latents):
    """All trainable tensors of a multimodal model.  `mods` = list of dicts with keys
    enc, dec, data_dim, (private).  Trainable prior theta = `_pz_params.1` (models/mmvae_base.py:35-38)."""
    s = {}
    for i, m in enumerate(mods):
        s.update(tower_param_shapes(f"vaes.mod_{i + 1}", m["enc"], m["dec"], m["data_dim"], n_latents,
                                    m.get("private")))
    s["_pz_params.1"] = (1, n_latents)
    return s


# ----------------------------------------------------------------------------------------------
# towers
# ----------------------------------------------------------------------------------------------
def process_output(h, w_mu, b_mu, w_lv, b_lv):
    """models/encoders.py:49-54: mu = linear; "logvar" = softmax(linear, -1) + eta."""
    return F.linear(h, w_mu, b_mu), F.softmax(F.linear(h, w_lv, b_lv), dim=-1) + ETA


def enc_cnn2(p, pre, x):
    """Enc_CNN2.forward, models/encoders.py:202-223: 4x [Conv2d(k4,s2,p1)+SiLU], flatten, Linear, heads."""
    o = x.to(torch.get_default_dtype())
    for i in (1, 2, 3, 4):
        o = F.silu(F.conv2d(o, p[f"{pre}.enc.conv{i}.weight"], p[f"{pre}.enc.conv{i}.bias"], stride=2, padding=1))
    o = o.reshape(o.shape[0], -1)
    o = F.linear(o, p[f"{pre}.enc.lin1.weight"], p[f"{pre}.enc.lin1.bias"])
    return process_output(o, p[f"{pre}.enc.mu_layer.weight"], p[f"{pre}.enc.mu_layer.bias"],
                          p[f"{pre}.enc.logvar_layer.weight"], p[f"{pre}.enc.logvar_layer.bias"])


def resnet50_logits(p, r, x, train=False, stats=None, taps=None):
    """torchvision resnet50(x) -> (B, 1000) with parameters p[f"{r}.<torchvision name>"].  resnet50 is NOT vendored by
    the reference (torchvision, absent in this image; parity with torchvision itself is UNPINNED): the published topology
    is restated here -- conv 7x7/2 (no bias), BatchNorm, ReLU, maxpool 3x3/2, [3, 4, 6, 3] bottlenecks (1x1, 3x3 with the
    stride, 1x1 x4; 1x1 projection shortcut on the first block of a stage), global average pool, Linear(2048, 1000) --
    and PINNED against an independent implementation of the same published architecture, Hugging Face transformers'
    ResNetForImageClassification (v1.5 bottlenecks, 320 state entries like torchvision's), with the same weights in
    float64: tests/test_oracle_resnet_independent.py (logits and every parameter gradient, train and eval mode).
    BatchNorm: batch statistics when `train` (a reference model in .train()), else the running statistics `stats`
    ({key.running_mean / key.running_var}; default 0 / 1 = a freshly constructed module in .eval())."""

    def bn(h, key):
        c = h.shape[1]
        rm = stats[f"{key}.running_mean"] if stats else torch.zeros(c, dtype=h.dtype)
        rv = stats[f"{key}.running_var"] if stats else torch.ones(c, dtype=h.dtype)
        return F.batch_norm(h, None if train else rm, None if train else rv, p[f"{key}.weight"], p[f"{key}.bias"],
                            training=bool(train), momentum=0.1, eps=1e-5)

    h = F.conv2d(x.to(torch.get_default_dtype()), p[f"{r}.conv1.weight"], None, stride=2, padding=3)
    h = F.max_pool2d(_relu(bn(h, f"{r}.bn1"), f"{r}.bn1"), 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(RESNET50_LAYERS):
        for b in range(blocks):
            k = f"{r}.layer{li + 1}.{b}"
            st = stride if b == 0 else 1
            o = _relu(bn(F.conv2d(h, p[f"{k}.conv1.weight"]), f"{k}.bn1"), f"{k}.bn1")
            o = _relu(bn(F.conv2d(o, p[f"{k}.conv2.weight"], None, stride=st, padding=1), f"{k}.bn2"), f"{k}.bn2")
            o = bn(F.conv2d(o, p[f"{k}.conv3.weight"]), f"{k}.bn3")
            idn = h
            if b == 0:
                idn = bn(F.conv2d(h, p[f"{k}.downsample.0.weight"], None, stride=st), f"{k}.downsample.1")
            h = _relu(o + idn, f"{k}.bn3")      # (the block's output: bn3 + identity, rectified)
            if taps is not None:        # (tests) the stage outputs, to localise a mismatch
                h.retain_grad()
                taps.append((k, h))
    h = F.adaptive_avg_pool2d(h, 1).flatten(1)
    return F.linear(h, p[f"{r}.fc.weight"], p[f"{r}.fc.bias"])


def enc_cnn_resnet50(p, pre, x, train=False, stats=None, taps=None):
    """Enc_CNN.forward, models/encoders.py:116-127: silu(resnet50(x)) -> heads (resnet50: resnet50_logits above)"""
    h = F.silu(resnet50_logits(p, f"{pre}.enc.resnet", x, train, stats, taps))
    return process_output(h, p[f"{pre}.enc.mu_layer.weight"], p[f"{pre}.enc.mu_layer.bias"],
                          p[f"{pre}.enc.logvar_layer.weight"], p[f"{pre}.enc.logvar_layer.bias"])


# Test infrastructure for ReLU kinks (tests/test_parity_e2e.py: _relu_masks_of): a pre-activation within fp32 rounding of
# zero takes either side of the kink depending on the summation order, and that ONE element moves whole rows of the
# upstream weight gradients.  With RELU_MASKS = {site: [mask of call 0, call 1, ...]} (site = the name of the layer that
# produces the pre-activation, e.g. "vaes.mod_1.dec.lin2"; the masks are `pre-activation > 0` as the implementation under
# test saw them) every ReLU becomes `x * mask`: the same piecewise-linear branch on both sides, so every gradient can be
# held to the plain 1e-4.  None (the default): torch.relu.
RELU_MASKS = None
# RELU_PRE = {} (tests): every ReLU site appends its pre-activation (detached) per call, `{site: [x of call 0, ...]}` --
# the fp64 run of the oracle records them so that a test can BOUND where the implementation's masks may differ from
# `pre > 0`: only at elements within rounding of the kink (tests/test_parity_e2e.py: _assert_relu_masks_near_kinks).
RELU_PRE = None


def _relu(x, site):
    if RELU_PRE is not None:
        RELU_PRE.setdefault(site, []).append(x.detach().clone())
    if RELU_MASKS is None:
        return torch.relu(x)
    calls = RELU_MASKS.setdefault("_calls", {})
    # a recorded mask may cover n passes of this site at once (the implementation under test decodes a tower's n passes
    # as one batch, rows pass-major: row k * B + b): it then serves the next n calls, pass k first
    pending = RELU_MASKS.setdefault("_pending", {}).setdefault(site, [])
    if not pending:
        k = calls.get(site, 0)
        calls[site] = k + 1
        m = RELU_MASKS[site][k]
        n = m.numel() // x.numel()
        assert n >= 1 and n * x.numel() == m.numel(), (site, tuple(m.shape), tuple(x.shape))
        pending.extend(m.reshape(n, -1).unbind(0))
    return x * pending.pop(0).reshape(x.shape).to(x.dtype)


def dec_cnn(p, pre, z, data_dim=(64, 64, 3)):
    """Dec_CNN.forward, models/decoders.py:71-98.  z (K,B,D') -> recon (K*B, 64,64,3)-*viewed* NCHW memory."""
    if z.dim() == 2:
        z = z.unsqueeze(0)
    K, B = z.shape[0], z.shape[1]
    x = z
    for n in ("lin1", "lin2", "lin3"):
        x = _relu(F.linear(x, p[f"{pre}.dec.{n}.module.weight"], p[f"{pre}.dec.{n}.module.bias"]), f"{pre}.dec.{n}")
    x = x.reshape(B * K, 32, 4, 4)
    for n in ("convT_64", "convT1", "convT2"):
        x = _relu(F.conv_transpose2d(x, p[f"{pre}.dec.{n}.module.weight"], p[f"{pre}.dec.{n}.module.bias"],
                                     stride=2, padding=1), f"{pre}.dec.{n}")
    x = F.conv_transpose2d(x, p[f"{pre}.dec.convT3.module.weight"], p[f"{pre}.dec.convT3.module.bias"],
                           stride=2, padding=1)
    d = torch.sigmoid(x.reshape(K, B, *data_dim)).clamp(ETA, 1 - ETA)   # decoders.py:96-97 (view, no permute)
    return d.reshape(-1, *data_dim)


def positional_table(d_model, n, dtype=None):
    """PositionalEncoding buffer rows [0, n): models/nn_modules.py:422-428.  Returns (n, d_model)."""
    dtype = dtype or torch.get_default_dtype()
    pe = torch.zeros(n, d_model, dtype=dtype)
    position = torch.arange(0, n, dtype=dtype).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).to(torch.get_default_dtype()) * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)[:, : d_model // 2]
    return pe


def _dropout(x, train, name=None):
    if not train:
        return x
    if isinstance(train, dict):
        return x * train[name].reshape(x.shape).to(x.dtype)
    return F.dropout(x, DROPOUT_P, True)


def _tower_call(train, tower):
    """site-name prefix '<tower>.' + call counter for explicit-mask mode"""
    if not isinstance(train, dict):
        return lambda site: None
    calls = train.setdefault("_calls", {})
    k = calls.get(tower, 0)
    calls[tower] = k + 1
    return lambda site: f"{tower}.{site}#{k}"


def mha(q_in, kv_in, w_in, b_in, w_out, b_out, nhead, kpm=None, train=False, name=None):
    """torch.nn.MultiheadAttention forward (seq-first).  q_in (L,N,E), kv_in (S,N,E), kpm (N,S) True=ignore."""
    L, N, E = q_in.shape
    S = kv_in.shape[0]
    hd = E // nhead
    q = F.linear(q_in, w_in[:E], b_in[:E])
    k = F.linear(kv_in, w_in[E:2 * E], b_in[E:2 * E])
    v = F.linear(kv_in, w_in[2 * E:], b_in[2 * E:])
    q = q.reshape(L, N * nhead, hd).transpose(0, 1) * (1.0 / math.sqrt(hd))
    k = k.reshape(S, N * nhead, hd).transpose(0, 1)
    v = v.reshape(S, N * nhead, hd).transpose(0, 1)
    att = torch.bmm(q, k.transpose(1, 2))                      # (N*h, L, S)
    if kpm is not None:
        att = att.reshape(N, nhead, L, S).masked_fill(kpm[:, None, None, :], float("-inf")).reshape(N * nhead, L, S)
    att = _dropout(F.softmax(att, dim=-1), train, name)
    o = torch.bmm(att, v).transpose(0, 1).reshape(L, N, E)
    return F.linear(o, w_out, b_out)


def transformer_encoder_layer(x, p, L, nhead, kpm, train=False, nm=lambda s: None, li=0):
    """torch.nn.TransformerEncoderLayer, norm_first=False, activation gelu (exact)."""
    sa = mha(x, x, p[f"{L}.self_attn.in_proj_weight"], p[f"{L}.self_attn.in_proj_bias"],
             p[f"{L}.self_attn.out_proj.weight"], p[f"{L}.self_attn.out_proj.bias"], nhead, kpm, train, nm(f"l{li}.attn"))
    x = F.layer_norm(x + _dropout(sa, train, nm(f"l{li}.drop1")), x.shape[-1:], p[f"{L}.norm1.weight"],
                     p[f"{L}.norm1.bias"], LN_EPS)
    ff = F.linear(_dropout(F.gelu(F.linear(x, p[f"{L}.linear1.weight"], p[f"{L}.linear1.bias"])), train, nm(f"l{li}.ffn")),
                  p[f"{L}.linear2.weight"], p[f"{L}.linear2.bias"])
    return F.layer_norm(x + _dropout(ff, train, nm(f"l{li}.drop2")), x.shape[-1:], p[f"{L}.norm2.weight"],
                        p[f"{L}.norm2.bias"], LN_EPS)


def transformer_decoder_layer(x, mem, p, L, nhead, tgt_kpm, train=False, nm=lambda s: None, li=0):
    """torch.nn.TransformerDecoderLayer, norm_first=False, activation gelu."""
    sa = mha(x, x, p[f"{L}.self_attn.in_proj_weight"], p[f"{L}.self_attn.in_proj_bias"],
             p[f"{L}.self_attn.out_proj.weight"], p[f"{L}.self_attn.out_proj.bias"], nhead, tgt_kpm, train,
             nm(f"l{li}.attn"))
    x = F.layer_norm(x + _dropout(sa, train, nm(f"l{li}.drop1")), x.shape[-1:], p[f"{L}.norm1.weight"],
                     p[f"{L}.norm1.bias"], LN_EPS)
    ca = mha(x, mem, p[f"{L}.multihead_attn.in_proj_weight"], p[f"{L}.multihead_attn.in_proj_bias"],
             p[f"{L}.multihead_attn.out_proj.weight"], p[f"{L}.multihead_attn.out_proj.bias"], nhead, None, train,
             nm(f"l{li}.xattn"))
    x = F.layer_norm(x + _dropout(ca, train, nm(f"l{li}.drop2")), x.shape[-1:], p[f"{L}.norm2.weight"],
                     p[f"{L}.norm2.bias"], LN_EPS)
    ff = F.linear(_dropout(F.gelu(F.linear(x, p[f"{L}.linear1.weight"], p[f"{L}.linear1.bias"])), train, nm(f"l{li}.ffn")),
                  p[f"{L}.linear2.weight"], p[f"{L}.linear2.bias"])
    return F.layer_norm(x + _dropout(ff, train, nm(f"l{li}.drop3")), x.shape[-1:], p[f"{L}.norm3.weight"],
                        p[f"{L}.norm3.bias"], LN_EPS)


def enc_txt_transformer(p, pre, data, mask, train=False):
    """Enc_TxtTransformer.forward, models/encoders.py:828-837, with PositionalEncoding's two branches
    (models/nn_modules.py:430-438).

    data (B,T,V) one-hot float, mask (B,T) bool or None.  Returns mu, lv (B, D').
    """
    B, T, V = data.shape
    if mask is None:
        mask = torch.ones(B, T, dtype=torch.bool)
    emb = p[f"{pre}.enc.embedding.weight"]
    x = emb[data.long()]                                        # (B,T,V,2): rows 0/1 of the table only
    pe = positional_table(2, 1000)[:B].reshape(B, 1, 2)         # self.pe[:x.shape[0]]  (B,1,2)
    if B == T or B == 1:
        # `x + pe[:B]` broadcasts: pe's first axis lines up with x's T axis; no permute happens
        x = x + pe
    else:
        # except-branch: permute to (T,B,V,2) then add pe[:B] (B,1,2) -> indexed by *batch* index
        x = x.permute(1, 0, 2, 3) + pe
    nm = _tower_call(train, f"{pre}.enc")
    x = _dropout(x.contiguous(), train, nm("pe")).contiguous()
    x = x.view(T, B, -1)                                        # encoders.py:835: nframes, bs
    x = transformer_encoder_layer(x, p, f"{pre}.enc.seqTransEncoder.layers.0", 2, ~mask, train, nm)
    z = x.mean(dim=0)                                           # includes padded steps
    return process_output(z, p[f"{pre}.enc.mu_layer.module.weight"], p[f"{pre}.enc.mu_layer.module.bias"],
                          p[f"{pre}.enc.logvar_layer.module.weight"], p[f"{pre}.enc.logvar_layer.module.bias"])


def dec_txt_transformer(p, pre, z, mask, data_dim=(45, 27, 1), train=False, keep_k=False):
    """Dec_TxtTransformer.forward, models/decoders.py:708-723.  z (K,B,D') is the *memory* (length K): for K > 1 the
    reference attends over the K samples and returns ONE (B,T,V) output, which its loss code then cannot reshape
    (objectives.py:120) -- no objective with K > 1 runs through it.

    keep_k (DEFINED EXTENSION, PARITY UNPINNED for K > 1; identical to the reference for K = 1): every latent sample
    is decoded on its own, as Dec_CNN does by flattening (K,B) into the batch axis (decoders.py:73-76): memory
    (1, K*B, D'), masks repeated K times, output (K*B, T, V) with row k*B + b = sample k of batch element b."""
    if z.dim() == 2:
        z = z.unsqueeze(0)
    if keep_k and z.shape[0] > 1:
        K = z.shape[0]
        z = z.reshape(1, -1, z.shape[-1])
        if mask is not None:
            mask = mask.repeat(K, 1)
    B, D = z.shape[1], z.shape[-1]
    if mask is None:
        mask = torch.ones(B, data_dim[0], dtype=torch.bool)
    T = mask.shape[1]
    tq = torch.zeros(T, B, D) + positional_table(D, T).reshape(T, 1, D)
    nm = _tower_call(train, f"{pre}.dec")
    tq = _dropout(tq, train, nm("pe"))
    out = transformer_decoder_layer(tq, z, p, f"{pre}.dec.seqTransDecoder.layers.0", 2, ~mask, train, nm)
    out = F.linear(out, p[f"{pre}.dec.finallayer.module.weight"], p[f"{pre}.dec.finallayer.module.bias"])
    return out.permute(1, 0, 2) * mask.unsqueeze(-1).to(torch.get_default_dtype())    # (B,T,V), zero at padding


def gru_cell(x_proj, h, w_hh, b_hh):
    """one step of torch.nn.GRU (its documented equations; gate order r, z, n in the stacked weights):
    r = sigma(W_ir x + b_ir + W_hr h + b_hr), z = sigma(W_iz x + b_iz + W_hz h + b_hz),
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), h' = (1 - z) * n + z * h.   x_proj = W_i x + b_i (B, 3H)."""
    H = h.shape[-1]
    gh = F.linear(h, w_hh, b_hh)
    r = torch.sigmoid(x_proj[:, :H] + gh[:, :H])
    z = torch.sigmoid(x_proj[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(x_proj[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def enc_txt_rnn(p, pre, data, mask=None):
    """Enc_TxtRNN.forward, models/encoders.py:854-869, as a DEFINED path -- PARITY UNPINNED: the reference's own forward
    crashes (`self.embed(x.long()).unsqueeze(1)` on the (B,T,V) one-hot batch hands nn.GRU a 5-D tensor -> ValueError;
    SURVEY 0.4), and there is no Dec_TxtRNN.  What its lines spell for a batch of token ids, restated here:
      ids (B,T) = the one-hot rows' token (argmax; an all-zero padding row is token 0 -- the reference never looks at
        `mask`), embedded = Embedding(ids) laid out sequence-first (T,B,512) as nn.GRU's default expects (the
        `.unsqueeze(1)` of :857 is that layout for one sequence);
      output, _ = bidirectional one-layer GRU (dropout 0.1 acts between layers only: none here);
      output[-1] (:862) = the LAST time position: the forward direction's final state after all T steps, and the
        reverse direction's state after its FIRST step (it starts at position T-1 from h = 0);
      sum of the two halves (:864), o2p, chunk(2) -> mu, softmax(logvar) + eta (:866-869).
    The arithmetic of torch.nn.GRU / nn.Embedding is restated from their documented equations (gru_cell) and checked
    against torch.nn.GRU itself in tests/test_oracle_txtrnn.py."""
    B, T, V = data.shape
    ids = data.argmax(-1).t()                                              # (T,B)
    E = p[f"{pre}.enc.embed.weight"]
    x = E[ids]                                                             # (T,B,H)
    H = E.shape[1]
    g = f"{pre}.enc.gru."
    h = torch.zeros(B, H, dtype=E.dtype)
    for t in range(T):
        h = gru_cell(F.linear(x[t], p[g + "weight_ih_l0"], p[g + "bias_ih_l0"]), h, p[g + "weight_hh_l0"], p[g + "bias_hh_l0"])
    hb = gru_cell(F.linear(x[T - 1], p[g + "weight_ih_l0_reverse"], p[g + "bias_ih_l0_reverse"]),
                  torch.zeros(B, H, dtype=E.dtype), p[g + "weight_hh_l0_reverse"], p[g + "bias_hh_l0_reverse"])
    ps = F.linear(h + hb, p[f"{pre}.enc.o2p.weight"], p[f"{pre}.enc.o2p.bias"])
    mu, lv = torch.chunk(ps, 2, dim=1)
    return mu, F.softmax(lv, dim=-1) + ETA


def enc_mnist(p, pre, x):
    """Enc_MNIST.forward, models/encoders.py:252-265"""
    h = x.reshape(x.shape[0], -1).to(torch.get_default_dtype())
    h = _relu(F.linear(h, p[f"{pre}.enc.enc.0.0.weight"], p[f"{pre}.enc.enc.0.0.bias"]), f"{pre}.enc.enc.0.0")
    h = _relu(F.linear(h, p[f"{pre}.enc.enc.1.0.weight"], p[f"{pre}.enc.enc.1.0.bias"]), f"{pre}.enc.enc.1.0")
    return process_output(h, p[f"{pre}.enc.hidden_mu.weight"], p[f"{pre}.enc.hidden_mu.bias"],
                          p[f"{pre}.enc.hidden_logvar.weight"], p[f"{pre}.enc.hidden_logvar.bias"])


def dec_mnist(p, pre, z, data_dim=(28, 28, 1)):
    """Dec_MNIST.forward, models/decoders.py:252-270: (K=1,B,D') -> sigmoid MLP -> reshape (B,28,28,1) -> permute to
    (B,1,28,28)"""
    h = _relu(F.linear(z, p[f"{pre}.dec.dec.0.0.weight"], p[f"{pre}.dec.dec.0.0.bias"]), f"{pre}.dec.dec.0.0")
    h = _relu(F.linear(h, p[f"{pre}.dec.dec.1.0.weight"], p[f"{pre}.dec.dec.1.0.bias"]), f"{pre}.dec.dec.1.0")
    x = torch.sigmoid(F.linear(h, p[f"{pre}.dec.fc3.weight"], p[f"{pre}.dec.fc3.bias"]))
    d = x.reshape(*z.shape[:-1], *data_dim)
    if d.dim() == 5:
        d = d.squeeze(0)
    return d.permute(0, 3, 1, 2) if d.dim() == 4 else d.permute(0, 1, 4, 2, 3)      # K > 1 keeps (K,B,...)


def enc_svhn(p, pre, x):
    """Enc_SVHN.forward, models/encoders.py:458-478"""
    h = x.to(torch.get_default_dtype())
    for i, pad in enumerate((1, 1, 1, 0)):
        h = _relu(F.conv2d(h, p[f"{pre}.enc.conv{i + 1}.weight"], p[f"{pre}.enc.conv{i + 1}.bias"], stride=2, padding=pad),
                  f"{pre}.enc.conv{i + 1}")
    h = h.reshape(h.shape[0], -1)
    return process_output(h, p[f"{pre}.enc.hidden_mu.weight"], p[f"{pre}.enc.hidden_mu.bias"],
                          p[f"{pre}.enc.hidden_logvar.weight"], p[f"{pre}.enc.hidden_logvar.bias"])


def dec_svhn(p, pre, z):
    """Dec_SVHN.forward, models/decoders.py:118-147: the output is permuted to (B,32,32,3) -- the loss then RESHAPES
    the NCHW target to that shape (objectives.py:120), it does not permute it"""
    zs = z.squeeze(0) if z.dim() == 3 else z
    h = _relu(F.linear(zs, p[f"{pre}.dec.linear.weight"], p[f"{pre}.dec.linear.bias"]), f"{pre}.dec.linear").reshape(-1, 128, 1, 1)
    h = _relu(F.conv_transpose2d(h, p[f"{pre}.dec.conv1.weight"], p[f"{pre}.dec.conv1.bias"], stride=1, padding=0), f"{pre}.dec.conv1")
    h = _relu(F.conv_transpose2d(h, p[f"{pre}.dec.conv2.weight"], p[f"{pre}.dec.conv2.bias"], stride=2, padding=1), f"{pre}.dec.conv2")
    h = _relu(F.conv_transpose2d(h, p[f"{pre}.dec.conv3.weight"], p[f"{pre}.dec.conv3.bias"], stride=2, padding=1), f"{pre}.dec.conv3")
    h = F.conv_transpose2d(h, p[f"{pre}.dec.conv4.weight"], p[f"{pre}.dec.conv4.bias"], stride=2, padding=1)
    d = torch.sigmoid(h).permute(0, 2, 3, 1)
    if z.dim() == 3 and z.shape[0] > 1:                 # K > 1: (K,B,32,32,3), decoders.py:119,145-146
        d = d.reshape(z.shape[0], z.shape[1], *d.shape[1:])
    return d


def enc_transformer(p, pre, data, mask, train=False):
    """Enc_Transformer.forward, models/encoders.py:702-729 (ACTOR-style encoder for sequences of (joints, feats)):
    Linear(joints*feats -> d) -> + pe[t] (PositionalEncoding's try-branch: a true time encoding here) -> dropout ->
    8 post-norm encoder layers with src_key_padding_mask = ~mask -> mean over ALL time steps -> heads."""
    x = data
    if x.dim() == 3:
        x = x.unsqueeze(-1)
    B, T = x.shape[0], x.shape[1]
    if mask is None:
        mask = torch.ones(B, T, dtype=torch.bool)
    x = x.permute(1, 0, 2, 3).reshape(T, B, -1).to(torch.get_default_dtype())
    x = F.linear(x, p[f"{pre}.enc.skel_Embedding.module.weight"], p[f"{pre}.enc.skel_Embedding.module.bias"])
    d = x.shape[-1]
    x = x + positional_table(d, T).reshape(T, 1, d)
    nm = _tower_call(train, f"{pre}.enc")
    x = _dropout(x, train, nm("pe"))
    for li in range(ENC_TRANSFORMER_LAYERS):
        x = transformer_encoder_layer(x, p, f"{pre}.enc.seqTransEncoder.layers.{li}", 2, ~mask, train, nm, li)
    z = x.mean(dim=0)
    return process_output(z, p[f"{pre}.enc.mu_layer.module.weight"], p[f"{pre}.enc.mu_layer.module.bias"],
                          p[f"{pre}.enc.logvar_layer.module.weight"], p[f"{pre}.enc.logvar_layer.module.bias"])


def dec_transformer(p, pre, z, mask, data_dim, train=False):
    """Dec_Transformer.forward, models/decoders.py:590-616: memory = z reshaped to (1, K*B, D'); queries = PE(zeros);
    4 post-norm decoder layers; Linear(d -> joints*feats); padded steps zeroed; (B, T, joints, feats)."""
    D = z.shape[-1]
    z = z.reshape(-1, D).unsqueeze(0)
    B = z.shape[1]
    if mask is None:
        mask = torch.ones(B, data_dim[0], dtype=torch.bool)
    elif B > mask.shape[0]:      # K samples per posterior: the mask is repeated K times (decoders.py:603-604)
        mask = mask.repeat(B // mask.shape[0], 1)
    T = mask.shape[1]
    joints = data_dim[1]
    feats = data_dim[2] if len(data_dim) > 2 else 1
    tq = torch.zeros(T, B, D) + positional_table(D, T).reshape(T, 1, D)
    nm = _tower_call(train, f"{pre}.dec")
    out = _dropout(tq, train, nm("pe"))
    for li in range(DEC_TRANSFORMER_LAYERS):
        out = transformer_decoder_layer(out, z, p, f"{pre}.dec.seqTransDecoder.layers.{li}", 2, ~mask, train, nm, li)
    out = F.linear(out, p[f"{pre}.dec.finallayer.module.weight"], p[f"{pre}.dec.finallayer.module.bias"])
    # `output[~mask.T] = 0`: an in-place masked write -- padded steps are exactly 0 AND receive exactly zero gradient
    # (a multiplication by the mask would turn the NaN gradients lprob produces there into NaN * 0 = NaN)
    out = out.reshape(T, B, joints, feats).masked_fill(~mask.T.reshape(T, B, 1, 1), 0.0)
    return out.permute(1, 0, 2, 3)


_ENC = {"CNN2": lambda p, pre, d, train: enc_cnn2(p, pre, d["data"]),
        "CNN": lambda p, pre, d, train: enc_cnn_resnet50(p, pre, d["data"], train),
        "TxtTransformer": lambda p, pre, d, train: enc_txt_transformer(p, pre, d["data"], d["masks"], train),
        "Transformer": lambda p, pre, d, train: enc_transformer(p, pre, d["data"], d["masks"], train),
        "TxtRNN": lambda p, pre, d, train: enc_txt_rnn(p, pre, d["data"], d["masks"]),
        "MNIST": lambda p, pre, d, train: enc_mnist(p, pre, d["data"]),
        "SVHN": lambda p, pre, d, train: enc_svhn(p, pre, d["data"])}


def encode(p, mods, i, inp, train=False):
    return _ENC[mods[i]["enc"]](p, f"vaes.mod_{i + 1}", inp, train)


def decode(p, mods, i, z, mask, train=False, keep_k=False):
    m = mods[i]
    pre = f"vaes.mod_{i + 1}"
    if m["dec"] == "CNN":
        return dec_cnn(p, pre, z, tuple(m["data_dim"]))
    if m["dec"] == "TxtTransformer":
        return dec_txt_transformer(p, pre, z, mask, tuple(m["data_dim"]), train, keep_k)
    if m["dec"] == "Transformer":
        return dec_transformer(p, pre, z, mask, tuple(m["data_dim"]), train)
    if m["dec"] == "MNIST":
        return dec_mnist(p, pre, z, tuple(m["data_dim"]))
    if m["dec"] == "SVHN":
        return dec_svhn(p, pre, z)
    raise NotImplementedError(m["dec"])


# ----------------------------------------------------------------------------------------------
# fusion, KL, reconstruction losses
# ----------------------------------------------------------------------------------------------
def product_of_experts(mu, logvar):
    """TorchMMVAE.product_of_experts, models/mmvae_base.py:203-222.  Returns (mu, VARIANCE)."""
    var = torch.exp(logvar) + 1e-8
    T = 1.0 / var
    return torch.sum(mu * T, dim=0) / torch.sum(T, dim=0), 1.0 / torch.sum(T, dim=0)


def mopoe_subsets(n_mods):
    """MoPOE.set_subsets, models/mmvae_models.py:279-294: non-empty subsets in combinations() order."""
    idx = list(range(n_mods))
    return [c for n in range(1, n_mods + 1) for c in itertools.combinations(idx, n)]


def chunk_bounds(n_comp, n_samples):
    """MoPOE.mixture_component_selection, models/mmvae_models.py:396-410 with uniform float32 weights
    (`:345`, renormalised `:377-378`)."""
    w = (1 / float(n_comp)) * torch.ones(n_comp)
    w = w / w.sum()
    bounds = [0]
    for k in range(n_comp):
        if k == n_comp - 1:
            bounds.append(n_samples)
        else:
            bounds.append(bounds[-1] + int(torch.floor(n_samples * w[k])))
    return bounds


def prior_sigma(theta):
    """pz_params property, models/mmvae_models.py:274-276: softmax(theta, dim=1) * D."""
    return F.softmax(theta, dim=1) * theta.size(-1)


def kl_normal(mu_q, sig_q, mu_p, sig_p):
    """torch.distributions.kl._kl_normal_normal (called from utils.py:399-402)."""
    var_ratio = (sig_q / sig_p).pow(2)
    t1 = ((mu_q - mu_p) / sig_p).pow(2)
    return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())


def recon_bce(x_hat, target):
    """ReconLoss.bce, models/objectives.py:392-406; target reshaped to the output's (viewed) shape
    (`:120`).  Returns (B, F) positive loss."""
    B = target.shape[0]
    return F.binary_cross_entropy(x_hat, target.to(torch.get_default_dtype()).reshape(x_hat.shape).detach(), reduction="none").reshape(B, -1)


def recon_category_ce(logits, target):
    """ReconLoss.category_ce, models/objectives.py:486-500: CrossEntropyLoss over dim 1 (= time) with
    probability targets.  logits/target (B,T,V) -> (B,V)."""
    lsm = F.log_softmax(logits, dim=1)
    return -(target.to(torch.get_default_dtype()).detach() * lsm).sum(dim=1)


PX_SCALE = 0.75     # the decoders' second return value: Normal(loc, 0.75), models/decoders.py:98,616,723


def recon_lprob(loc, target, scale=None, laplace=False):
    """ReconLoss.lprob, models/objectives.py:409-424: -log p(target) under Normal (or Laplace)(loc, scale) evaluated in
    fp32 by torch.distributions, THEN cast to float64, NaN -> 0 (an in-place masked write: those elements carry no
    gradient).  scale = 0.75, or `loc` itself when the modality has masks (`output.scale = output.loc`, :43-45)."""
    B = target.shape[0]
    t = target.to(torch.get_default_dtype()).reshape(loc.shape).detach()
    sc = torch.full_like(loc, PX_SCALE) if scale is None else scale
    if laplace:
        lp = -torch.log(2 * sc) - torch.abs(t - loc) / sc                     # torch.distributions.Laplace.log_prob
    else:
        lp = -((t - loc) ** 2) / (2 * sc ** 2) - sc.log() - math.log(math.sqrt(2 * math.pi))   # Normal.log_prob
    out = lp.reshape(B, -1).double()
    out = torch.where(torch.isnan(out), torch.zeros_like(out), out)
    return -out


def softclip(t, lo):
    """utils.softclip, utils.py:66-69"""
    return lo + F.softplus((t - lo).to(torch.get_default_dtype()))


def recon_optimal_sigma(loc, target):
    """ReconLoss.optimal_sigma, models/objectives.py:503-509 (sigma-VAE): ONE log sigma = softclip(log sqrt(mean over
    every element of (t - x)^2), -6) per call; the squared term is `.clone().detach()`-ed, so the only gradient path
    into the decoder is through log sigma."""
    B = target.shape[0]
    t = target.to(torch.get_default_dtype()).reshape(loc.shape).detach()
    log_sigma = softclip(((t - loc) ** 2).mean().sqrt().log(), -6.0)
    sq = (((t - loc) / log_sigma.exp()) ** 2).detach()
    return (sq + log_sigma + 0.5 * math.log(2 * math.pi)).reshape(B, -1)


def recon_l1(x, target):
    """ReconLoss.l1, models/objectives.py:427-442: torch.nn.L1Loss(reduction="none")"""
    return (x - target.to(torch.get_default_dtype()).reshape(x.shape).detach()).abs().reshape(target.shape[0], -1)


def recon_mse(x, target):
    """ReconLoss.mse, models/objectives.py:444-459: torch.nn.MSELoss(reduction="none")"""
    return ((x - target.to(torch.get_default_dtype()).reshape(x.shape).detach()) ** 2).reshape(target.shape[0], -1)


_RECON = {"bce": lambda o, t, sc, lap=False: recon_bce(o, t), "category_ce": lambda o, t, sc, lap=False: recon_category_ce(o, t),
          "lprob": lambda o, t, sc, lap=False: recon_lprob(o, t, sc, lap),
          "optimal_sigma": lambda o, t, sc, lap=False: recon_optimal_sigma(o, t),
          "l1": lambda o, t, sc, lap=False: recon_l1(o, t), "mse": lambda o, t, sc, lap=False: recon_mse(o, t)}


def recon_loss(ltype, out, target, laplace=False):
    """BaseObjective.recon_loss_fn, models/objectives.py:30-52: slice to the mask length (and, with masks, the
    likelihood's scale becomes its loc), positive loss.  laplace: the likelihood is `vae.px_z` = Laplace (the config's
    `prior: laplace`, models/trainer.py:104; only `lprob` looks at the family)."""
    scale = None
    if target["masks"] is not None:
        out = out[:, : target["masks"].shape[1]]
        scale = out
    return _RECON[ltype](out, target["data"], scale, laplace)


# ----------------------------------------------------------------------------------------------
# objectives
# ----------------------------------------------------------------------------------------------
def mopoe_forward(p, mods, batch, eps, n_latents, train=False):
    """MoPOE.modality_mixing + forward, models/mmvae_models.py:322-370 (all modalities present)."""
    M = len(mods)
    enc = [encode(p, mods, i, batch[f"mod_{i + 1}"], train) for i in range(M)]
    B = enc[0][0].shape[0]
    subs = mopoe_subsets(M)
    s_mu, s_var = [], []
    for S in subs:
        mus = torch.stack([enc[i][0] for i in S])
        lvs = torch.stack([enc[i][1] for i in S])
        if len(S) == M:                                          # prior expert only in the full subset, :386-389
            mus = torch.cat((mus, torch.zeros(1, B, n_latents)), dim=0)
            lvs = torch.cat((lvs, torch.zeros(1, B, n_latents)), dim=0)
        m_, v_ = product_of_experts(mus, lvs)
        s_mu.append(m_)
        s_var.append(v_)
    # moe_fusion / mixture_component_selection, :380-410.  The reference stacks the subset posteriors as
    # (n_subsets, 1, B, D) (poe_fusion unsqueezes, then `:343` unsqueezes again), so the "batch-chunk
    # selection" runs over the SINGLETON axis 1: num_samples == 1, every chunk but the last is empty and the
    # joint posterior is the LAST subset (= PoE of all modalities + prior expert) for every sample
    # [verified on the reference, tests/golden/make_golden.py].  Restated literally:
    mus = torch.stack([m_.unsqueeze(0) for m_ in s_mu])          # (n, 1, B, D)
    vrs = torch.stack([v_.unsqueeze(0) for v_ in s_var])
    bounds = chunk_bounds(mus.shape[0], mus.shape[1])
    j_mu = torch.cat([mus[k, bounds[k]:bounds[k + 1], :] for k in range(len(subs))]).squeeze(0)
    j_var = torch.cat([vrs[k, bounds[k]:bounds[k + 1], :] for k in range(len(subs))]).squeeze(0)
    zs, recons = [], []
    for i in range(M):
        z = j_mu + j_var * eps[i].reshape(-1, B, n_latents)      # Normal(mu, VAR-as-sigma).rsample([K]), :365-366
        zs.append(z)
        recons.append(decode(p, mods, i, z, batch[f"mod_{i + 1}"]["masks"], train))
    return {"enc": enc, "subsets": dict(zip(subs, zip(s_mu, s_var))), "bounds": bounds,
            "joint": (j_mu, j_var), "z": zs, "recon": recons}


def mopoe_objective(p, mods, batch, eps, n_latents, beta=1.0, train=False, prior="normal", **_):
    """MoPOE.objective, models/mmvae_models.py:296-320 + weighted_group_kld, models/objectives.py:184-201."""
    M = len(mods)
    fw = mopoe_forward(p, mods, batch, eps, n_latents, train)
    sig_p = prior_sigma(p["_pz_params.1"])
    dists = list(fw["enc"]) + [fw["joint"]]
    klds = [kl_normal(mu, sig, 0.0, sig_p) for (mu, sig) in dists]
    w = (1 / len(dists)) * torch.ones(len(dists))
    group = (torch.stack(klds).sum(-1).mean(1) * w).sum()
    lpx_zs = []
    for i in range(M):
        scale = float(mods[i].get("llik_scaling", 1.0))
        # `prior: laplace`: MoPOE.forward hard-codes dist.Normal posteriors (mmvae_models.py:363-365); only the
        # likelihood `vae.px_z` (:369) follows the config -- a Laplace log-prob for recon_loss lprob
        lpx = -recon_loss(mods[i]["ltype"], fw["recon"][i], batch[f"mod_{i + 1}"], prior == "laplace") * scale
        lpx_zs.append(lpx.sum(-1))
    lpx = torch.stack(lpx_zs).sum(0).mean()
    loss = -(lpx - beta * group)
    rec = [-m / float(mods[i].get("llik_scaling", 1.0)) for i, m in enumerate(lpx_zs)]
    return {"loss": loss, "kld": group, "reconstruction_loss": rec, "_fw": fw, "_klds": klds}


def poe_subset_order(n_mods, order=None):
    """utils.subsample_input_modalities, utils.py:86-112.  Within one subset size the reference iterates
    a Python `set` of tuples of str => order depends on PYTHONHASHSEED; pass the recorded order."""
    if order is not None:
        return [tuple(o) for o in order]
    idx = list(range(n_mods))
    return [c for n in range(1, n_mods + 1) for c in itertools.combinations(idx, n)]


def poe_objective(p, mods, batch, eps, n_latents, beta=1.0, order=None, train=False, prior="normal", **_):
    """POE.objective / forward / modality_mixing, models/mmvae_models.py:159-232."""
    M = len(mods)
    B = next(batch[f"mod_{i + 1}"]["data"].shape[0] for i in range(M) if batch[f"mod_{i + 1}"]["data"] is not None)
    sig_p = prior_sigma(p["_pz_params.1"])
    lpx_rec = [[] for _ in range(M)]
    klds, losses, inter = [], [], []
    for s_idx, S in enumerate(poe_subset_order(M, order)):
        mus = [torch.zeros(1, B, n_latents)]                     # prior expert: mu 0, logvar log(1), :222,246-247
        lvs = [torch.zeros(1, B, n_latents)]
        for i in range(M):
            if i in S:
                m_, l_ = encode(p, mods, i, batch[f"mod_{i + 1}"], train)
                mus.append(m_.unsqueeze(0))
                lvs.append(l_.unsqueeze(0))
        j_mu, j_var = product_of_experts(torch.cat(mus), torch.cat(lvs))
        z = j_mu + j_var * eps[s_idx].reshape(-1, B, n_latents)  # ONE shared rsample, :200-201
        kld = kl_normal(j_mu, j_var, 0.0, sig_p).sum(-1)
        klds.append(kld)
        loc = []
        for i in range(M):
            mask = batch[f"mod_{i + 1}"]["masks"] if i in S else None   # absent modality: masks None, utils.py:104-106
            rec = decode(p, mods, i, z, mask, train)
            # (POE.forward: qz_x = dist.Normal hard-coded :200, px = vae.px_z :204 -- the config's `prior` family)
            lpx = (-recon_loss(mods[i]["ltype"], rec, batch[f"mod_{i + 1}"], prior == "laplace") * float(mods[i].get("llik_scaling", 1.0))).sum(-1)
            loc.append(lpx)
            if i == s_idx:
                lpx_rec[i].append(lpx)
        lpx_sum = torch.stack(loc).sum(0)
        losses.append(-(lpx_sum.sum(-1) - beta * kld.sum()).sum())
        inter.append({"joint": (j_mu, j_var), "z": z})
    rec = [-torch.stack(m).sum() / float(mods[i].get("llik_scaling", 1.0)) for i, m in enumerate(lpx_rec)]
    return {"loss": torch.stack(losses).sum(), "reconstruction_loss": rec,
            "kld": torch.stack(klds).mean(0).sum(), "_inter": inter}


def normal_log_prob(z, mu, sigma):
    """torch.distributions.Normal.log_prob"""
    return -((z - mu) ** 2) / (2 * sigma ** 2) - sigma.log() - math.log(math.sqrt(2 * math.pi))


def laplace_log_prob(z, mu, scale):
    """torch.distributions.Laplace.log_prob"""
    return -torch.log(2 * scale) - torch.abs(z - mu) / scale


def kl_laplace_normal(mu_q, scale_q, mu_p, sig_p):
    """torch.distributions.kl._kl_laplace_normal: KL(Laplace(mu_q, scale_q) || Normal(mu_p, sig_p)), reached through
    utils.kl_divergence (utils.py:399-402) when the config says `prior: laplace` -- the POSTERIOR is then a Laplace
    (models/trainer.py:104) while MOE.objective still builds its prior with the model-level `self.pz` = Normal
    (models/mmvae_base.py:31, models/mmvae_models.py:45)"""
    var_n = sig_p ** 2
    ratio = scale_q ** 2 / var_n
    t1 = 0.5 * torch.log(2 * ratio / math.pi)
    return -t1 + ratio + (0.5 * mu_q ** 2 - mu_q * mu_p + 0.5 * mu_p ** 2) / var_n - 1


def resolve_llik_scaling(mods):
    """TorchMMVAE.set_likelihood_scales, models/mmvae_base.py:41-47: "auto" -> min_m prod(data_dim) / prod(data_dim_m)"""
    dims = [float(torch.tensor(m["data_dim"]).prod()) for m in mods]
    return [min(dims) / d if m.get("llik_scaling", 1.0) == "auto" else float(m.get("llik_scaling", 1.0))
            for m, d in zip(mods, dims)]


def recon_lprob_k(loc, target, K, laplace=False):
    """recon_loss_fn + reshape_for_loss + ReconLoss.lprob for a K-sample decoder output (objectives.py:30-52,103-125,
    409-424): the target is repeated K times and reshaped like loc (K,B,...) -- or (B,...) when the decoder squeezed
    K = 1 away --, bs = loc.shape[0]; returns the positive fp64 loss of shape (loc.shape[0], -1)."""
    t = target.to(torch.get_default_dtype()).repeat(K, *([1] * (target.dim() - 1))).reshape(loc.shape).detach()
    sc = torch.full_like(loc, PX_SCALE)
    lp = laplace_log_prob(t, loc, sc) if laplace else normal_log_prob(t, loc, sc)
    out = lp.reshape(loc.shape[0], -1).double()
    return -torch.where(torch.isnan(out), torch.zeros_like(out), out)


def moe_dreg_objective(p, mods, batch, eps, n_latents, K, prior="normal", train=False):
    """MOE.forward + objective with obj "dreg" (the shipped configs/config_mnistsvhn.yml: K = 30, prior laplace,
    llik_scaling auto), models/mmvae_models.py:32-117 + MultimodalObjective.dreg / _m_dreg_looser
    (models/objectives.py:361-387).  Two modalities, towers that keep the K axis (MNIST / SVHN), `lprob`.

      q_m = Normal | Laplace(mu_m, scale = lv_m) per the config's `prior`; z_m = q_m.rsample([K]) (K,B,D).
      own reconstruction: Normal(dec_m(z_m), 0.75) (dist.Normal, :101-103); cross reconstruction of modality t from
      the other modality's z: `vae.px_z` = the config's `prior` class again (:115).
      lpx (K,) = sum over batch AND features of log p(x) * llik_scaling (the decoders keep K, so recon_loss_fn's
      bs is K and `.view(batch_shape[:1], -1).sum(-1)` sums the batch away, :47-48,64-70).
      lw_r (K,) = sum_b log N(z_r; 0, softmax(theta) D) + lpx_own_r + lpx_cross_r
                  - sum_b log-mean-exp_m log q_m(z_r)                               (objectives.py:366-373)
      grad_wt = softmax_K(lw) detached; the hook of objectives.py:382-383 sits on a fresh torch.stack(zss) that
      nothing consumes, so it never fires.  loss = -(grad_wt * lw).mean(0).sum();  kld = tensor(0);
      reconstruction_loss = (M, 2, K) [own, cross]."""
    M = len(mods)
    assert M == 2, "MOE dreg: the reference's cross-term indexing (mmvae_models.py:64-70) is only meaningful for M = 2"
    lap = prior == "laplace"
    logq = laplace_log_prob if lap else normal_log_prob
    enc = [encode(p, mods, i, batch[f"mod_{i + 1}"], train) for i in range(M)]
    B = enc[0][0].shape[0]
    zs = [enc[i][0] + enc[i][1] * eps[i].reshape(K, B, n_latents) for i in range(M)]            # (K,B,D)
    lam = resolve_llik_scaling(mods)
    sig_p = prior_sigma(p["_pz_params.1"])
    lws, recs = [], []
    for r in range(M):
        o = 1 - r
        tgt = batch[f"mod_{r + 1}"]["data"]
        own = decode(p, mods, r, zs[r], None, train)
        cross = decode(p, mods, r, zs[o], None, train)
        lpx_own = (-recon_lprob_k(own, tgt, K, False) * lam[r]).sum(-1)
        lpx_cross = (-recon_lprob_k(cross, tgt, K, lap) * lam[r]).sum(-1)
        lpz = normal_log_prob(zs[r], 0.0, sig_p).sum(-1)                                       # (K,B)
        lq = torch.stack([logq(zs[r], enc[m][0], enc[m][1]).sum(-1) for m in range(M)])        # (M,K,B)
        lqz = torch.logsumexp(lq, 0) - math.log(M)
        lws.append(lpz.sum(-1) + (lpx_own + lpx_cross) - lqz.sum(-1))
        recs.append(torch.stack([lpx_own, lpx_cross]))
    lw = torch.stack(lws)                                                                      # (M,K) fp64
    with torch.no_grad():
        grad_wt = (lw - torch.logsumexp(lw, 1, keepdim=True)).exp()
    return {"loss": -(grad_wt * lw).mean(0).sum(), "kld": torch.tensor(0), "reconstruction_loss": torch.stack(recs),
            "_lw": lw, "_z": zs, "_enc": enc}


def recon_loss_k(ltype, out, target, K, laplace=False):
    """BaseObjective.recon_loss_fn with K samples (objectives.py:30-52 + reshape_for_loss :103-125): the output is
    sliced to the mask length (then scale := loc for lprob), the target is repeated K times along its first axis and
    reshaped like the output, bs = the output's leading axis; positive loss (out.shape[0], -1)."""
    masked = target["masks"] is not None
    if masked:
        out = out[:, : target["masks"].shape[1]]
    data = target["data"]
    t = data.to(torch.get_default_dtype()).repeat(K, *([1] * (data.dim() - 1))).reshape(out.shape).detach()
    if ltype == "lprob":
        if masked:
            return recon_lprob(out, t, out, laplace)
        return recon_lprob_k(out, data, K, laplace)
    return _RECON[ltype](out, t, None)


def moe_iwae_objective(p, mods, batch, eps, n_latents, K, prior="normal", train=False, beta=1.0):
    """MOE.objective's non-elbo branch + MultimodalObjective.iwae, models/mmvae_models.py:63-78 and
    models/objectives.py:342-359, LITERALLY (tests/golden/moe_*_iwae_*.npz, generated from the reference under the
    tuple-`.cuda()` shim of tests/golden/ref_harness.py, pin it):
      lpx_r = (own, cross) reconstruction sums over everything but the decoders' LEADING axis (rows n0), times llik_scaling;
      lpz (K,B) = sum_d log N(z_r; 0, softmax(theta) D);  lqz (K,B) = log-mean-exp_m sum_d log q_m(z_r);
      lw_r = lpz + (lpx_own + lpx_cross).reshape(K, B) - beta * lqz                                    (:356)
      loss = - sum_b log-mean-exp over cat_r(lw_r) along dim 0 (the M K weights of sample b);  kld = tensor(0);
      reconstruction_loss = (M, 2, n0).
    The reshape at :356 needs n0 == K * B.  The reference's towers give n0 = K (MNIST / SVHN keep K and recon_loss_fn's
    batch axis is then K), n0 = B (any tower at K = 1) or crash in the text decoder (K > 1): so it runs at B = 1 or at
    K = 1 only.  EXTENSION (parity unpinned beyond those two edges, which the fixtures pin): the per-(k, b) row sums are
    used for every K and B -- K-preserving towers are summed per sample instead of per k, and the text decoder decodes
    every sample (dec_txt_transformer keep_k).  fp64 where a `lprob` term is (the reference's .double()), else fp32."""
    M = len(mods)
    assert M == 2, "MOE iwae: the reference's cross-term indexing (mmvae_models.py:64-70) is only meaningful for M = 2"
    lap = prior == "laplace"
    logq = laplace_log_prob if lap else normal_log_prob
    enc = [encode(p, mods, i, batch[f"mod_{i + 1}"], train) for i in range(M)]
    B = enc[0][0].shape[0]
    zs = [enc[i][0] + enc[i][1] * eps[i].reshape(K, B, n_latents) for i in range(M)]            # (K,B,D)
    lam = resolve_llik_scaling(mods)
    sig_p = prior_sigma(p["_pz_params.1"])
    lws, recs = [], []
    for r in range(M):
        o = 1 - r
        tgt = batch[f"mod_{r + 1}"]
        own = decode(p, mods, r, zs[r], tgt["masks"], train, keep_k=True)
        cross = decode(p, mods, r, zs[o], tgt["masks"], train, keep_k=True)
        # own: dist.Normal (:101-103); cross: vae.px_z = the config's `prior` family (:115)
        lpx_own = (-recon_loss_k(mods[r]["ltype"], own, tgt, K, False) * lam[r]).reshape(K * B, -1).sum(-1)
        lpx_cross = (-recon_loss_k(mods[r]["ltype"], cross, tgt, K, lap) * lam[r]).reshape(K * B, -1).sum(-1)
        lpz = normal_log_prob(zs[r], 0.0, sig_p).sum(-1)                                       # (K,B)
        lq = torch.stack([logq(zs[r], enc[m][0], enc[m][1]).sum(-1) for m in range(M)])        # (M,K,B)
        lqz = torch.logsumexp(lq, 0) - math.log(M)
        lws.append(lpz + (lpx_own + lpx_cross).reshape(K, B) - beta * lqz)
        recs.append(torch.stack([lpx_own, lpx_cross]))
    lw = torch.cat(lws)                                                                        # (M K, B)
    loss = -(torch.logsumexp(lw, 0) - math.log(lw.shape[0])).sum()
    return {"loss": loss, "kld": torch.tensor(0), "reconstruction_loss": torch.stack(recs), "_lw": lw, "_z": zs,
            "_enc": enc}


def moe_objective(p, mods, batch, eps, n_latents, beta=1.0, train=False, obj="elbo", K=1, prior="normal"):
    """MOE.forward + objective (obj "elbo", K = 1), models/mmvae_models.py:32-117.

    q_m = Normal(mu_m, lv_m as sigma), z_m = rsample; own decode; cross decode of every target from the LAST other
    source in iteration order (dict overwrite, :112-116); KL against the per-VAE fixed prior N(0, 1)
    (`vae._pz_params`, models/vae.py:159-162; :45); importance weight exp(log q_r(z_o) - log q_o(z_o).detach()) on
    the cross terms with z_o detached (:56-62); rows [own_1, w_1 cross_1, own_2, ...]; loss = elbo / M (:73-77).
    """
    if obj == "dreg":
        return moe_dreg_objective(p, mods, batch, eps, n_latents, K, prior, train)
    if obj == "iwae":
        return moe_iwae_objective(p, mods, batch, eps, n_latents, K, prior, train, beta)
    assert obj == "elbo" and K == 1
    M = len(mods)
    lap = prior == "laplace"      # posterior Laplace(mu, scale = lv), cross likelihood Laplace, KL(Laplace || N(0,1))
    enc = [encode(p, mods, i, batch[f"mod_{i + 1}"], train) for i in range(M)]
    B = enc[0][0].shape[0]
    zs = [enc[i][0] + enc[i][1] * eps[i].reshape(-1, B, n_latents) for i in range(M)]     # (1,B,D)
    lam = resolve_llik_scaling(mods)
    rows, klds = [], []
    own = [decode(p, mods, i, zs[i], batch[f"mod_{i + 1}"]["masks"], train) for i in range(M)]
    cross_src = [[s for s in range(M) if s != t][-1] for t in range(M)]
    cross = [decode(p, mods, t, zs[cross_src[t]], batch[f"mod_{t + 1}"]["masks"], train) for t in range(M)]
    for r in range(M):
        mu, sig = enc[r]
        one = torch.ones(1, n_latents)
        klds.append((kl_laplace_normal(mu, sig, 0.0 * one, one) if lap else kl_normal(mu, sig, 0.0, one)).sum(-1))
        lpx_own = (-recon_loss(mods[r]["ltype"], own[r], batch[f"mod_{r + 1}"]) * lam[r]).sum(-1)
        if lap:      # the cross reconstruction's likelihood is `vae.px_z` = Laplace (mmvae_models.py:115); lprob only
            assert mods[r]["ltype"] == "lprob" and batch[f"mod_{r + 1}"]["masks"] is None
            lpx_cross = (-recon_lprob(cross[r], batch[f"mod_{r + 1}"]["data"], None, True) * lam[r]).sum(-1)
        else:
            lpx_cross = (-recon_loss(mods[r]["ltype"], cross[r], batch[f"mod_{r + 1}"]) * lam[r]).sum(-1)
        o = cross_src[r]
        z_o = zs[o].detach()
        logq = laplace_log_prob if lap else normal_log_prob
        lwt = (logq(z_o, mu, sig) - logq(z_o, enc[o][0], enc[o][1]).detach()).sum(-1).reshape(-1)
        rows.append(lpx_own)
        rows.append(lwt.exp() * lpx_cross)
    # `lp.sum() != 0` filter (:73): rows whose importance weight underflowed to exactly 0 are DROPPED, which also
    # changes how many times beta * kld.sum() is subtracted by the broadcast in BaseObjective.elbo
    lpx = torch.stack([lp for lp in rows if lp.sum() != 0])
    kld = torch.stack(klds)
    loss = -(lpx.sum(-1) - beta * kld.sum()).sum() / M
    return {"loss": loss, "kld": kld, "reconstruction_loss": lpx, "_rows": torch.stack(rows), "_z": zs, "_enc": enc}


def moe_forward(p, mods, batch, eps, n_latents, train=False):
    """MOE.forward (K = 1) with missing modalities, models/mmvae_models.py:80-117 -- what `save_reconstructions`
    (models/trainer.py:179-215) calls for cross-generation.  `batch[m]["data"]` is None for a missing modality (its
    masks stay); `eps`: one recorded draw per PRESENT modality, in modality order.

    Present m: q_m = Normal(mu, lv as sigma), z_m = rsample, px_m = dec_m(z_m, masks_m).  Missing m (:105-108):
    zs[m] IS the dict of the first present modality f -- the same object, whose masks are overwritten with masks_m --
    and px_m = dec_m(z_f, masks_m).  Cross (:109-114): for every source s in dict order and every target t != s,
    cross[t] = {s: dec_t(z_s, masks_t)} -- a fresh one-entry dict, so the LAST source wins.
    Returns (q [None for missing], z [per modality], px, cross {t: (s, loc)})."""
    M = len(mods)
    present = [i for i in range(M) if batch[f"mod_{i + 1}"]["data"] is not None]
    assert present, "at least one modality must be present for forward call"
    q, z = [None] * M, [None] * M
    for j, i in enumerate(present):
        mu, sig = encode(p, mods, i, batch[f"mod_{i + 1}"], train)
        q[i] = (mu, sig)
        z[i] = mu + sig * eps[j].reshape(-1, mu.shape[0], n_latents)
    f = present[0]
    for i in range(M):
        if z[i] is None:
            z[i] = z[f]
    masks = [batch[f"mod_{i + 1}"]["masks"] for i in range(M)]
    px = [decode(p, mods, i, z[i], masks[i], train) for i in range(M)]
    cross = {}
    for s in range(M):
        for t in range(M):
            if t != s:
                cross[t] = (s, decode(p, mods, t, z[s], masks[t], train))
    return q, z, px, cross


def dmvae_objective(p, mods, batch, eps, n_latents, beta=1.0, train=False, prior="normal", **_):
    """DMVAE.forward + objective, models/mmvae_models.py:436-503 (all modalities present, K = 1).

    Encoder outputs (B, D+P) are split into shared [:D] / private [D:] AFTER the softmax over all D+P columns
    (mmvae_base.py:152-156); joint = product of the shared experts WITHOUT a prior expert (:478-479); draws in the
    reference's order: z_joint, then per modality z_shared, z_private and one fresh shared draw of every other
    modality for the cross reconstruction (:486-502); three ELBOs per modality (:458-459):
      -(sum lpx_own - beta sum KL(q_shared || p)) - (sum lpx_joint - beta sum KL(joint || p))
      -(sum lpx_cross - beta sum_over_cross KL(q_private || N(0,1)))        p = N(0, softmax(theta) * D)."""
    M, D = len(mods), n_latents
    enc = [encode(p, mods, i, batch[f"mod_{i + 1}"], train) for i in range(M)]
    B = enc[0][0].shape[0]
    sh = [(mu[:, :D], lv[:, :D]) for mu, lv in enc]
    pr = [(mu[:, D:], lv[:, D:]) for mu, lv in enc]
    j_mu, j_var = product_of_experts(torch.stack([m for m, _ in sh]), torch.stack([l for _, l in sh]))
    sig_p = prior_sigma(p["_pz_params.1"])
    it = iter(eps)
    nxt = lambda d: next(it).reshape(1, B, d)
    z_joint = j_mu + j_var * nxt(D)
    kld_joint = kl_normal(j_mu, j_var, 0.0, sig_p)
    losses, ind, klds, inter = [], [], [], []
    for i in range(M):
        lam = float(mods[i].get("llik_scaling", 1.0))
        tgt = batch[f"mod_{i + 1}"]
        mask = tgt["masks"]
        z_sh = sh[i][0] + sh[i][1] * nxt(D)
        z_pr = pr[i][0] + pr[i][1] * nxt(pr[i][0].shape[1])
        # (DMVAE.forward: posteriors from the model-level self.qz_x = Normal :480-485, likelihoods vae.px_z :495-501)
        rec = lambda z: (-recon_loss(mods[i]["ltype"], decode(p, mods, i, torch.cat([z, z_pr], -1), mask, train), tgt, prior == "laplace") * lam).sum(-1)
        lpx_own = rec(z_sh)
        lpx_joint = rec(z_joint)
        lpx_cross, kl_priv = [], []
        for m in range(M):
            if m != i:
                z_c = sh[m][0] + sh[m][1] * nxt(D)
                lpx_cross.append(rec(z_c))
                kl_priv.append(kl_normal(pr[i][0], pr[i][1], 0.0, torch.ones(1, pr[i][0].shape[1])))
        kld = kl_normal(sh[i][0], sh[i][1], 0.0, sig_p)
        e1 = -(lpx_own.sum(-1) - beta * kld.sum(-1).sum()).sum()
        e2 = -(lpx_joint.sum(-1) - beta * kld_joint.sum()).sum()
        e3 = -(torch.stack(lpx_cross).sum() - beta * torch.stack(kl_priv).sum(-1).sum())
        losses.append(e1 + e2 + e3)
        ind.append(lpx_own)
        klds.append(kld)
        inter.append({"z_shared": z_sh, "z_private": z_pr})
    rec_out = [-(m).sum() / float(mods[i].get("llik_scaling", 1.0)) for i, m in enumerate(ind)]
    return {"loss": torch.stack(losses).sum(), "reconstruction_loss": rec_out, "kld": torch.stack(klds).mean(0).sum(),
            "_joint": (j_mu, j_var), "_inter": inter}


def dmvae_forward(p, mods, batch, eps, n_latents, train=False):
    """DMVAE.forward (K = 1) with missing modalities, models/mmvae_models.py:467-503 (the container the evaluation code
    reads; `batch[m]["data"]` None = missing, masks kept).  joint = product of the PRESENT shared experts (:477-479).
    Draw order: z_joint; then per modality z_shared -- a missing modality draws from the FIRST present modality's shared
    posterior (:489-490) --, z_private -- a missing one from N(0, I) (:491-493) --, and one fresh shared draw of every
    OTHER PRESENT modality for the cross reconstructions (:499-502).
    Returns per modality: q_shared, q_private (None if missing), z_shared, px, joint_px, cross {src: loc}; and joint."""
    M, D = len(mods), n_latents
    present = [i for i in range(M) if batch[f"mod_{i + 1}"]["data"] is not None]
    enc = {i: encode(p, mods, i, batch[f"mod_{i + 1}"], train) for i in present}
    B = enc[present[0]][0].shape[0]
    sh = {i: (enc[i][0][:, :D], enc[i][1][:, :D]) for i in present}
    pr = {i: (enc[i][0][:, D:], enc[i][1][:, D:]) for i in present}
    j_mu, j_var = product_of_experts(torch.stack([sh[i][0] for i in present]), torch.stack([sh[i][1] for i in present]))
    it = iter(eps)
    nxt = lambda d: next(it).reshape(1, B, d)
    z_joint = j_mu + j_var * nxt(D)
    out = []
    for i in range(M):
        mask = batch[f"mod_{i + 1}"]["masks"]
        P = int(mods[i]["private"])
        src = i if i in present else present[0]
        z_sh = sh[src][0] + sh[src][1] * nxt(D)
        z_pr = (pr[i][0] + pr[i][1] * nxt(P)) if i in present else nxt(P)
        dec = lambda z: decode(p, mods, i, torch.cat([z, z_pr], -1), mask, train)
        px, jpx = dec(z_sh), dec(z_joint)
        cross = {}
        for m in present:
            if m != i:
                cross[m] = dec(sh[m][0] + sh[m][1] * nxt(D))
        out.append({"q_shared": sh.get(i), "q_private": pr.get(i), "z_shared": z_sh, "px": px, "joint_px": jpx,
                    "cross": cross})
    assert next(it, None) is None, "unused noise draws"
    return out, (j_mu, j_var)


def vae_param_shapes(mod, n_latents):
    """trainable tensors of a unimodal VAE (models/trainer.py:112-113: `self.model = vaes["mod_1"]`): the tower keys
    without the `vaes.mod_1.` prefix (its `_pz_params` are not trainable, models/vae.py:159-162)"""
    full = tower_param_shapes("vaes.mod_1", mod["enc"], mod["dec"], mod["data_dim"], n_latents, mod.get("private"))
    return {k[len("vaes.mod_1."):]: v for k, v in full.items()}


def vae_objective(p, mods, batch, eps, n_latents, beta=1.0, train=False):
    """VAE.forward + objective, the unimodal case (models/vae.py:92-119,268-282) with UnimodalObjective.elbo
    (models/objectives.py:233-247): q = Normal(mu, lv as sigma), ONE rsample, recon = dec(z);
    lpx_z = -ReconLoss (B, F);  kld = KL(q || N(0, 1)) (B, D) against the RAW `_pz_params` (zeros, ones);
    loss = -(lpx_z.sum(-1) - beta * kld.sum()).sum()  -- kld.sum() is the batch TOTAL, subtracted from every row."""
    mod = mods[0]
    pp = {"vaes.mod_1." + k: v for k, v in p.items()}
    mu, sig = encode(pp, [mod], 0, batch["mod_1"], train)
    B = mu.shape[0]
    z = mu + sig * eps[0].reshape(1, B, n_latents)
    out = decode(pp, [mod], 0, z, batch["mod_1"]["masks"], train)
    lpx_z = -recon_loss(mod["ltype"], out, batch["mod_1"])
    kld = kl_normal(mu, sig, 0.0, torch.ones(1, n_latents))
    loss = -(lpx_z.sum(-1) - beta * kld.sum()).sum()
    return {"loss": loss, "kld": kld, "reconstruction_loss": lpx_z}


OBJECTIVES = {"vae": vae_objective, "mopoe": mopoe_objective, "poe": poe_objective, "moe": moe_objective, "dmvae": dmvae_objective}


# ----------------------------------------------------------------------------------------------
# optimiser
# ----------------------------------------------------------------------------------------------
def adabelief_step(params, grads, state, lr, step, b1=0.9, b2=0.999, eps=1e-16):
    """`optimizer: adabelief` (models/trainer.py:82-86): adabelief_pytorch.AdaBelief(lr, eps=1e-16, betas=(0.9, 0.999),
    weight_decouple=True, rectify=False), weight_decay 0 (the package's default: the decoupled decay is a no-op).  The
    package is NOT vendored by the reference and absent in this image (environment.yml lists it without a version):
    PARITY UNPINNED -- restated from Zhuang et al., "AdaBelief Optimizer", NeurIPS 2020, Algorithm 2 and the package's
    (0.2.x) update order, including its in-place `exp_avg_var.add_(eps)` (the eps of the paper's s_t update stays in the
    state).  `step` is 1-based.  state: dict name -> (exp_avg, exp_avg_var), updated in place; params updated in place."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k, g in grads.items():
        m, s_ = state[k]
        m.mul_(b1).add_(g, alpha=1 - b1)
        r = g - m
        s_.mul_(b2).addcmul_(r, r, value=1 - b2)
        denom = (s_.add_(eps).sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].data.addcdiv_(m, denom, value=-lr / bc1)


def adam_amsgrad_step(params, grads, state, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(amsgrad=True) single-tensor update (models/trainer.py:79-81); `step` is 1-based.
    state: dict name -> (m, v, vmax) updated in place; params updated in place."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k, g in grads.items():
        m, v, vmax = state[k]
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        torch.maximum(vmax, v, out=vmax)
        denom = (vmax.sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].data.addcdiv_(m, denom, value=-lr / bc1)
