"""The K-preserving text decoder (`oracle.dec_txt_transformer(keep_k=True)`: BASELINE configs[2] as stated, K = 8) is a
DEFINED EXTENSION -- the reference's Dec_TxtTransformer (models/decoders.py:708-723) attends over the K latent samples
and cannot run any K > 1 objective -- so it cannot be pinned against the reference.  What can be pinned is
that the restatement equals torch's OWN decoder modules applied to every latent sample separately: for each k,
torch.nn.TransformerDecoder(TransformerDecoderLayer(d, 2, 128, gelu)) over the length-1 memory z[k], then the final
Linear, in float64 -- output, gradient with respect to z and every parameter gradient (the same kind
of independent pin the ResNet restatement and the GRU restatement have)."""
import math

import pytest
import torch
import torch.nn as nn

from oracle import golden_weights as gw
from oracle import mmvae_oracle as orc


class _TorchTxtDecoder(nn.Module):
    """the reference's constructor lines (decoders.py:679-697) with torch's modules; forward = its lines :708-723 for ONE
    latent sample (memory length 1)"""

    def __init__(self, D, V):
        super().__init__()
        layer = nn.TransformerDecoderLayer(d_model=D, nhead=2, dim_feedforward=128, dropout=0.1, activation="gelu")
        self.seqTransDecoder = nn.TransformerDecoder(layer, num_layers=1)
        self.finallayer = nn.Linear(D, V)
        self.D = D

    def forward(self, z1, mask):                              # z1 (B,D), mask (B,T) bool
        B, T = mask.shape
        pos = torch.arange(T, dtype=z1.dtype).unsqueeze(1)
        div = torch.exp(torch.arange(0, self.D, 2, dtype=z1.dtype) * (-math.log(10000.0) / self.D))
        pe = torch.zeros(T, 1, self.D, dtype=z1.dtype)
        pe[:, 0, 0::2] = torch.sin(pos * div)
        pe[:, 0, 1::2] = torch.cos(pos * div)[:, : self.D // 2]
        tq = torch.zeros(T, B, self.D, dtype=z1.dtype) + pe
        out = self.seqTransDecoder(tgt=tq, memory=z1.unsqueeze(0), tgt_key_padding_mask=~mask)
        out = self.finallayer(out)
        return out.permute(1, 0, 2) * mask.unsqueeze(-1).to(z1.dtype)


@pytest.mark.parametrize("K,B,T,D", [(3, 4, 6, 8), (8, 2, 5, 16), (1, 3, 7, 8)])
def test_keep_k_decoder_is_torch_decoder_per_sample(K, B, T, D):
    V = 27
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        shapes = orc.tower_param_shapes("vaes.mod_2", "TxtTransformer", "TxtTransformer", [T, V, 1], D)
        shapes = {k: v for k, v in shapes.items() if ".dec." in k}
        p = {k: v.double().detach().requires_grad_(True) for k, v in gw.make_params(shapes, 5).items()}
        g = torch.Generator().manual_seed(K * 100 + B * 10 + T)
        z = torch.randn(K, B, D, generator=g, dtype=torch.float64).requires_grad_(True)
        lens = torch.randint(1, T + 1, (B,), generator=g)
        lens[0] = T
        mask = torch.arange(T)[None, :] < lens[:, None]
        wgt = torch.randn(K * B, T, V, generator=g, dtype=torch.float64)

        out = orc.dec_txt_transformer(p, "vaes.mod_2", z, mask, data_dim=(T, V, 1), train=False, keep_k=True)
        assert tuple(out.shape) == (K * B, T, V)
        (out * wgt).sum().backward()

        ref = _TorchTxtDecoder(D, V).double().eval()
        strip = "vaes.mod_2.dec."
        sd = {}
        for k, v in p.items():
            name = k[len(strip):].replace("finallayer.module.", "finallayer.")
            sd[name] = v.detach()
        missing, unexpected = ref.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all(".pe" in m or m.endswith("pe") for m in missing), missing
        z_ref = z.detach().clone().requires_grad_(True)
        outs = [ref(z_ref[k], mask) for k in range(K)]        # every latent sample on its own
        out_ref = torch.cat(outs, 0)                          # row k * B + b
        (out_ref * wgt).sum().backward()

        assert float((out - out_ref).detach().abs().max()) <= 1e-9 * max(1.0, float(out_ref.detach().abs().max()))
        assert float((z.grad - z_ref.grad).abs().max()) <= 1e-9 * max(1.0, float(z_ref.grad.abs().max()))
        rp = dict(ref.named_parameters())
        for k, v in p.items():
            name = k[len(strip):].replace("finallayer.module.", "finallayer.")
            a = v.grad if v.grad is not None else torch.zeros_like(v)
            b = rp[name].grad if rp[name].grad is not None else torch.zeros_like(v)
            assert float((a - b).abs().max()) <= 1e-9 * max(1.0, float(b.abs().max())), k
    finally:
        torch.set_default_dtype(prev)
