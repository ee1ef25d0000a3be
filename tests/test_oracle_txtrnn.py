"""Enc_TxtRNN (reference models/encoders.py:840-869) has no running oracle: the reference crashes on its own batch format
(SURVEY 0.4), so the tower is a DEFINED path, parity unpinned against the reference.  What CAN be pinned is the
third-party arithmetic it is made of: the oracle's restatement of nn.Embedding + bidirectional nn.GRU + `output[-1]` +
direction sum + Linear + chunk + softmax is checked here against torch's own modules (CPU)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import golden_weights as gw
from oracle import mmvae_oracle as orc


class _TorchTxtRNN(nn.Module):
    """the reference's constructor lines (encoders.py:841-852) and forward lines (:856-869) with a (B,T) id batch laid
    out sequence-first for nn.GRU"""

    def __init__(self, feats, out_dim, hidden=512):
        super().__init__()
        self.hidden_size = hidden
        self.embed = nn.Embedding(feats, hidden)
        self.gru = nn.GRU(hidden, hidden, 1, dropout=0.0, bidirectional=True)
        self.o2p = nn.Linear(hidden, out_dim * 2)

    def forward(self, ids):                                  # ids (B,T)
        embedded = self.embed(ids.t())                       # (T,B,H)
        output, _ = self.gru(embedded, None)
        output = output[-1]
        output = output[:, :self.hidden_size] + output[:, self.hidden_size:]
        mu, logvar = torch.chunk(self.o2p(output), 2, dim=1)
        return mu, F.softmax(logvar, dim=-1) + 1e-6


@pytest.mark.parametrize("B,T,D", [(5, 6, 8), (3, 1, 4), (9, 12, 16)])
def test_oracle_txtrnn_is_torch_gru(B, T, D):
    V = 27
    shapes = orc.tower_param_shapes("vaes.mod_1", "TxtRNN", "TxtTransformer", [45, V, 1], D)
    shapes = {k: v for k, v in shapes.items() if ".enc." in k}
    p = gw.make_params(shapes, 3, requires_grad=True)
    g = torch.Generator().manual_seed(B + T)
    ids = torch.randint(0, V, (B, T), generator=g)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    mask = torch.arange(T)[None, :] < lens[:, None]
    onehot = F.one_hot(ids, V).float() * mask[..., None]
    ref = _TorchTxtRNN(V, D)
    ref.load_state_dict({k[len("vaes.mod_1.enc."):]: v.detach() for k, v in p.items()})
    ids_eff = onehot.argmax(-1)                               # padding rows are token 0
    mu_r, lv_r = ref(ids_eff)
    (mu_r.sum() + (lv_r * torch.arange(D)).sum()).backward()
    mu, lv = orc.enc_txt_rnn(p, "vaes.mod_1", onehot, mask)
    (mu.sum() + (lv * torch.arange(D)).sum()).backward()
    assert torch.allclose(mu, mu_r, rtol=1e-5, atol=1e-6) and torch.allclose(lv, lv_r, rtol=1e-5, atol=1e-7)
    for k, t in p.items():
        rg = dict(ref.named_parameters())[k[len("vaes.mod_1.enc."):]].grad
        a = t.grad if t.grad is not None else torch.zeros_like(t)
        b = rg if rg is not None else torch.zeros_like(t)
        err = float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))
        assert err <= 1e-4, (k, err)
