"""Bisect: exp_avg after ONE cfg1 step, several execution modes against the one-stream captured step."""
import os
import sys
import torch
sys.path.insert(0, ".")
from multimodal_vae_comparison_amd import ops
from multimodal_vae_comparison_amd.models.mmvae_models import POE
from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload
DEV = torch.device("cuda", 0)


def run(streams, balance, captured, tweak=None):
    POE.balance_decoder_calls = balance
    ops.StreamPlan.enabled = streams
    torch.manual_seed(0)
    DropoutState._next_seed[0] = 0x1234567
    desc, cfg, dims, data, meta = workload("cfg1", 32, device=DEV, seed=1)
    cfg = dict(cfg, lr=1e-3)
    tr = MultimodalVAE(cfg, feature_dims=dims, device=DEV)
    tr.model.train()
    tr.configure_optimizers()
    if tweak:
        tweak(tr)
    if captured:
        tr.capture(data)
    else:
        tr._one = torch.ones((), device=DEV)
        ops.LincombRows.unit_seed_ptr = tr._one.data_ptr()
        for _ in range(2):
            tr._fwd_bwd(data)
            tr._finish_step()
        tr.flat.zero_grad()
    tr.model._rng_state[1:].zero_()
    for m in tr.model.modules():
        if isinstance(m, DropoutState):
            m.state[1:].zero_()
    torch.cuda.synchronize()
    if captured:
        tr.fused_step()
    else:
        tr._fwd_bwd(data)
        tr.optimizer.step()
        tr._finish_step()
    torch.cuda.synchronize()
    return tr.optimizer.m.clone()


ref = run(False, False, True)
for name, args in (("two streams, unbalanced, captured", (True, False, True)),
                   ("two streams, unbalanced, eager", (True, False, False)),
                   ("one stream, eager", (False, False, False)),
                   ("two streams, balanced, captured", (True, True, True)),
                   ("two streams, balanced, eager", (True, True, False))):
    m = run(*args)
    bad = (m != ref).nonzero().flatten()
    print(f"{name:40s} differing {bad.numel():6d}  first {bad[:3].tolist()}  max abs {float((m - ref).abs().max()):.3e}")
out = os.environ.get("DUMP")
if out:
    torch.save({"ref": ref.cpu(), "unb": run(True, False, True).cpu()}, out)
