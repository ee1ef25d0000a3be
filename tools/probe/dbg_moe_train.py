import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_e2e as t
from multimodal_vae_comparison_amd import ops, hipops
from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
from multimodal_vae_comparison_amd.models.mmvae_models import MOE
from multimodal_vae_comparison_amd.synthetic import cdsprites_batch
orc, gw = t.orc, t.gw
for bp, streams in ((False, False), (True, False), (False, True), (True, True)):
    MOE.batch_passes = bp
    ops.StreamPlan.enabled = streams
    B, T, D = 5, 6, 8
    torch.manual_seed(1234); DropoutState._next_seed[0] = 0x1234567
    params = gw.make_params(orc.model_param_shapes(t.MODS, D), 11, requires_grad=True)
    tr = t._build("moe", D, 1.0, {k: v.detach() for k, v in params.items()}, hipops.lib())
    tr.model.train()
    batch = cdsprites_batch(B, T, seed=3)
    g = torch.Generator().manual_seed(5)
    eps = [torch.randn(1, B, D, generator=g) for _ in range(2)]
    tr.model.eps_override = [e.clone() for e in eps]
    ops.DropSpec.recorder = []
    out = tr.model.objective(t._to_dev(batch))
    torch.cuda.synchronize()
    rec = ops.DropSpec.recorder; ops.DropSpec.recorder = None
    print(bp, streams, "loss", out["loss"].item(), [r.detach().cpu().numpy().round(3) for r in out["reconstruction_loss"]], len(rec), [r[0] for r in rec])
