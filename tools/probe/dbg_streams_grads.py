"""Gradients of one objective + backward with the two-stream plan on and off (same parameters, same recorded noise), after
smaller cases have populated the caching allocator: a difference is a missing cross-stream dependency or a tensor that one
stream still reads when the allocator hands its memory out again.  (Round 4: this sequence exposed the input batch being
freed under the side stream's last kernel -- TorchMMVAE._fork now registers the batch with the side streams; the same
sequence is tests/test_parity_e2e.py::test_two_stream_step_with_a_temporary_batch.)

    python tools/probe/dbg_streams_grads.py [small-first|alone]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_e2e as t  # noqa: E402
from multimodal_vae_comparison_amd import hipops, ops  # noqa: E402
from multimodal_vae_comparison_amd.synthetic import cdsprites_batch  # noqa: E402

orc, gw = t.orc, t.gw


def run(B, T, D, K, streams):
    ops.StreamPlan.enabled = streams
    params = gw.make_params(orc.model_param_shapes(t.MODS, D), 13, requires_grad=False)
    batch = cdsprites_batch(B, T, seed=7)
    g = torch.Generator().manual_seed(8)
    eps = [torch.randn(K, B, D, generator=g) for _ in range(2)]
    tr = t._build("moe", D, 1.0, params, hipops.lib(), mods=t.MODS, obj="iwae", K=K)
    tr.model.eps_override = [e.clone() for e in eps]
    out = tr.model.objective(t._to_dev(batch))        # a TEMPORARY device batch: freed as soon as autograd lets go of it
    out["loss"].backward()
    torch.cuda.synchronize()
    return t._grads(tr)


if __name__ == "__main__":
    order = sys.argv[1] if len(sys.argv) > 1 else "small-first"
    if order == "small-first":
        run(6, 5, 8, 3, True)
        run(6, 6, 8, 2, True)
    a = run(256, 32, 32, 8, True)
    b = run(256, 32, 32, 8, False)
    worst = 0.0
    for k in a:
        d = float((a[k] - b[k]).abs().max() / max(float(b[k].abs().max()), 1e-20))
        worst = max(worst, d)
        if d > 1e-5:
            print(f"{d:9.2e} {k}")
    print("--- done", order, "worst relative difference", worst)
