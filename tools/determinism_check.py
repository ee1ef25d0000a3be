"""Two identical objective + backward passes must give bit-identical gradients (no races in the HIP path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import cdsprites_batch, cdsprites_config

torch.manual_seed(0)
B, T, D = 128, 32, 32
tr = MultimodalVAE(cdsprites_config("mopoe", D, batch_size=B), device="cuda")
tr.model.train(os.environ.get("TRAIN", "0") == "1")
batch = cdsprites_batch(B, T, seed=3, device="cuda")
g = torch.Generator().manual_seed(5)
eps = [torch.randn(1, B, D, generator=g) for _ in range(2)]
ref = None
for it in range(int(os.environ.get("ITERS", "8"))):
    tr.flat.zero_grad()
    tr.model.eps_override = [e.clone() for e in eps]
    for vae in tr.model.vaes.values():          # same dropout masks every pass
        for part in (vae.enc, vae.dec):
            st = getattr(part, "drop_state", None)
            if st is not None:
                st.state[1:].zero_()
    out = tr.model.objective(batch)
    out["loss"].backward()
    torch.cuda.synchronize()
    gcur = tr.flat.grad.clone()
    if ref is None:
        ref = gcur
    else:
        diff = (gcur != ref)
        if diff.any():
            names = []
            for k, p in tr.model.named_parameters():
                if p.grad is not None and (p.grad != ref[(p.grad.data_ptr() - tr.flat.grad.data_ptr()) // 4:][:p.numel()].view_as(p.grad)).any():
                    names.append(k)
            print(f"pass {it}: {int(diff.sum())} elements differ, max abs {float((gcur - ref).abs().max()):.3e} in {names[:6]}")
        else:
            print(f"pass {it}: identical, loss {out['loss'].item():.4f}")
import hashlib
print("grad sha1", hashlib.sha1(ref.cpu().numpy().tobytes()).hexdigest()[:16], "loss", out["loss"].item())
