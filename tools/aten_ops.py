"""Which ATen ops still launch kernels in one objective + backward (they should all be ours)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import cdsprites_batch, cdsprites_config
tr = MultimodalVAE(cdsprites_config("mopoe", 32), device="cuda"); tr.model.train(); tr.configure_optimizers()
batch = cdsprites_batch(128, 32, seed=1, device="cuda")
one = torch.ones((), device="cuda")
for _ in range(2):
    tr.model.objective(batch)["loss"].backward(one)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.model.objective(batch)["loss"].backward(one)
    torch.cuda.synchronize()
for e in prof.key_averages(group_by_stack_n=12):
    if e.key in ("aten::zeros", "aten::select_backward", "aten::copy_", "aten::select") and e.device_time_total > 0:
        print(e.key, e.count, f"{e.device_time_total:.1f}us")
        for s in e.stack[:12]:
            print("     ", s[-120:])
