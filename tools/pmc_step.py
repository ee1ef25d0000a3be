"""A few EAGER (not graph-replayed) training steps of a bench workload, for rocprofv3 --pmc / --kernel-trace: the same
kernels the captured step replays, one dispatch record each.  PMC_CFG (cfg2), PMC_B (the workload's batch), PMC_STEPS (3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload

name = os.environ.get("PMC_CFG", "cfg2")
B = int(os.environ.get("PMC_B", "0")) or None
dev = torch.device("cuda", 0)
desc, cfg, dims, data, meta = workload(name, B, device=dev, seed=1)
torch.manual_seed(0)
tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
tr.model.train()
tr.configure_optimizers()
tr._one = torch.ones((), device=dev)
for _ in range(int(os.environ.get("PMC_STEPS", "3"))):
    tr._fwd_bwd(data)
    tr._finish_step()
    tr.optimizer.step()
torch.cuda.synchronize()
