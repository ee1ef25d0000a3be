import sys, torch
sys.path.insert(0, ".")
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload
dev = torch.device("cuda", 0)
desc, cfg, dims, data, meta = workload("cfg1", None, device=dev, seed=1)
tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
tr.model.train(); tr.configure_optimizers(); tr.capture(data, 1)
print(tr.model._job_calls, tr.model.decoder_calls_moved, tr.abi_calls_in_graph)
