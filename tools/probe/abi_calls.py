import sys, collections
sys.path.insert(0, "/root/repo")
import torch
from multimodal_vae_comparison_amd import ops
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload
names = []
_c = ops._call
def call(name, *a):
    names.append(name)
    return _c(name, *a)
dev = torch.device("cuda", 0)
desc, cfg, dims, data, meta = workload(sys.argv[1] if len(sys.argv) > 1 else "cfg2", int(sys.argv[2]) if len(sys.argv) > 2 else None, device=dev, seed=1)
tr = MultimodalVAE(cfg, feature_dims=dims, device=dev); tr.model.train(); tr.configure_optimizers()
tr.capture(data, 1)
ops._call = call
import multimodal_vae_comparison_amd.ops as O
names.clear()
tr.capture(data, 1)
print(len(names)); print(collections.Counter(names).most_common())
print(names)
