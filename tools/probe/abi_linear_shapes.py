"""The Linear / weight-gradient C-ABI calls of one captured cfg2 step with their integer arguments (M, N, K, ...):

    python tools/probe/abi_linear_shapes.py 1000
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import ops
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload
rows = []
_c = ops._call
def call(name, *a):
    if "linear" in name or "gemm" in name or "wgrad" in name:
        rows.append((name, [x for x in a if isinstance(x, int) and abs(x) < 10**7]))
    return _c(name, *a)
dev = torch.device("cuda", 0)
desc, cfg, dims, data, meta = workload("cfg2", int(sys.argv[1]), device=dev, seed=1)
tr = MultimodalVAE(cfg, feature_dims=dims, device=dev); tr.model.train(); tr.configure_optimizers()
tr.capture(data, 1)
ops._call = call
tr.capture(data, 1)
for r in rows: print(r)
