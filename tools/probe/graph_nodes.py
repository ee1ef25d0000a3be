"""How many nodes does the captured cfg2 step have?  (hipGraphDebugDotPrint through torch.cuda.CUDAGraph.debug_dump)"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MMVAE_GRAPH_DUMP"] = "/tmp/mmvae_step.dot"
import torch
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
desc, cfg, dims, data, meta = workload(name, None, device=dev, seed=1)
torch.manual_seed(0)
tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
tr.model.train(); tr.configure_optimizers()
tr.capture(data, 1)
txt = open("/tmp/mmvae_step.dot").read()
kinds = {}
for m in re.finditer(r'label="([^"]*)"', txt):
    k = m.group(1).split("\\n")[0][:40]
    kinds[k] = kinds.get(k, 0) + 1
nodes = len(re.findall(r'^\s*"?[\w]+"?\s*\[', txt, flags=re.M))
edges = txt.count("->")
print(name, "dot bytes", len(txt), "node statements", nodes, "edges", edges)
print(sorted(kinds.items(), key=lambda kv: -kv[1])[:12])
print(txt[:1500])
