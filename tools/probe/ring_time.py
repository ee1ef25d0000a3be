"""in-graph input ring (MultimodalVAE.capture(..., input_ring=...)): step time of the bare replay and with the ring"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload
dev = torch.device("cuda", 0)
desc, cfg, dims, data, meta = workload("cfg2", None, device=dev, seed=1)
B, T, V = 128, 32, 27
g = torch.Generator().manual_seed(3)
def host_batches(tr):
    out = []
    for _ in range(4):
        u8 = torch.randint(0, 256, (B, 3, 64, 64), generator=g, dtype=torch.uint8)
        tok = torch.randint(0, V, (B, T), generator=g, dtype=torch.int32)
        lens = torch.randint(3, T + 1, (B,), generator=g, dtype=torch.int32)
        out.append(tr.pack_compact_pinned({"mod_1": {"u8": u8}, "mod_2": {"tokens": tok, "lengths": lens}}))
    return out
for ring in (False, True):
    torch.manual_seed(0)
    tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
    tr.model.train(); tr.configure_optimizers()
    tr.capture(data, 1, input_ring=host_batches(tr) if ring else None)
    for _ in range(30): tr.fused_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): tr.fused_step()
    torch.cuda.synchronize()
    print("ring" if ring else "bare", f"{(time.perf_counter() - t0) / 300 * 1e3:.4f} ms")
