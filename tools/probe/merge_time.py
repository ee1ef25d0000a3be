"""step time with / without merged fold segments (ops.GradReducer.merge_adjacent), one process, interleaved"""
import sys, time
import torch
sys.path.insert(0, ".")
from multimodal_vae_comparison_amd import ops
from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
from multimodal_vae_comparison_amd.synthetic import workload
dev = torch.device("cuda", 0)
for name in sys.argv[1:] or ["cfg1", "cfg5", "cfg2"]:
    trs = {}
    for flag in (False, True):
        ops.GradReducer.merge_adjacent = flag
        desc, cfg, dims, data, meta = workload(name, None, device=dev, seed=1)
        torch.manual_seed(0)
        tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
        tr.model.train(); tr.configure_optimizers(); tr.capture(data, 1)
        trs[flag] = tr
        for _ in range(60): tr.fused_step()
    for rep in range(3):
        for flag in (False, True):
            tr = trs[flag]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(300): tr.fused_step()
            torch.cuda.synchronize()
            print(name, "merged" if flag else "plain ", f"{(time.perf_counter() - t0) / 300 * 1e3:.4f} ms", tr.abi_calls_in_graph)
