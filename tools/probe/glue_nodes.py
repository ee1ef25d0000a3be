"""Which Python lines of the product path launch torch's OWN kernels (copies, fills, adds) inside a training step?

Every such launch is a graph node of ~4-5 us on the chain that is not one of the path's kernels.  Runs one eager step of a
workload under torch.profiler with Python stacks and prints, per aten op that launched a kernel, the innermost frames inside
the package.

    python tools/probe/glue_nodes.py [--config cdsprites_shipped] [--batch 0]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cdsprites_shipped")
    ap.add_argument("--batch", type=int, default=0)
    a = ap.parse_args()
    from multimodal_vae_comparison_amd.models.trainer import MultimodalVAE
    from multimodal_vae_comparison_amd.synthetic import workload
    dev = torch.device("cuda", 0)
    desc, cfg, dims, data, meta = workload(a.config, a.batch or None, device=dev, seed=1)
    torch.manual_seed(0)
    tr = MultimodalVAE(cfg, feature_dims=dims, device=dev)
    tr.model.train()
    tr.configure_optimizers()
    tr.capture(data, 1)               # allocates the static batch / warm buffers exactly as the bench does
    batch = tr._static_batch
    for _ in range(2):
        tr._fwd_bwd(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        tr._fwd_bwd(batch)
        torch.cuda.synchronize()
    glue = collections.Counter()
    kernels = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::"):
            continue
        ks = [k.name for k in ev.kernels] if ev.kernels else []
        if not ks:
            continue
        # the launching op itself: skip parents whose children launched the kernel (they repeat the kernel list)
        if any(c.kernels for c in ev.cpu_children):
            continue
        frames = [f for f in (ev.stack or []) if "multimodal_vae_comparison_amd" in f or "multimodal-vae" in f]
        where = " <- ".join(f.split("multimodal_vae_comparison_amd/")[-1] for f in frames[:3])
        if not where:      # no Python frames (autograd thread): the chain of parent ops / autograd nodes instead
            chain, p = [], ev.cpu_parent
            while p is not None and len(chain) < 4:
                chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd of "))
                p = p.cpu_parent
            where = " <- ".join(chain) or "(top level)"
        where += "  " + str([tuple(s) for s in (ev.input_shapes or []) if s][:3])
        glue[(ev.name, ks[0][:60], where)] += 1
        kernels[ks[0][:60]] += 1
    print(desc)
    for (op, k, where), n in sorted(glue.items(), key=lambda kv: -kv[1]):
        print(f"{n:3d}  {op:28s} {k:60s} {where}")
    print("total torch-launched kernels per step:", sum(glue.values()))


if __name__ == "__main__":
    main()
