"""debug: back-to-back chain launches against the per-layer result; where do mismatches sit?"""
import math, sys
import torch
sys.path.insert(0, ".")
from multimodal_vae_comparison_amd import ops

DEV = "cuda"
M, widths, acts = 128, [32, 512, 512, 512], [0, 2, 2]
g = torch.Generator().manual_seed(5)
lay = []
for i, a in enumerate(acts):
    w = (torch.randn(widths[i + 1], widths[i], generator=g) / math.sqrt(widths[i])).to(DEV)
    b = (torch.randn(widths[i + 1], generator=g) * 0.1).to(DEV)
    lay.append((w, b, a, None, None))
xs = [torch.randn(M, widths[0], generator=g).to(DEV) for _ in range(40)]
ops.LINEAR_CHAIN = False
want = [ops.linear_chain(x, lay, "dbg") for x in xs]
ops.LINEAR_CHAIN = True
big = torch.randn(64 << 20, device=DEV)
side = torch.cuda.Stream()
dev = torch.device("cuda", 0)
for load in (False, True):
    with torch.no_grad():
        for rep in range(6):
            if load:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        big.mul_(1.0001)
            got = [ops.linear_chain(x, lay, "dbg") for x in xs]
            torch.cuda.synchronize()
            bad = []
            for i, (a, b) in enumerate(zip(got, want)):
                d = (a - b).abs() > 1e-5 * b.abs().max()
                if bool(d.any()):
                    rows = d.any(1).nonzero().flatten().tolist()
                    cols = d.any(0).nonzero().flatten().tolist()
                    bad.append((i, len(rows), rows[:4], rows[-1], len(cols), cols[:4], cols[-1], float((a - b).abs().max())))
            print("load", load, "rep", rep, "bad launches:", len(bad), bad[:3], "timeouts", ops.chain_timeouts(dev),
                  "sync", [v[0].tolist() for k, v in ops._CHAIN_SYNC.items()], flush=True)
