"""What does the optimiser launch cost inside a graph?  Chains of 20 dependent launches captured in one hipGraph:
a 1-element fill (the per-node floor), mmvae_adam_amsgrad_flat at the cfg2 parameter count and at half / double of it
(intercept + slope), a plain 5-array read + 4-array write of the same bytes done with torch ops for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import hipops as H, ops

dev = torch.device("cuda", 0)
N0 = 986890


def timed(fn, reps=20, iters=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        for _ in range(5):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


one = torch.zeros(4, device=dev)
print(f"1-element fill           {timed(lambda: H.lib().mmvae_fill(one.data_ptr(), 1, 0.0, H.stream())):6.2f} us / launch")
for n in (N0 // 2, N0, 2 * N0, 4 * N0):
    p, g, m, v, x = (torch.randn(n, device=dev) for _ in range(5))
    v.abs_(); x.abs_()
    sd = torch.zeros(8, dtype=torch.int32, device=dev)
    t = timed(lambda: ops.adam_amsgrad_flat(p, g, m, v, x, 1e-3, 0.9, 0.999, 1e-8, -1, sd, 1.0, True))
    print(f"adam n = {n:8d}        {t:6.2f} us / launch   ({n * 40 / t / 1e6:5.2f} TB/s of 40 B/param)")
