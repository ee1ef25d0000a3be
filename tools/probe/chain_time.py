"""The fused linear chain (csrc/chain.hip) against one launch per layer, graph-timed on the two chains of the cfg2 step:
Dec_CNN lin1 -> lin2 -> lin3 (M, 32 -> 512 -> 512 -> 512) and Enc_CNN2 lin1 -> heads (M, 512 -> 512 -> 64), forward and
forward + backward.  python tools/probe/chain_time.py [M ...]"""
import math
import sys

import torch

sys.path.insert(0, ".")
from multimodal_vae_comparison_amd import ops  # noqa: E402

DEV = "cuda"


def graph_time(fn, reps=200):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 10)


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [32, 128, 256]
    for M in Ms:
        for name, widths, acts in (("dec lin1-3", [32, 512, 512, 512], [0, 2, 2]), ("enc lin1+heads", [512, 512, 64], [1, 0])):
            g = torch.Generator().manual_seed(1)
            x = torch.randn(M, widths[0], generator=g).to(DEV).requires_grad_(True)
            lay = []
            for i, a in enumerate(acts):
                w = (torch.randn(widths[i + 1], widths[i], generator=g) / math.sqrt(widths[i])).to(DEV)
                b = torch.zeros(widths[i + 1], device=DEV)
                lay.append((w, b, a, torch.zeros_like(w), torch.zeros_like(b)))
            dy = torch.randn(M, widths[-1], generator=g).to(DEV)
            res = {}
            for fused in (False, True):
                ops.LINEAR_CHAIN = fused

                def fwd():
                    with torch.no_grad():
                        return ops.linear_chain(x, lay, ("probe", name))

                def fwdbwd():
                    y = ops.linear_chain(x, lay, ("probe", name))
                    y.backward(dy)
                    x.grad = None
                res[fused] = (graph_time(fwd), graph_time(fwdbwd))
            print(f"M={M:4d} {name:15s} fwd: per-layer {res[False][0]:6.2f} us  chain {res[True][0]:6.2f} us | "
                  f"fwd+bwd: per-layer {res[False][1]:6.2f} us  chain {res[True][1]:6.2f} us", flush=True)
    print("timeouts:", ops.chain_timeouts(torch.device("cuda", 0)))


if __name__ == "__main__":
    main()
