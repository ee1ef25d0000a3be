"""Graph-timed probe of the gather convolution variants (tools/probe/probe_<bits>.so)."""
import ctypes, glob, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
dev = "cuda"
def gtime(fn, n=20, reps=20):
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn(s.cuda_stream); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn(s.cuda_stream)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n
names = {0: "full", 1: "-mfma", 2: "-act", 8: "-weights", 11: "staging+epilogue only"}
for B, Hin, plans in ((128, 32, (0, 1, 3, 2)), (128, 16, (0, 1, 3, 2)), (128, 8, (0, 3, 2)), (2048, 32, (0, 1, 3)), (2048, 16, (0, 1, 3))):
    x = torch.randn(B, 32, Hin, Hin, device=dev); w = torch.randn(32, 32, 4, 4, device=dev) * .05
    b = torch.zeros(32, device=dev); y = torch.empty(B, 32, Hin // 2, Hin // 2, device=dev)
    for plan in plans:
        for bits in sorted(names):
            L = ctypes.CDLL(os.path.join(here, f"probe_{bits}.so"))
            L.probe_gather.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
            us = gtime(lambda st: L.probe_gather(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, Hin, plan, 1, st))
            print(f"B={B} Hin={Hin} plan={plan} {names[bits]:46s} {us:8.2f} us")
