"""Dec_CNN's last layer: the col2im kernel (csrc/conv_t3.inc) plain and fused with the bce loss, graph-timed alone.
python tools/probe/t3_time.py [B ...]"""
import sys
import torch
sys.path.insert(0, ".")
from multimodal_vae_comparison_amd import ops, hipops as H


def graph_time(fn, reps=100):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph(); torch.cuda.synchronize()
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
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 10)


for B in [int(a) for a in sys.argv[1:]] or [128, 512, 1000]:
    x = torch.randn(B, 32, 32, 32, device="cuda").requires_grad_(True)
    w = torch.randn(32, 3, 4, 4, device="cuda") * 0.1
    b = torch.zeros(3, device="cuda")
    tgt = torch.rand(B, 3, 64, 64, device="cuda")
    seed = torch.full((B,), 0.5, device="cuda")
    with torch.no_grad():
        t_plain = graph_time(lambda: ops.convT2d_k4s2(x, w, b, H.ACT_RELU, H.EP_SIGMOID_CLAMP))
        y = ops.convT2d_k4s2(x, w, b, H.ACT_RELU, H.EP_SIGMOID_CLAMP)
        yr = y.detach().requires_grad_(True)

    def bce():
        with ops.ConstSeed(seed, 0.5):
            return ops.bce_sigmoid_rowsum(yr, tgt)

    def fused():
        with ops.ConstSeed(seed, 0.5):
            return ops.convT3_bce(x, w, b, H.ACT_RELU, None, None, tgt)
    t_bce, t_fused = graph_time(bce), graph_time(fused)
    mb = B * (131072 + 2 * 49152) / 1e6
    print(f"B={B:5d} convT3 fwd {t_plain:7.2f} us | bce_rowsum seeded {t_bce:7.2f} us | fused {t_fused:7.2f} us "
          f"({mb / t_fused / 1e6 * 1e6:.0f} GB/s algorithmic)", flush=True)
