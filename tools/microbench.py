"""Per-kernel timings on the GPU box (HIP events around hipGraph replays of 20 back-to-back launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_vae_comparison_amd import hipops as H
from multimodal_vae_comparison_amd import ops

dev = "cuda"


def timeit(fn, reps=20, n=20):
    """Average time of one call: n calls captured in a hipGraph (no host launch gaps), replayed reps times."""
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n


out = {}
batches = [int(b) for b in os.environ.get("MB_BATCHES", "128,512,2048,4096").split(",")]
with torch.no_grad():
    L = H.lib()
    for B in batches:
        for name, Cin, Hin in (("conv1", 3, 64), ("conv2", 32, 32), ("conv3", 32, 16), ("conv4", 32, 8)):
            x = torch.randn(B, Cin, Hin, Hin, device=dev)
            w = torch.randn(32, Cin, 4, 4, device=dev) * .05
            b = torch.zeros(32, device=dev)
            us = timeit(lambda: ops.conv2d_k4s2(x, w, b, H.ACT_SILU if Cin == 32 else 0))
            fl = 2.0 * B * (Hin // 2) ** 2 * 32 * Cin * 16
            out[f"gather {name} B={B}"] = (us, fl / us / 1e6)
        for name, Cout, Hin in (("convT_64", 32, 4), ("convT1", 32, 8), ("convT2", 32, 16), ("convT3", 3, 32)):
            x = torch.randn(B, 32, Hin, Hin, device=dev)
            w = torch.randn(32, Cout, 4, 4, device=dev) * .05
            b = torch.zeros(Cout, device=dev)
            us = timeit(lambda: ops.convT2d_k4s2(x, w, b, H.ACT_RELU, H.EP_SIGMOID_CLAMP if Cout == 3 else 0))
            fl = 2.0 * B * Hin ** 2 * 32 * Cout * 16
            out[f"scatter {name} B={B}"] = (us, fl / us / 1e6)
        for name, Q, Hs in (("conv2.wgrad", 32, 16), ("conv3.wgrad", 32, 8), ("conv4.wgrad", 32, 4), ("conv1.wgrad", 3, 32)):
            dy = torch.randn(B, 32, Hs, Hs, device=dev)
            x = torch.randn(B, Q, 2 * Hs, 2 * Hs, device=dev)
            dw = torch.zeros(32, Q, 4, 4, device=dev)
            db = torch.zeros(32, device=dev)
            ws = torch.empty(L.mmvae_conv_wgrad_ws_floats(B, 32, Q, Hs), device=dev)
            us = timeit(lambda: L.mmvae_conv2d_k4s2_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                           ws.data_ptr(), B, Q, 32, Hs, 1, 1, torch.cuda.current_stream().cuda_stream))
            fl = 2.0 * B * Hs * Hs * 32 * Q * 16
            out[f"{name} (+reduce) B={B}"] = (us, fl / us / 1e6)
    for M, K, N in ((128, 512, 512), (4096, 54, 162), (4096, 128, 54), (4096, 54, 128), (128, 32, 512), (131072, 54, 162)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev)
        b = torch.zeros(N, device=dev)
        us = timeit(lambda: ops.linear(x, w, b))
        out[f"linear fwd M={M} K={K} N={N}"] = (us, 2.0 * M * K * N / us / 1e6)
for k, v in out.items():
    print(f"{k:48s} {v[0]:10.2f} us" + (f"   {v[1]:8.2f} TFLOP/s" if v[1] else ""))
# input expansion on the device
for B in batches:
    u8 = torch.randint(0, 256, (B, 3, 64, 64), dtype=torch.uint8, device=dev)
    img = torch.empty(B, 3, 64, 64, device=dev)
    tok = torch.randint(0, 27, (B, 32), dtype=torch.int32, device=dev)
    ln = torch.full((B,), 32, dtype=torch.int32, device=dev)
    oh = torch.empty(B, 32, 27, device=dev)
    mk = torch.empty(B, 32, dtype=torch.uint8, device=dev)
    us = timeit(lambda: (ops.expand_image_u8(u8, img), ops.expand_text_tokens(tok, ln, oh, mk)))
    print(f"{'input expansion (u8 image + tokens) B=' + str(B):48s} {us:10.2f} us   {(5 * u8.numel() + 4 * oh.numel()) / us / 1e3:8.2f} GB/s")
