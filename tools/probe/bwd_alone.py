"""Stand-alone times of a conv layer's backward pieces at the bench batch: data gradient, weight gradient (deferred
partials) and the fused launch of both -- is the fused launch the max or the sum of its parts when nothing else runs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import hipops as H

L = H.lib()
dev = "cuda"
B = int(os.environ.get("PROBE_B", 128))


def timeit(fn, reps=20, n=20):
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


s = lambda: torch.cuda.current_stream().cuda_stream
for Hout in (16, 8):
    dy = torch.randn(B, 32, Hout, Hout, device=dev)
    x = torch.randn(B, 32, 2 * Hout, 2 * Hout, device=dev)
    w = torch.randn(32, 32, 4, 4, device=dev) * .05
    dx = torch.empty_like(x)
    dw = torch.zeros(32, 32, 4, 4, device=dev)
    db = torch.zeros(32, device=dev)
    ws = torch.empty(L.mmvae_conv_wgrad_ws_floats(B, 32, 32, Hout), device=dev)
    p = [t.data_ptr() for t in (dy, x, w, dx, dw, db, ws)]
    td = timeit(lambda: L.mmvae_conv2d_k4s2_dgrad(p[0], p[2], p[1], p[3], B, 32, 32, Hout, H.EP_MUL_SILU_GRAD, s()))
    tw = timeit(lambda: L.mmvae_conv2d_k4s2_wgrad(p[0], p[1], p[4], p[5], p[6], B, 32, 32, Hout, H.ACT_SILU, H.ACC_DEFER, s()))
    tf = timeit(lambda: L.mmvae_conv2d_k4s2_bwd(p[0], p[1], p[2], p[3], p[4], p[5], p[6], B, 32, 32, Hout, H.ACT_SILU, H.ACC_DEFER, s()))
    print(f"conv2d 32->32 out {Hout:2d}x{Hout:<2d} B={B}: dgrad {td:6.2f}  wgrad {tw:6.2f}  fused {tf:6.2f} us")
    # ConvTranspose2d 32->32, input Hout x Hout (its dgrad is the gather conv over dy 2Hout x 2Hout)
    xt = torch.randn(B, 32, Hout, Hout, device=dev)
    dyt = torch.randn(B, 32, 2 * Hout, 2 * Hout, device=dev)
    dxt = torch.empty_like(xt)
    q = [t.data_ptr() for t in (dyt, xt, w, dxt, dw, db, ws)]
    td = timeit(lambda: L.mmvae_convT2d_k4s2_dgrad(q[0], q[2], q[1], q[3], B, 32, 32, Hout, H.EP_MUL_RELU_MASK, s()))
    tw = timeit(lambda: L.mmvae_convT2d_k4s2_wgrad(q[1], q[0], q[4], q[5], q[6], B, 32, 32, Hout, H.ACT_RELU, H.ACC_DEFER, s()))
    tf = timeit(lambda: L.mmvae_convT2d_k4s2_bwd(q[0], q[1], q[2], q[3], q[4], q[5], q[6], B, 32, 32, Hout, H.ACT_RELU, H.ACC_DEFER, s()))
    print(f"convT  32->32 in  {Hout:2d}x{Hout:<2d} B={B}: dgrad {td:6.2f}  wgrad {tw:6.2f}  fused {tf:6.2f} us")
