import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_vae_comparison_amd import hipops as H
dev = "cuda"; L = H.lib()
def timeit(fn, reps=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
B = 128
for name, Hout in (("conv2", 16), ("conv3", 8), ("conv4", 4)):
    dy = torch.randn(B, 32, Hout, Hout, device=dev); x = torch.randn(B, 32, 2 * Hout, 2 * Hout, device=dev)
    w = torch.randn(32, 32, 4, 4, device=dev) * .05
    dx = torch.empty_like(x); dw = torch.zeros_like(w); db = torch.zeros(32, device=dev)
    ws = torch.empty(L.mmvae_conv_wgrad_ws_floats(B, 32, 32, Hout), device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    t_f = timeit(lambda: L.mmvae_conv2d_k4s2_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, 32, 32, Hout, 1, 2, st()))
    t_d = timeit(lambda: L.mmvae_conv2d_k4s2_dgrad(dy.data_ptr(), w.data_ptr(), x.data_ptr(), dx.data_ptr(), B, 32, 32, Hout, 3, st()))
    t_w = timeit(lambda: L.mmvae_conv2d_k4s2_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, 32, 32, Hout, 1, 2, st()))
    print(f"{name} bwd B={B}: fused {t_f:.1f} us | dgrad {t_d:.1f} + wgrad(no reduce) {t_w:.1f} = {t_d + t_w:.1f} us")
for name, Cout, Hin in (("convT2", 32, 16), ("convT1", 32, 8), ("convT3", 3, 32)):
    x = torch.randn(B, 32, Hin, Hin, device=dev); dy = torch.randn(B, Cout, 2 * Hin, 2 * Hin, device=dev)
    w = torch.randn(32, Cout, 4, 4, device=dev) * .05
    dx = torch.empty_like(x); dw = torch.zeros_like(w); db = torch.zeros(Cout, device=dev)
    ws = torch.empty(L.mmvae_conv_wgrad_ws_floats(B, 32, Cout, Hin), device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    t_f = timeit(lambda: L.mmvae_convT2d_k4s2_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, 32, Cout, Hin, 2, 2, st()))
    t_d = timeit(lambda: L.mmvae_convT2d_k4s2_dgrad(dy.data_ptr(), w.data_ptr(), x.data_ptr(), dx.data_ptr(), B, 32, Cout, Hin, 2, st()))
    t_w = timeit(lambda: L.mmvae_convT2d_k4s2_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, 32, Cout, Hin, 2, 2, st()))
    print(f"{name} bwd B={B}: fused {t_f:.1f} us | dgrad {t_d:.1f} + wgrad(no reduce) {t_w:.1f} = {t_d + t_w:.1f} us")
