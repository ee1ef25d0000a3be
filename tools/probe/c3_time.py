"""The image tower's 3-channel layers alone (graph-timed): conv1 forward, conv1 weight gradient, convT3 backward.
MMVAE_HIP_LIB=<lib> python tools/probe/c3_time.py [B ...]"""
import sys
import torch
sys.path.insert(0, ".")
from multimodal_vae_comparison_amd import ops, hipops as H


def graph_time(fn, reps=50):
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


L = H.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
for B in [int(a) for a in sys.argv[1:]] or [128, 1000]:
    img = torch.rand(B, 3, 64, 64, device="cuda")
    y = torch.empty(B, 32, 32, 32, device="cuda")
    dy = torch.randn(B, 32, 32, 32, device="cuda")
    x32 = torch.randn(B, 32, 32, 32, device="cuda")
    dl = torch.randn(B, 3, 64, 64, device="cuda")
    dx = torch.empty_like(x32)
    w = torch.randn(32, 3, 4, 4, device="cuda") * 0.1
    b = torch.zeros(32, device="cuda")
    dw, db, db3 = torch.zeros_like(w), torch.zeros(32, device="cuda"), torch.zeros(3, device="cuda")
    nws = L.mmvae_conv_wgrad_ws_floats(B, 32, 3, 32)
    ws = torch.empty(nws, device="cuda")
    t_f = graph_time(lambda: L.mmvae_conv2d_k4s2_fwd(img.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, 3, 32, 64, 0, 0, st()))
    t_w = graph_time(lambda: L.mmvae_conv2d_k4s2_wgrad(dy.data_ptr(), img.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), B, 3, 32, 32, 0, H.ACC_DEFER, st()))
    t_b = graph_time(lambda: L.mmvae_convT2d_k4s2_bwd(dl.data_ptr(), x32.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), db3.data_ptr(), ws.data_ptr(), B, 32, 3, 32, H.ACT_RELU, H.ACC_DEFER, st()))
    print(f"B={B:5d} conv1 fwd {t_f:7.2f} us | conv1 wgrad {t_w:7.2f} us | convT3 bwd (dgrad + wgrad) {t_b:7.2f} us", flush=True)
