"""Batch-128 Linear layers of the image tower: forward and fused backward (data + weight gradient) timings,
graph-timed like tools/microbench.py.  Run once with MMVAE_RGEMM=0 and once with =1 to compare the staged and
the register-operand GEMM bodies."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_vae_comparison_amd import hipops as H

dev = "cuda"


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


L = H.lib()
print("MMVAE_RGEMM =", os.environ.get("MMVAE_RGEMM", "1"))
for M, K, N in ((128, 512, 512), (128, 512, 64), (128, 32, 512), (128, 512, 1024), (64, 512, 512), (256, 512, 512)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    ws = torch.empty(max(1, L.mmvae_linear_bwd_ws_floats(M, N, K)), device=dev)
    s = torch.cuda.current_stream
    f = timeit(lambda: L.mmvae_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, K,
                                          H.ACT_SILU, H.EP_NONE, s().cuda_stream))
    bw = timeit(lambda: L.mmvae_linear_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), x.data_ptr(), dx.data_ptr(),
                                           dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N, K, K, H.ACT_SILU,
                                           H.EP_MUL_SILU_GRAD, 0, s().cuda_stream))
    print(f"M={M:4d} K={K:4d} N={N:4d}   fwd {f:7.2f} us   bwd(data+weight) {bw:7.2f} us")

# operand-orientation probe through the plain GEMM entry point: same 128x512x512 problem, B read k-major (Linear
# forward: W[n][k]) or n-major (a transposed copy W^T[k][n])
M, K, N = 128, 512, 512
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
y = torch.empty(M, N, device=dev)
s = torch.cuda.current_stream
for name, sbk, sbn in (("B k-major", 1, K), ("B n-major", N, 1)):
    t = timeit(lambda: L.mmvae_gemm_f32(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), None, None, M, N, K, K, 1,
                                        sbk, sbn, N, 0, 0, 0, 0, 1, s().cuda_stream))
    print(f"gemm 128x512x512 {name}: {t:7.2f} us")
for name, sam, sak in (("A m-major", 1, M),):
    t = timeit(lambda: L.mmvae_gemm_f32(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), None, None, M, N, K, sam,
                                        sak, N, 1, N, 0, 0, 0, 0, 1, s().cuda_stream))
    print(f"gemm 128x512x512 {name}, B n-major: {t:7.2f} us")
