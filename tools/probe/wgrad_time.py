"""Stand-alone time of the text towers' (L*N)-row weight gradients (mmvae_linear_bwd_weight, deferred partials):
one launch per gradient vs mmvae_linear_bwd_weight_batch, at the cfg2 batch (4096 rows) and at B=1000 (32000 rows)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import hipops as H


def timeit(fn, reps=20, n=20):      # n dependent launches in one captured graph
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
dev = "cuda"
for rows in (4096, 32000):
    for name, D, FF in (("dec d=32", 32, 128), ("enc d=54", 54, 128)):
        shapes = [(rows, 3 * D, D), (rows, D, D), (rows, FF, D), (rows, D, FF)]
        bufs = []
        for M, N, K in shapes:
            nz = L.mmvae_linear_bwd_weight_splits(M, N, K)
            bufs.append((torch.randn(M, N, device=dev), torch.randn(M, K, device=dev), torch.zeros(N, K, device=dev),
                         torch.zeros(N, device=dev), torch.zeros(max(1, L.mmvae_linear_bwd_weight_ws_floats(M, N, K)), device=dev), nz))
        s = torch.cuda.current_stream
        each = []
        for (M, N, K), (dy, x, dw, db, ws, nz) in zip(shapes, bufs):
            t = timeit(lambda: L.mmvae_linear_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                                         M, N, K, K, H.ACT_NONE, H.ACC_DEFER, s().cuda_stream))
            each.append(t)
            print(f"rows {rows:6d} {name}  dW ({N:3d},{K:3d})  splits {nz:3d}  {t:7.2f} us")
        arr = (H.WgradJob * len(shapes))()
        for j, (M, N, K), (dy, x, dw, db, ws, nz) in zip(arr, shapes, bufs):
            j.dy, j.x, j.dw, j.db, j.ws = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr()
            j.M, j.N, j.K, j.ldx, j.x_act, j.accumulate = M, N, K, K, H.ACT_NONE, H.ACC_DEFER
        t = timeit(lambda: L.mmvae_linear_bwd_weight_batch(ctypes.cast(arr, ctypes.c_void_p), len(shapes), s().cuda_stream))
        print(f"rows {rows:6d} {name}  all four: separate {sum(each):7.2f} us, one batch {t:7.2f} us")
