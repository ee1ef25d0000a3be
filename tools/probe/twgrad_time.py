"""Stand-alone time of a fused text layer's weight gradients: one launch per gradient (mmvae_linear_bwd_weight) vs
mmvae_txt_wgrad (csrc/twgrad.hip), at the cfg2 batch (4096 rows), B=512 (16384) and B=1000 (32000 rows)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import hipops as H
from tools.gtime import timeit

L = H.lib()
dev = "cuda"
s = torch.cuda.current_stream
for rows in (4096, 16384, 32000):
    for name, D, FF, dec in (("dec d=32", 32, 128, True), ("enc d=54", 54, 128, False)):
        shapes = [(rows, 3 * D, D), (rows, D, D), (rows, FF, D), (rows, D, FF)]
        if dec:
            shapes += [(rows, D, D), (rows // 32, D, D)]
        prob = [(torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)) for M, N, K in shapes]
        each = []
        for (M, N, K), (dy, x) in zip(shapes, prob):
            dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
            ws = torch.zeros(max(1, L.mmvae_linear_bwd_weight_ws_floats(M, N, K)), device=dev)
            each.append(timeit(lambda: L.mmvae_linear_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                                                  ws.data_ptr(), M, N, K, K, H.ACT_NONE, H.ACC_DEFER,
                                                                  s().cuda_stream)))
        arr = (H.TxtWgradJob * len(shapes))()
        keep = []
        flop = 0
        byts = 0
        for j, (M, N, K), (dy, x) in zip(arr, shapes, prob):
            ws = torch.zeros(L.mmvae_txt_wgrad_ws_floats(M, N, K), device=dev)
            keep.append(ws)
            j.dy, j.x, j.ws, j.M, j.N, j.K = dy.data_ptr(), x.data_ptr(), ws.data_ptr(), M, N, K
            flop += 2 * M * N * K
            byts += 4 * M * (N + K) + 4 * ws.numel()
        t = timeit(lambda: L.mmvae_txt_wgrad(ctypes.cast(arr, ctypes.c_void_p), len(shapes), s().cuda_stream))
        print(f"rows {rows:6d} {name}: separate {sum(each):7.2f} us ({' '.join(f'{e:.1f}' for e in each)}), "
              f"txt_wgrad {t:7.2f} us = {flop / t * 1e-6:.1f} TFLOP/s, {byts / t * 1e-6:.2f} TB/s algorithmic")
