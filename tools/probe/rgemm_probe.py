import torch, sys
sys.path.insert(0, "/root/repo")
from multimodal_vae_comparison_amd import hipops as H
L = H.lib()
dev = "cuda"
for (M, K, N) in ((6, 8, 128), (6, 8, 512), (128, 512, 512), (5, 16, 64)):
    for acc in (0, 1, 2):
        torch.manual_seed(0)
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); dy = torch.randn(M, N, device=dev)
        dx = torch.empty(M, K, device=dev); dw0 = torch.randn(N, K, device=dev); db0 = torch.randn(N, device=dev)
        dw, db = dw0.clone(), db0.clone()
        ws = torch.empty(max(1, L.mmvae_linear_bwd_ws_floats(M, N, K)), device=dev)
        rc = L.mmvae_linear_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), None, dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                ws.data_ptr(), M, N, K, K, 0, 0, acc, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        rw = dy.double().T @ x.double() + (dw0.double() if acc else 0)
        rb = dy.double().sum(0) + (db0.double() if acc else 0)
        rx = dy.double() @ w.double()
        e = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        print(M, K, N, "acc", acc, "rc", rc, "dw", e(dw, rw), "db", e(db, rb), "dx", e(dx, rx), "splits", L.mmvae_linear_bwd_splits(M, N, K))
