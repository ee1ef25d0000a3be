"""Linear backward of the action towers' projection shapes (12 800 rows, d_model 32): grouped launch vs data gradient +
tall-skinny weight-gradient kernel (csrc/twgrad.hip), graph-timed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import hipops as H
from gather_b16 import timed  # noqa: E402
L = H.lib()
for M in (12800, 3200):
    for (N, K) in ((96, 32), (32, 32), (32, 6), (6, 32)):
        g = torch.Generator().manual_seed(M)
        x = torch.randn(M, K, generator=g).cuda(); w = torch.randn(N, K, generator=g).cuda()
        dy = torch.randn(M, N, generator=g).cuda(); dx = torch.empty(M, K).cuda()
        dw, db = torch.zeros(N, K).cuda(), torch.zeros(N).cuda()
        nws = max(L.mmvae_linear_bwd_ws_floats(M, N, K), L.mmvae_linear_bwd_weight_ws_floats(M, N, K), L.mmvae_txt_wgrad_ws_floats(M, N, K), 4)
        ws = torch.empty(nws).cuda()
        st = lambda: torch.cuda.current_stream().cuda_stream
        res = {}
        res["grouped"] = timed(lambda: L.mmvae_linear_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), None, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N, K, K, 0, 0, 2, st()))
        res["data"] = timed(lambda: L.mmvae_linear_bwd_data(dy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), M, N, K, 0, 0, st()))
        if L.mmvae_txt_wgrad_supported(M, N, K):
            job = (H.TxtWgradJob * 1)()
            job[0].dy, job[0].x, job[0].ws, job[0].M, job[0].N, job[0].K = dy.data_ptr(), x.data_ptr(), ws.data_ptr(), M, N, K
            res["txt_wgrad x1"] = timed(lambda: L.mmvae_txt_wgrad(ctypes.cast(job, ctypes.c_void_p), 1, st()))
            jobs = (H.TxtWgradJob * 8)()
            wss = [torch.empty(nws).cuda() for _ in range(8)]
            for i in range(8):
                jobs[i].dy, jobs[i].x, jobs[i].ws, jobs[i].M, jobs[i].N, jobs[i].K = dy.data_ptr(), x.data_ptr(), wss[i].data_ptr(), M, N, K
            res["txt_wgrad x8"] = timed(lambda: L.mmvae_txt_wgrad(ctypes.cast(jobs, ctypes.c_void_p), 8, st()))
        print(f"M={M} N={N} K={K}: " + "  ".join(f"{k} {v:.1f}" for k, v in res.items()))
