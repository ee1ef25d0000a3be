"""Graph-timed Linear launches of the cfg2 step's shapes (20 per graph): forward with / without an input activation,
data gradient alone, grouped data + weight gradient, the weight-gradient batch launch."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multimodal_vae_comparison_amd import hipops as H
from gather_b16 import timed  # noqa: E402

L = H.lib()
CORE = int(os.environ.get('LIN_CORE', '1'))
L.mmvae_gemm_b16_set(CORE)
for M in (128, 384, 512, 1000, 2048, 4096, 7680):
    g = torch.Generator().manual_seed(M)
    for (N, K) in ((512, 512), (400, 784)) if M > 128 else ((512, 512), (64, 512), (512, 32)):
        x = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) * 0.05).cuda()
        b = torch.randn(N, generator=g).cuda()
        y = torch.empty(M, N).cuda()
        dy = torch.randn(M, N, generator=g).cuda()
        dx = torch.empty(M, K).cuda()
        dw, db = torch.zeros(N, K).cuda(), torch.zeros(N).cuda()
        nws = max(L.mmvae_linear_bwd_ws_floats(M, N, K), L.mmvae_linear_bwd_weight_ws_floats(M, N, K), 4)
        ws = torch.empty(nws).cuda()
        st = lambda: torch.cuda.current_stream().cuda_stream
        res = {}
        for name, act in (("fwd", 0), ("fwd silu", 1), ("fwd relu", 2)):
            res[name] = timed(lambda: L.mmvae_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, K, act, 0, st()))
        res["bwd data"] = timed(lambda: L.mmvae_linear_bwd_data(dy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), M, N, K, 0, 0, st()))
        res["bwd data relu-mask"] = timed(lambda: L.mmvae_linear_bwd_data(dy.data_ptr(), w.data_ptr(), x.data_ptr(), dx.data_ptr(), M, N, K, H.EP_MUL_RELU_MASK, 0, st()))
        res["bwd grouped"] = timed(lambda: L.mmvae_linear_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), None, dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N, K, K, 0, 0, 1, st()))
        res["bwd weight"] = timed(lambda: L.mmvae_linear_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), M, N, K, K, 0, 1, st()))
        print(f"M={M} N={N} K={K} b16={CORE}: " + "  ".join(f"{k} {v:.1f}" for k, v in res.items()))
