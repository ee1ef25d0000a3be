import ctypes, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools/probe")
import torch
from multimodal_vae_comparison_amd import hipops as H
from gather_b16 import timed
L = H.lib()
for (M, N) in ((12800, 96), (12800, 32), (3200, 96)):
    x = torch.randn(M, 32).cuda(); w = torch.randn(N, 32).cuda(); b = torch.randn(N).cuda(); y = torch.empty(M, N).cuda()
    st = lambda: torch.cuda.current_stream().cuda_stream
    t = timed(lambda: L.mmvae_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, 32, 32, 0, 0, st()))
    ref = x.double() @ w.double().t() + b.double()
    print(M, N, f"{t:.2f} us", float((y.double() - ref).abs().max() / ref.abs().max()))
