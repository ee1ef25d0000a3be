"""conv2-shape gather conv at batch 128: every plan against plan 1 (bit-level agreement is not expected: the K-split
sums in a different order) and its graph-timed duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gtime import timeit
from multimodal_vae_comparison_amd import hipops as H
L = H.lib()
plan = int(os.environ.get("MMVAE_GATHER_PLAN", "-1"))
for B, Hin in ((128, 32), (128, 16), (97, 32), (512, 32), (2048, 32)):
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 32, Hin, Hin, generator=g).cuda(); w = (torch.randn(32, 32, 4, 4, generator=g) * .05).cuda()
    b = torch.randn(32, generator=g).cuda(); y = torch.empty(B, 32, Hin // 2, Hin // 2, device="cuda")
    f = lambda: L.mmvae_conv2d_k4s2_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, 32, 32, Hin, H.ACT_SILU, 0,
                                        torch.cuda.current_stream().cuda_stream)
    assert f() == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(torch.nn.functional.silu(x.double()), w.double(), b.double(), stride=2, padding=1)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"plan {plan} B={B} Hin={Hin}: rel err {err:.2e}  {timeit(f):7.2f} us")
