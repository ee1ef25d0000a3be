"""Phase stamps (s_memtime, 100 MHz) of workgroup 0 / wave 0 of the tiled split-bf16 GEMM: make -C csrc probe_g16, then
MMVAE_HIP_LIB=tools/probe/libmmvae_g16probe.so python tools/probe/g16_stamps.py [M N K]"""
import ctypes, os, sys
here = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MMVAE_HIP_LIB", os.path.join(here, "libmmvae_g16probe.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch
from multimodal_vae_comparison_amd import hipops as H

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1000, 512, 512)
L = H.lib()
x = torch.randn(M, K).cuda(); w = torch.randn(N, K).cuda(); b = torch.randn(N).cuda(); y = torch.empty(M, N).cuda()
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.mmvae_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), M, N, K, K, 0, 0, st)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
P = ctypes.CDLL(os.environ["MMVAE_HIP_LIB"])
P.mmvae_g16_probe(buf)
v = list(buf)
names = {0: "start", 1: "first two chunks' loads issued", 60: "loop done", 61: "outputs stored"}
t0 = v[0]
prev = t0
for i in sorted(k for k in range(64) if v[k]):
    lab = names.get(i) or ["stored", "barrier", "loads issued + 24 MFMAs", "barrier"][(i - 2) % 4] + f" (chunk {(i - 2) // 4})"
    print(f"{(v[i] - t0) * 10:8d} ns  (+{(v[i] - prev) * 10:6d})  {lab}")
    prev = v[i]
