"""s_memtime stamps inside the pipelined split-bf16 gather kernel (B16_PROBE=32 build): median over workgroups of the
cycles between prologue / per-chunk MFMA loop / barrier"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gather_b16 import sig  # noqa: E402
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libgatherb16_p32.so")); L.probe_conv_b16.argtypes = sig
for B, tm in ((1000, 14), (1000, 24), (128, 14), (128, 24)):
    x = torch.randn(B, 32, 32, 32).cuda(); w = torch.randn(32, 32, 4, 4).cuda() * 0.05; b = torch.randn(32).cuda()
    y = torch.empty(B, 32, 16, 16, device="cuda")
    nwg = B * 16 // (4 if tm % 10 == 2 else 8)
    st = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device="cuda")
    for _ in range(2):
        L.probe_conv_b16(x.data_ptr(), w.data_ptr(), b.data_ptr(), st.data_ptr(), y.data_ptr(), B, 32, 32, 32, 2, 0, tm, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(nwg, 4, 16)[:, 0, :]
    n = int((s[0] != 0).sum())
    d = np.diff(s[:, :n], axis=1)
    # s_memtime ticks at 100 MHz; shader clock ~2.4 GHz
    print(f"B={B} tm={tm}: {n} stamps; median ticks between stamps: {np.median(d, axis=0).astype(int).tolist()}  total {int(np.median(s[:, n - 1] - s[:, 0]))} ticks (x24 = shader cycles)")
