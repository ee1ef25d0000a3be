"""graph-timed fused feed-forward kernels at the action towers' shape (12 800 rows, FF 1024), with / without dropout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch
from gtime import timeit
from multimodal_vae_comparison_amd import hipops as H, ops
from multimodal_vae_comparison_amd.models.nn_modules import DropoutState
L = H.lib()
M, FF = int(os.environ.get("M", 12800)), int(os.environ.get("FF", 1024))
x = torch.randn(M, 32, device="cuda"); dy = torch.randn(M, 32, device="cuda")
w1 = torch.randn(FF, 32, device="cuda") * .3; b1 = torch.randn(FF, device="cuda") * .3
w2 = torch.randn(32, FF, device="cuda") * .1; b2 = torch.randn(32, device="cuda")
y = torch.empty_like(x); dx = torch.empty_like(x)
parts, rowlen = L.mmvae_ffn32_bwd_parts(M, FF), L.mmvae_ffn32_bwd_rowlen(FF)
ws = torch.empty(parts * rowlen, device="cuda")
st = DropoutState().to("cuda"); slot, call = st.begin(); spec = st.spec(slot, call, 3, 0.1, "ffn")
s = lambda: torch.cuda.current_stream().cuda_stream
for name, d in (("dropout 0.1", spec.c()), ("no dropout", None)):
    f = lambda: L.mmvae_ffn32_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), M, FF, d, s())
    bd = lambda: L.mmvae_ffn32_bwd(x.data_ptr(), dy.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), dx.data_ptr(), ws.data_ptr(), M, FF, d, s())
    bw = lambda: L.mmvae_ffn32_bwd(x.data_ptr(), dy.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), None, ws.data_ptr(), M, FF, d, s())
    tf, tb, tw = timeit(f), timeit(bd), timeit(bw)
    fl = 2.0 * M * 32 * FF * 2
    print(f"{name}: fwd {tf:6.1f} us ({fl / tf / 1e6:5.1f} TFLOP/s)  bwd data+weights {tb:6.1f} us  weights only {tw:6.1f} us ({parts} row slices)")
