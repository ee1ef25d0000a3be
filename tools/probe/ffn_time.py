"""graph-timed fused feed-forward kernels at the action towers' shape (12 800 rows, FF 1024), with / without dropout,
fp32-MFMA kernels (csrc/ffn.hip) beside the split-bf16 ones (csrc/ffn_b16.inc)"""
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
wsplit = torch.empty(L.mmvae_ffn32_wsplit_bytes(FF), dtype=torch.uint8, device="cuda")
rsplit = torch.empty(L.mmvae_ffn32_rsplit_bytes(M), dtype=torch.uint8, device="cuda")
st = DropoutState().to("cuda"); slot, call = st.begin(); spec = st.spec(slot, call, 3, 0.1, "ffn")
s = lambda: torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
for name, d in (("dropout 0.1", spec.c()), ("no dropout", None)):
    f = lambda: L.mmvae_ffn32_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y), M, FF, d, s())
    bd = lambda: L.mmvae_ffn32_bwd(P(x), P(dy), P(w1), P(b1), P(w2), P(dx), None, M, FF, d, s())
    bw = lambda: L.mmvae_ffn32_bwd(P(x), P(dy), P(w1), P(b1), P(w2), None, P(ws), M, FF, d, s())
    pw = lambda: L.mmvae_ffn32_prep_weights(P(w1), P(w2), P(wsplit), FF, s())
    f16 = lambda: L.mmvae_ffn32_fwd_b16(P(x), P(wsplit), P(b1), P(b2), P(y), M, FF, d, s())
    bd16 = lambda: L.mmvae_ffn32_bwd_b16(P(x), P(dy), P(wsplit), P(b1), P(dx), None, P(rsplit), None, M, FF, d, s())
    bw16 = lambda: L.mmvae_ffn32_bwd_b16(P(x), P(dy), P(wsplit), P(b1), None, P(ws), P(rsplit), None, M, FF, d, s())
    pw()
    t = [timeit(k) for k in (f, bd, bw, pw, f16, bd16, bw16)]
    fl = 2.0 * M * 32 * FF * 2
    print(f"{name}: fp32 MFMA  fwd {t[0]:6.1f} us ({fl / t[0] / 1e6:5.1f} TFLOP/s)  data {t[1]:6.1f}  weights {t[2]:6.1f} ({parts} row slices)")
    print(f"{name}: split bf16 fwd {t[4]:6.1f} us ({fl / t[4] / 1e6:5.1f} TFLOP/s)  data {t[5]:6.1f}  weights (incl. row split) {t[6]:6.1f}  weight split {t[3]:5.1f}")
