"""graph-timed attention forward / backward at the action towers' shape (Ta = 100, B = 128, 2 heads of 16)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch
from gtime import timeit
from multimodal_vae_comparison_amd import hipops as H
Lb = H.lib()
L, N, E, NH = int(os.environ.get("L", 100)), 128, 32, 2
hd = E // NH
qkv = torch.randn(L, N, 3 * E, device="cuda"); dqkv = torch.empty_like(qkv)
mask = torch.ones(N, L, dtype=torch.uint8, device="cuda")
do = torch.randn(L, N, E, device="cuda"); out = torch.empty(L, N, E, device="cuda")
probs = torch.empty(N, NH, L, L, device="cuda")
state = torch.zeros(2 + H.DROPOUT_SLOTS, dtype=torch.int32, device="cuda"); state[0] = 12345
p, d = qkv.data_ptr(), dqkv.data_ptr()
s = lambda: torch.cuda.current_stream().cuda_stream
for name, dr in (("dropout 0.1", ctypes.byref(H.Dropout(state.data_ptr(), 0, 1, 0.1))), ("no dropout", None)):
    fwd = lambda: Lb.mmvae_attn_fwd(p, p + 4 * E, p + 8 * E, mask.data_ptr(), out.data_ptr(), probs.data_ptr(), L, L, N, NH, hd,
                                    3 * E, 3 * E, 3 * E, 1, dr, s())
    bwd = lambda: Lb.mmvae_attn_bwd(p, p + 4 * E, p + 8 * E, probs.data_ptr(), do.data_ptr(), d, d + 4 * E, d + 8 * E, L, L, N,
                                    NH, hd, 3 * E, 3 * E, 3 * E, dr, s())
    assert fwd() == 0 and bwd() == 0
    print(f"{name}: fwd {timeit(fwd):6.1f} us   bwd {timeit(bwd):6.1f} us")
