import ctypes, os, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
Lb = ctypes.CDLL(os.path.join(here, "attn_probe.so"))
Lb.attn_probe.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 5 + [ctypes.c_long, ctypes.c_void_p]
L, N, E, NH = 100, 128, 32, 2
qkv = torch.randn(L, N, 3 * E, device="cuda"); mask = torch.ones(N, L, dtype=torch.uint8, device="cuda")
out = torch.empty(L, N, E, device="cuda"); probs = torch.empty(N, NH, L, L, device="cuda")
st = torch.zeros(N * NH * 4 * 8, dtype=torch.int64, device="cuda")
p = qkv.data_ptr()
for _ in range(3):
    assert Lb.attn_probe(st.data_ptr(), p, p + 4 * E, p + 8 * E, mask.data_ptr(), out.data_ptr(), probs.data_ptr(), L, L, N, NH, E // NH, 3 * E, None) == 0
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 4, 8)
rel = a[:, 0, :7] - a[:, 0, :1]
print("wave 0 median ticks since start: start, staged, scores, softmax, PV, out stored, probs stored")
print(np.median(rel, axis=0).astype(int).tolist())
print("WG start spread:", int(a[:, 0, 0].max() - a[:, 0, 0].min()), " last end - first start:", int(a[:, :3, 6].max() - a[:, 0, 0].min()))
