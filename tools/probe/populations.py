"""Which workgroups of the gather conv are the slow ones?  (probe bit 32 stamps + placement in slot 15)"""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
B, Hin, plan = 128, 32, 1
x = torch.randn(B, 32, Hin, Hin, device="cuda"); w = torch.randn(32, 32, 4, 4, device="cuda") * .05
b = torch.zeros(32, device="cuda"); y = torch.empty(B, 32, Hin // 2, Hin // 2, device="cuda")
L = ctypes.CDLL(os.path.join(here, "probe_32_0.so"))
L.probe_gather_stamps.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
st = torch.zeros(4096 * 4 * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    L.probe_gather_stamps(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), B, Hin, plan, 1,
                          torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 4, 16)
nwg = int((a[:, 0, 0] != 0).sum()); a = a[:nwg]
nst = int((a[0, 0, :15] != 0).sum())
tot = a[:, 0, nst - 1] - a[:, 0, 0]
lds = (a[:, 0, 15] >> 32) & 0xFF
hwid = a[:, 0, 15] & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
print("total cycles by LDS base (first / second resident):")
for v in np.unique(lds):
    m = lds == v
    print(f"  lds_base {v:3d}: n={m.sum():4d} median total {int(np.median(tot[m]))}")
print("by blockIdx quartile:", [int(np.median(tot[i * nwg // 4:(i + 1) * nwg // 4])) for i in range(4)])
rel = a[:, 0, :nst] - a[:, 0, :1]
slow = tot > np.median(tot)
print("median phases fast:", np.median(rel[~slow], axis=0).astype(int).tolist())
print("median phases slow:", np.median(rel[slow], axis=0).astype(int).tolist())
print("slow fraction by se:", {int(s): round(float(slow[se == s].mean()), 2) for s in np.unique(se)})
print("slow fraction by cu:", {int(s): round(float(slow[cu == s].mean()), 2) for s in np.unique(cu)})
