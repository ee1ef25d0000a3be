"""Does running the two co-resident workgroups of a CU out of phase shorten the gather conv?  (probe bit 64:
the second resident of a CU -- non-zero LDS base -- sleeps DELAY x 4096 cycles before it starts)"""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from gtime import timeit
B, Hin, plan = 128, 32, 1
x = torch.randn(B, 32, Hin, Hin, device="cuda"); w = torch.randn(32, 32, 4, 4, device="cuda") * .05
b = torch.zeros(32, device="cuda"); y = torch.empty(B, 32, Hin // 2, Hin // 2, device="cuda")
for name in ("0_0", "64_1", "64_2", "64_3"):
    L = ctypes.CDLL(os.path.join(here, f"probe_{name}.so"))
    L.probe_gather.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    us = timeit(lambda: L.probe_gather(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, Hin, plan, 1,
                                       torch.cuda.current_stream().cuda_stream))
    print(f"probe {name:6s}: {us:7.2f} us")
for name in ("32_0", "96_2"):
    L = ctypes.CDLL(os.path.join(here, f"probe_{name}.so"))
    L.probe_gather_stamps.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    st = torch.zeros(4096 * 4 * 16, dtype=torch.int64, device="cuda")
    for _ in range(3):
        L.probe_gather_stamps(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), B, Hin, plan, 1,
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    a = st.cpu().numpy().reshape(-1, 4, 16)
    nwg = int((a[:, 0, 0] != 0).sum()); a = a[:nwg]
    nst = int((a[0, 0] != 0).sum())
    rel = a[:, 0, :nst] - a[:, 0, :1]
    tot = rel[:, -1]
    print(f"stamps {name}: {nwg} WGs; median phases:", np.median(rel, axis=0).astype(int).tolist())
    print(f"   per-WG total cycles: min {tot.min()} median {int(np.median(tot))} max {tot.max()};  two populations:",
          int(np.median(np.sort(tot)[: nwg // 2])), int(np.median(np.sort(tot)[nwg // 2:])))
