"""Phase timestamps (clock64) of individual workgroups of the gather conv: where one workgroup's time goes."""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "probe_32.so"))
L.probe_gather_stamps.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
B, Hin = 128, 32
for plan in (1, 0):
    x = torch.randn(B, 32, Hin, Hin, device="cuda"); w = torch.randn(32, 32, 4, 4, device="cuda") * .05
    b = torch.zeros(32, device="cuda"); y = torch.empty(B, 32, Hin // 2, Hin // 2, device="cuda")
    st = torch.zeros(4096 * 4 * 16, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.probe_gather_stamps(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), st.data_ptr(), B, Hin, plan, 1, s)
    torch.cuda.synchronize()
    a = st.cpu().numpy().reshape(-1, 4, 16)
    nwg = int((a[:, 0, 0] != 0).sum())
    a = a[:nwg]
    t0 = a[:, :, 0].min()
    names = ["start", "slots done", "ch0 pre-store", "ch0 stored", "ch0 barrier", "ch1 pre-store", "ch1 stored", "ch1 barrier", "c2a", "c2b", "c2c", "end"]
    print(f"plan {plan}: {nwg} workgroups; clock64 ticks (100 MHz = 10 ns each if s_memrealtime, else shader clocks)")
    nst = int((a[0, 0] != 0).sum())
    rel = a[:, 0, :nst] - a[:, 0, :1]
    print("  median per-WG phase ticks since WG start:", np.median(rel, axis=0).astype(int).tolist())
    print("  WG start spread:", int(a[:, 0, 0].min() - t0), int(np.median(a[:, 0, 0] - t0)), int(a[:, 0, 0].max() - t0), " last end:", int(a[:, 0, nst - 1].max() - t0))
