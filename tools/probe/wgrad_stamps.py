"""Phase timestamps (clock64) of the weight-gradient workgroups (conv2 shape: 32 -> 32, 16x16 output, B=128):
build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -I../../include [-DWGRAD_VARIANT=bits] wgrad_probe.hip
-o wgrad_probe.so  (bits: 1 no bias sums, 2 no scheduling barriers, 4 no MFMA; WGRAD_PROBE_SO selects the file)"""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, os.environ.get("WGRAD_PROBE_SO", "wgrad_probe.so")))
L.probe_wgrad.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
B = int(os.environ.get("PROBE_B", 128))
for Q, Hs in ((32, 16),):
    dy = torch.randn(B, 32, Hs, Hs, device="cuda"); x = torch.randn(B, Q, 2 * Hs, 2 * Hs, device="cuda")
    dw = torch.zeros(32, Q, 4, 4, device="cuda"); db = torch.zeros(32, device="cuda")
    ws = torch.zeros(64 << 20, device="cuda")
    st = torch.zeros(1024 * 4 * 48, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert L.probe_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), st.data_ptr(), B, Q, Hs, 1, s) == 0
    torch.cuda.synchronize()
    a = st.cpu().numpy().reshape(-1, 4, 48)
    nwg = int((a[:, 0, 0] != 0).sum())
    a = a[:nwg]
    t0 = a[:, :, 0].min()
    nst = int((a[0, 0] != 0).sum())
    rel = np.median(a[:, 0, :nst] - a[:, 0, :1], axis=0).astype(int)
    print(f"Q={Q} Hs={Hs}: {nwg} workgroups, {nst} stamps; median ticks since WG start (wave 0):")
    print("  start, then per macro tile [pre-barrier, barrier, stored, barrier2], ..., mfma-done, end")
    print(" ", rel.tolist())
    d = np.diff(rel)
    print("  deltas:", d.tolist())
    print("  WG start spread (min/median/max):", int(a[:, 0, 0].min() - t0), int(np.median(a[:, 0, 0] - t0)), int(a[:, 0, 0].max() - t0),
          " last end:", int(a[:, :, :nst].max() - t0))
