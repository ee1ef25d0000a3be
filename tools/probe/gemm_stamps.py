"""Phase timestamps (clock64) of the register-operand GEMM workgroups (library built with -DGEMM_PROBE)."""
import ctypes, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "gemm_probe.so"))
c_p, c_i, c_l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
L.mmvae_gemm_f32.argtypes = [c_p] * 7 + [c_i] * 3 + [c_l] * 5 + [c_i] * 5 + [c_p]
M, K, N = 128, 512, 512
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(M, N, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, sbk, sbn in (("fwd (A k-major, B k-major)", 1, K), ("dgrad form (B n-major)", N, 1)):
    st = torch.zeros(64 * 8 * 8, dtype=torch.int64, device="cuda")
    for _ in range(5):
        L.mmvae_gemm_f32(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), None, st.data_ptr(), M, N, K, K, 1, sbk, sbn,
                         N, 0, 0, 0, 0, 1, s)
    torch.cuda.synchronize()
    a = st.cpu().numpy().reshape(64, 8, 8)
    t0 = a[:, :, 0].min()
    rel = a[:, :, :5] - a[:, :, :1]
    print(name)
    print("  phases: start, loads issued, loads landed, mfma done, end")
    print("  median over waves :", np.median(rel.reshape(-1, 5), axis=0).astype(int).tolist())
    print("  max over waves    :", rel.reshape(-1, 5).max(axis=0).astype(int).tolist())
    print("  WG start spread: min/med/max", int(np.median(a[:, 0, 0] - t0)), int((a[:, 0, 0] - t0).max()), " last end:", int(a[:, :, 4].max() - t0))
