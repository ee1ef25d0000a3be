"""cycles per (MFMA + K independent v_fma_f32) for one wave per SIMD: additive (64 + 4K) or hidden (max)?"""
import ctypes, os, subprocess, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "mfma_valu.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma_valu.hip"), "-o", so])
L = ctypes.CDLL(so)
L.mfma_valu.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
out = torch.zeros(1 << 20, device="cuda"); tk = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
iters = 64
for bf16 in (0, 1):
    for blocks, threads in ((1, 64), (256, 256)):
        row = []
        for k in (0, 2, 4, 8, 12, 16, 24):
            tk.zero_()
            for _ in range(2):
                L.mfma_valu(out.data_ptr(), tk.data_ptr(), iters, k, bf16, blocks, threads, None)
            torch.cuda.synchronize()
            nw = blocks * threads // 64
            row.append(f"K={k}: {np.median(tk[:nw].cpu().numpy()) / (iters * 16):6.1f}")
        print(f"{'bf16 32x32x16' if bf16 else 'f32 32x32x2  '} blocks {blocks:3d} x {threads} threads: cycles per (MFMA + K v_fma_f32): " + "  ".join(row))
