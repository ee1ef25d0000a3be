import ctypes, os, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "mfma_rate.so"))
L.mfma_rate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(1 << 20, device="cuda"); tk = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
iters = 64
for blocks in (1, 256, 512):
    for threads in (256, 512):
        for nacc in (1, 2, 4):
            tk.zero_()
            for _ in range(2):
                L.mfma_rate(out.data_ptr(), tk.data_ptr(), iters, nacc, blocks, threads, None)
            torch.cuda.synchronize()
            nw = blocks * threads // 64
            t = tk[:2 * nw].cpu().numpy().reshape(-1, 2)
            n = iters * 16
            print(f"blocks {blocks:4d} threads {threads} ({threads // 256} wave(s)/SIMD) accs {nacc}: {np.median(t[:, 0]) / n:6.1f} clock64 ticks per MFMA per wave, "
                  f"{np.median(t[:, 1]) * 10 / n:6.2f} ns per MFMA per wave")
