import ctypes, os, numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "mfma_lds.so"))
L.mfma_lds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(1 << 20, device="cuda"); tk = torch.zeros(1 << 14, dtype=torch.int64, device="cuda")
iters = 64
for blocks in (256, 512):
    for mode, name in ((0, "operands in registers"), (1, "A, B from LDS (1 ds_read2_b32 per MFMA)"), (2, "two accumulators share B (0.75 per MFMA)")):
        for _ in range(2):
            L.mfma_lds(out.data_ptr(), tk.data_ptr(), iters, mode, blocks, None)
        torch.cuda.synchronize()
        t = tk[:blocks * 4].cpu().numpy()
        n = iters * 8 * (2 if mode == 2 else 1)
        print(f"{blocks} workgroups ({blocks // 256} wave(s) per SIMD), {name}: {np.median(t) / n:6.1f} cycles per MFMA per wave")
