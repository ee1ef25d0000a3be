"""split-bf16 weight-gradient kernel (csrc/conv_wgrad_b16.inc) against the fp32-MFMA one: summed partial rows vs an fp64
autograd weight gradient, bias sums, graph-timed launches."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gather_b16 import timed  # noqa: E402
here = os.path.dirname(os.path.abspath(__file__))
ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(here)), "multimodal_vae_comparison_amd", "libmmvae_hip.so"), mode=ctypes.RTLD_GLOBAL)
L = ctypes.CDLL(os.path.join(here, "libwgradb16.so"))
L.probe_wgrad.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
L.probe_wgrad_rows.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3
for Hs in (16, 8):
    for B in (6, 128, 1000):
        g = torch.Generator().manual_seed(B + Hs)
        small = torch.randn(B, 32, Hs, Hs, generator=g).cuda()
        large = torch.randn(B, 32, 2 * Hs, 2 * Hs, generator=g).cuda()
        rows, rowlen, bcol = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.probe_wgrad_rows(B, Hs, ctypes.byref(rows), ctypes.byref(rowlen), ctypes.byref(bcol))
        rows, rowlen, bcol = rows.value, rowlen.value, bcol.value
        if B <= 128:
            w = torch.zeros(32, 32, 4, 4, dtype=torch.float64, device="cuda", requires_grad=True)
            y = F.conv2d(torch.relu(large.double()), w, None, stride=2, padding=1)
            (ref,) = torch.autograd.grad(y, w, small.double())
            ref_a = small.double().sum((0, 2, 3))
            ref_b = torch.relu(large.double()).sum((0, 2, 3))
        for bias_from in (1, 2):
            res = {}
            for b16 in (0, 1, 128, 256):
                if b16 > 1 and bias_from == 2:
                    continue
                nrows = rows if b16 <= 1 else min(b16, rows * (B * Hs // (64 // Hs)) // max(rows, 1))
                nrows = rows if b16 <= 1 else min(b16, (B * Hs + (64 // Hs) - 1) // (64 // Hs))
                ws = torch.full((max(nrows, rows) * rowlen,), float("nan"), device="cuda")
                call = lambda: L.probe_wgrad(small.data_ptr(), large.data_ptr(), ws.data_ptr(), B, Hs, 0, 2, bias_from, b16,
                                             torch.cuda.current_stream().cuda_stream)
                rc = call()
                torch.cuda.synchronize()
                part = ws.view(-1, rowlen)[:nrows].double()
                dw = part[:, :bcol].sum(0).view(32, 32, 4, 4)
                db = part[:, bcol:bcol + 32].sum(0)
                res[b16] = dw
                msg = ""
                if B <= 128:
                    e = float((dw - ref).abs().max() / ref.abs().max())
                    rb = ref_a if bias_from == 1 else ref_b
                    eb = float((db - rb).abs().max() / rb.abs().max())
                    msg = f"dw err vs fp64 {e:.2e}  db err {eb:.2e}"
                t = timed(call) if bias_from == 1 else float("nan")
                print(f"Hs={Hs:2d} B={B:5d} bias_from={bias_from} {('split-bf16' if b16 == 1 else f'b16 x{b16:3d}  ') if b16 else 'fp32 MFMA '} rc={rc} {msg}   {t:7.1f} us")
            print("      b16 vs f32 max rel diff", float((res[1] - res[0]).abs().max() / res[0].abs().max()))
