"""split-bf16 scatter convolution (csrc/conv_scatter_b16.inc: ConvTranspose2d 32 -> 32 forward, 16 x 16 -> 32 x 32) against the
fp32-MFMA kernel: error vs fp64, graph-timed."""
import ctypes, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gather_b16 import timed  # noqa: E402
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libscatterb16.so"))
L.probe_scatter.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
SHAPES = [(32, 32, 16), (32, 32, 8), (64, 32, 8), (64, 64, 8)]       # (reduced channels, output channels, input grid)
for (C, O, H), B in [(s, B) for s in SHAPES for B in (5, 128, 512, 1000, 3840)]:
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, C, H, H, generator=g).cuda()
    w = (torch.randn(C, O, 4, 4, generator=g) * 0.05).cuda()
    b = torch.randn(O, generator=g).cuda()
    ref = F.conv_transpose2d(torch.relu(x.double()), w.double(), b.double(), stride=2, padding=1) if B <= 512 else None
    out = {}
    for name, b16 in (("fp32 MFMA", 0), ("split-bf16", 1)):
        y = torch.full((B, O, 2 * H, 2 * H), float("nan"), device="cuda")
        call = lambda: L.probe_scatter(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, C, O, H, 2, 0, b16,
                                       torch.cuda.current_stream().cuda_stream)
        rc = call()
        torch.cuda.synchronize()
        err = float((y.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
        out[b16] = y
        print(f"C={C} O={O} H={H} B={B:5d} {name:11s} rc={rc} max rel err vs fp64 {err:.2e}   {timed(call):7.1f} us")
    print("      b16 vs f32 max abs diff", float((out[1] - out[0]).abs().max()))
