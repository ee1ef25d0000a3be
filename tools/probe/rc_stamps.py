"""Where a data-gradient workgroup of the fused engine spends its time: wall-clock stamps (RC_PROBE build of
csrc/rconv.hip, tools/probe/libmmvae_rcprobe.so) at fixed points of rc_dgrad_body, one shape per run.
usage: python tools/probe/rc_stamps.py <H> <Cin> <Cout> <k> <stride> [B]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multimodal_vae_comparison_amd import hipops as H  # noqa: E402
H.LIB_PATH = os.path.join(ROOT, "tools", "probe", "libmmvae_rcprobe.so")
from multimodal_vae_comparison_amd import rconv  # noqa: E402
from multimodal_vae_comparison_amd.models.resnet import ConvW, BatchNorm2d  # noqa: E402

Hh, Cin, Cout, k, st = (int(v) for v in sys.argv[1:6])
B = int(sys.argv[6]) if len(sys.argv) > 6 else 24
DEV = torch.device("cuda:0")
lib = H.lib()
lib.mmvae_rc_probe.restype = ctypes.c_int
lib.mmvae_rc_probe.argtypes = [ctypes.c_void_p]
conv = ConvW(Cin, Cout, k, st, k // 2, channels_last=True).to(DEV)
bn, bnp = BatchNorm2d(Cout).to(DEV), BatchNorm2d(Cin).to(DEV)
u, up = rconv.Unit(conv, bn), rconv.Unit(ConvW(64, Cin, 1, 1, 0).to(DEV), bnp)
Ho = (Hh - 1) // st + 1
Min, M = B * Hh * Hh, B * Ho * Ho
gm = (Hh, Hh, k, st, k // 2) if (k > 1 or st > 1) else rconv.IDENT
x = torch.randn(Min, Cin, device=DEV)
bp = up.buffers(Min, DEV)
bp["mean"].zero_(); bp["sc"].fill_(1.0); bp["rstd"].fill_(1.0)
y, b = rconv._fwd(u, x, Min, M, rconv.PRE_BN_RELU, (bp, bnp.bias), gm, False)
G = torch.randn(M, Cout, device=DEV)
grads = {p: (torch.zeros_like(p), 1) for p in (conv.weight, bn.weight, bn.bias, bnp.weight, bnp.bias)}
stt = rconv._stat(u, b, y, False, grads)
H.check(lib.mmvae_rc_bn_bwd_stats(H.ptr(G), ctypes.byref(stt), M, Cout, H.stream()), "stats")
nz = lib.mmvae_rc_conv_splits(Min, Cin, Cout, k * k)
rt = 64
nwg = ((Min + rt - 1) // rt) * (Cin // rt) * nz
buf = torch.zeros(nwg * 8, dtype=torch.int64, device=DEV)


def run():
    return rconv._dgrad(u, b, G, y, gm, None, rconv.MASK_BN, x, (bp, bnp.bias), Min, [rconv._stat(up, bp, x, False, grads)])


for _ in range(3):
    run()
torch.cuda.synchronize()
H.check(lib.mmvae_rc_probe(buf.data_ptr()), "probe")
run()
torch.cuda.synchronize()
t = buf.view(nwg, 8).cpu().double() * 0.01        # 100 MHz -> us
t0 = t[:, 0].min()
names = ["entry", "first loads landed", "end of K loop", "tile complete (after split reduce)", "epilogue stores landed",
         "partials stored", "after statistics tail"]
print(f"dgrad H={Hh} Cin={Cin} Cout={Cout} k={k} s={st} B={B}: {nwg} workgroups ({rt}-row tiles, split {nz}); us since the first entry")
for i, nm in enumerate(names):
    v = t[:, i]
    v = v[v > 0] - t0
    if len(v):
        print(f"  {nm:36s} n {len(v):5d}  min {v.min():6.1f}  median {v.median():6.1f}  max {v.max():6.1f}")
