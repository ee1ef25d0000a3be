"""Per-shape timing of the fused convolution + BatchNorm engine (csrc/rconv.hip) over the ResNet-50 bottleneck shapes:
forward / data gradient / weight gradient of every distinct (rows, Cin, Cout, k, stride) at batch B (default 24).
usage: python tools/probe/rc_time.py [B]"""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodal_vae_comparison_amd import hipops as H, rconv  # noqa: E402
from multimodal_vae_comparison_amd.models.resnet import ConvW, BatchNorm2d  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
DEV = torch.device("cuda:0")
REP = 30


def timeit(fn):
    """REP back-to-back launches captured in a hipGraph (no host gaps), replayed 3 times; us per launch"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(REP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * REP) * 1e3


# (name, H of the input map, Cin, Cout, k, stride)
SHAPES = [("l1.c1a", 16, 64, 64, 1, 1), ("l1.c1", 16, 256, 64, 1, 1), ("l1.c2", 16, 64, 64, 3, 1), ("l1.c3", 16, 64, 256, 1, 1),
          ("l2.c1a", 16, 256, 128, 1, 1), ("l2.c2s", 16, 128, 128, 3, 2), ("l2.ds", 16, 256, 512, 1, 2),
          ("l2.c1", 8, 512, 128, 1, 1), ("l2.c2", 8, 128, 128, 3, 1), ("l2.c3", 8, 128, 512, 1, 1),
          ("l3.c1a", 8, 512, 256, 1, 1), ("l3.c2s", 8, 256, 256, 3, 2), ("l3.ds", 8, 512, 1024, 1, 2),
          ("l3.c1", 4, 1024, 256, 1, 1), ("l3.c2", 4, 256, 256, 3, 1), ("l3.c3", 4, 256, 1024, 1, 1),
          ("l4.c1a", 4, 1024, 512, 1, 1), ("l4.c2s", 4, 512, 512, 3, 2), ("l4.ds", 4, 1024, 2048, 1, 2),
          ("l4.c1", 2, 2048, 512, 1, 1), ("l4.c2", 2, 512, 512, 3, 1), ("l4.c3", 2, 512, 2048, 1, 1)]

if os.environ.get("ONLY"):
    SHAPES = [sh for sh in SHAPES if sh[0] in os.environ["ONLY"].split(",")]
tot = [0.0, 0.0, 0.0]
print(f"B = {B}\n{'layer':8s} {'rows':>6s} {'Cin':>5s} {'Cout':>5s} k s | {'fwd us':>8s} {'TF/s':>6s} {'no-stat':>7s} | {'dgrad us':>8s} {'TF/s':>6s} | "
      f"{'wgrad us':>8s} {'TF/s':>6s} | {'d+w us':>7s} nz")
for name, Hh, Cin, Cout, k, st in SHAPES:
    conv = ConvW(Cin, Cout, k, st, k // 2, channels_last=True).to(DEV)
    bn, bnp = BatchNorm2d(Cout).to(DEV), BatchNorm2d(Cin).to(DEV)
    u, up = rconv.Unit(conv, bn), rconv.Unit(ConvW(64, Cin, 1, 1, 0).to(DEV), bnp)
    Ho = (Hh - 1) // st + 1
    Min, M = B * Hh * Hh, B * Ho * Ho
    tf, tb = (rconv.tables(DEV, B, Hh, Hh, k, st, k // 2) if (k > 1 or st > 1) else (None, None))
    gm = (Hh, Hh, k, st, k // 2) if (k > 1 or st > 1) else rconv.IDENT
    x = torch.randn(Min, Cin, device=DEV)
    bp = up.buffers(Min, DEV)
    bp["mean"].zero_(); bp["sc"].fill_(1.0); bp["rstd"].fill_(1.0)
    y, b = rconv._fwd(u, x, Min, M, rconv.PRE_BN_RELU, (bp, bnp.bias), gm, False)
    G = torch.randn(M, Cout, device=DEV)
    grads = {p: (torch.zeros_like(p), 1) for p in (conv.weight, bn.weight, bn.bias, bnp.weight, bnp.bias)}
    stt = rconv._stat(u, b, y, False, grads)
    H.check(H.lib().mmvae_rc_bn_bwd_stats(H.ptr(G), ctypes.byref(stt), M, Cout, H.stream()), "stats")
    t_f = timeit(lambda: rconv._fwd(u, x, Min, M, rconv.PRE_BN_RELU, (bp, bnp.bias), gm, False))
    def fwd_nostat():
        j, _, _ = rconv.fwd_job(u, x, M, rconv.PRE_BN_RELU, (bp, bnp.bias), gm, False)
        j.f.part = None
        rconv.launch(j)
    t_n = timeit(fwd_nostat)

    def pair():
        jd, _ = rconv.dgrad_job(u, b, G, y, gm, None, None, rconv.MASK_BN, x, (bp, bnp.bias), Min, [rconv._stat(up, bp, x, False, grads)])
        rconv.launch(jd, rconv.wgrad_job(u, b, G, y, x, rconv.PRE_BN_RELU, (bp, bnp.bias), tf, grads))
    t_p = timeit(pair)
    t_d = timeit(lambda: rconv._dgrad(u, b, G, y, gm, None, rconv.MASK_BN, x, (bp, bnp.bias), Min,
                                      [rconv._stat(up, bp, x, False, grads)]))
    t_w = timeit(lambda: rconv._wgrad(u, b, G, y, x, rconv.PRE_BN_RELU, (bp, bnp.bias), tf, grads))
    fl = 2.0 * M * Cin * Cout * k * k
    nz = (H.lib().mmvae_rc_conv_splits(M, Cout, Cin, k * k), H.lib().mmvae_rc_conv_splits(Min, Cin, Cout, k * k),
          H.lib().mmvae_rc_wgrad_splits(M, Cin, Cout, k * k))
    tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
    print(f"{name:8s} {M:6d} {Cin:5d} {Cout:5d} {k} {st} | {t_f:8.1f} {fl / t_f * 1e-6:6.1f} {t_n:7.1f} | {t_d:8.1f} {fl / t_d * 1e-6:6.1f} | "
          f"{t_w:8.1f} {fl / t_w * 1e-6:6.1f} | {t_p:7.1f} {nz}")
print(f"sum over the distinct shapes: fwd {tot[0]:.0f} us, dgrad {tot[1]:.0f} us, wgrad {tot[2]:.0f} us (graph replays of back-to-back launches)")
