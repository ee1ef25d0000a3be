"""stride-2 data gradients of the ResNet engine, parity-class row order (what ships) against raster order:
python tools/probe/rc_s2_time.py [B]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodal_vae_comparison_amd import hipops as H, rconv  # noqa: E402
from multimodal_vae_comparison_amd.models.resnet import ConvW, BatchNorm2d  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
DEV = torch.device("cuda:0")
REP = 20


def timeit(fn):
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


for name, Hh, Cin, Cout, k, st in [("l2.c2s", 16, 128, 128, 3, 2), ("l2.ds", 16, 256, 512, 1, 2), ("l3.c2s", 8, 256, 256, 3, 2),
                                   ("l4.c2s", 4, 512, 512, 3, 2), ("l2.c2", 8, 128, 128, 3, 1)]:
    conv = ConvW(Cin, Cout, k, st, k // 2, channels_last=True).to(DEV)
    bn, bnp = BatchNorm2d(Cout).to(DEV), BatchNorm2d(Cin).to(DEV)
    u, up = rconv.Unit(conv, bn), rconv.Unit(ConvW(64, Cin, 1, 1, 0).to(DEV), bnp)
    Ho = (Hh - 1) // st + 1
    Min, M = B * Hh * Hh, B * Ho * Ho
    gm = (Hh, Hh, k, st, k // 2)
    x = torch.randn(Min, Cin, device=DEV)
    bp = up.buffers(Min, DEV)
    bp["mean"].zero_(); bp["sc"].fill_(1.0); bp["rstd"].fill_(1.0)
    y, b = rconv._fwd(u, x, Min, M, rconv.PRE_BN_RELU, (bp, bnp.bias), gm, False)
    G = torch.randn(M, Cout, device=DEV)
    grads = {p: (torch.zeros_like(p), 1) for p in (conv.weight, bn.weight, bn.bias, bnp.weight, bnp.bias)}
    stt = rconv._stat(u, b, y, False, grads)
    H.check(H.lib().mmvae_rc_bn_bwd_stats(H.ptr(G), ctypes.byref(stt), M, Cout, H.stream()), "stats")
    res = []
    for rm in ([rconv.parity_row_map(DEV, B, Hh, Hh), None] if (st == 2 and k > 1) else [None]):
        def run():
            jd, _ = rconv.dgrad_job(u, b, G, y, gm, None, None, rconv.MASK_BN, x, (bp, bnp.bias), Min,
                                    [rconv._stat(up, bp, x, False, grads)], row_map=rm)
            rconv.launch(jd)
        res.append(timeit(run))
    fl = 2.0 * M * Cin * Cout * k * k
    print(f"{name:7s} B={B} dgrad us: " + "  ".join(f"{'class order' if (i == 0 and len(res) > 1) else 'raster'} {t:7.1f} ({fl / t * 1e-6:5.1f} TF/s)" for i, t in enumerate(res)))
