"""Out-of-bounds guard for the fused conv backward launches: every buffer the kernel gets lives inside a larger one whose
margins hold a sentinel; after the launch the margins must be untouched (writes) -- run per shape / batch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multimodal_vae_comparison_amd import hipops as H

PAD = 1 << 16
SENT = 1234.5


def guarded(n, fill=None):
    big = torch.full((n + 2 * PAD,), SENT, device="cuda")
    v = big[PAD:PAD + n]
    if fill is None:
        v.normal_()
    else:
        v.fill_(fill)
    return big, v


def check(name, big, n):
    lo, hi = big[:PAD], big[PAD + n:]
    bad = int((lo != SENT).sum()) + int((hi != SENT).sum())
    if bad:
        il = (lo != SENT).nonzero().flatten()
        ih = (hi != SENT).nonzero().flatten()
        print(f"   !! {name}: {bad} guard floats overwritten; below: {il[:3].tolist()}..{il[-3:].tolist() if len(il) else []} "
              f"above: {ih[:3].tolist()}..{ih[-3:].tolist() if len(ih) else []}")
    return bad


def run(kind, B, Cin, Cout, Hout):
    lib = H.lib()
    st = torch.cuda.current_stream().cuda_stream
    Hin = 2 * Hout
    if kind == "conv2d":      # y (B,Cout,Hout,Hout) = conv(x (B,Cin,Hin,Hin))
        nx, ny = B * Cin * Hin * Hin, B * Cout * Hout * Hout
        nws = lib.mmvae_conv_wgrad_ws_floats(B, Cout, Cin, Hout)      # (B, small-map channels, large-map channels, small map)
        fn = "mmvae_conv2d_k4s2_bwd"
    else:                     # convT: y (B,Cout,Hin,Hin) = convT(x (B,Cin,Hout,Hout)); weights (Cin,Cout,4,4)
        nx, ny = B * Cin * Hout * Hout, B * Cout * Hin * Hin
        nws = lib.mmvae_conv_wgrad_ws_floats(B, Cin, Cout, Hout)
        fn = "mmvae_convT2d_k4s2_bwd"
    bufs = {"dy": guarded(ny), "x": guarded(nx), "w": guarded(Cout * Cin * 16), "dx": guarded(nx, 0.0),
            "ws": guarded(nws, 0.0), "dw": guarded(Cout * Cin * 16, 0.0), "db": guarded(Cout, 0.0)}
    p = {k: v[1].data_ptr() for k, v in bufs.items()}
    rc = getattr(lib, fn)(p["dy"], p["x"], p["w"], p["dx"], p["dw"], p["db"], p["ws"], B, Cin, Cout, Hout, H.ACT_RELU, H.ACC_DEFER, st)
    torch.cuda.synchronize()
    bad = sum(check(k, v[0], v[1].numel()) for k, v in bufs.items())
    print(f"{kind} B={B} Cin={Cin} Cout={Cout} Hout={Hout} rc={rc} ws={nws} -> {'OK' if not bad else 'OUT OF BOUNDS'}")


if __name__ == "__main__":
    for B in (24, 128, 256, 512, 1000, 4096):
        for kind, Cin, Cout, Hout in (("conv2d", 3, 32, 32), ("conv2d", 32, 32, 16), ("conv2d", 32, 32, 8), ("conv2d", 32, 32, 4),
                                      ("convT", 32, 32, 4), ("convT", 32, 32, 8), ("convT", 32, 32, 16), ("convT", 32, 3, 32)):
            try:
                run(kind, B, Cin, Cout, Hout)
            except Exception as e:       # noqa: BLE001
                print(kind, B, Cin, Cout, Hout, "error", e)
