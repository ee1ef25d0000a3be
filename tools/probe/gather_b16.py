"""split-bf16 gather convolution (csrc/conv_gather_b16.inc) against the fp32-MFMA kernel: error vs an fp64 convolution and
graph-timed launches (20 per graph)."""
import ctypes, os, subprocess, sys
import torch
import torch.nn.functional as F
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
so = os.path.join(here, "libgatherb16.so")
L = ctypes.CDLL(so)
sig = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
L.probe_conv_b16.argtypes = sig
L.probe_conv_f32.argtypes = sig


def timed(fn, reps=20, rounds=5):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


if __name__ == "__main__":
  for B in (128, 512, 1000, 2048):
      g = torch.Generator().manual_seed(B)
      x = torch.randn(B, 32, 32, 32, generator=g).cuda()
      w = (torch.randn(32, 32, 4, 4, generator=g) * 0.05).cuda()
      b = torch.randn(32, generator=g).cuda()
      ref = F.conv2d(torch.relu(x.double()), w.double(), b.double(), stride=2, padding=1) if B <= 512 else None
      out = {}
      for name, fn, arg in (("f32 plan5", L.probe_conv_f32, 5), ("f32 plan0", L.probe_conv_f32, 0), ("b16p tm2 cc8", L.probe_conv_b16, 12), ("b16p tm4 cc8", L.probe_conv_b16, 14),
                            ("b16p tm2 cc4", L.probe_conv_b16, 22), ("b16p tm4 cc4", L.probe_conv_b16, 24)):
          y = torch.full((B, 32, 16, 16), float("nan"), device="cuda")
          st = lambda: torch.cuda.current_stream().cuda_stream
          call = lambda: fn(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, 32, 32, 32, 2, 0, arg, st())
          rc = call()
          torch.cuda.synchronize()
          err = float((y.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
          t = timed(call)
          out[name] = y
          print(f"B={B:5d} {name:14s} rc={rc} max rel err vs fp64 {err:.2e}   {t:7.1f} us   {2 * B * 256 * 32 * 512 / t / 1e6:6.1f} TFLOP/s")
      print("      b16 vs f32 max abs diff", float((out["b16p tm4 cc4"] - out["f32 plan5"]).abs().max()))

  print("conv3 shape (32 ch, 16 x 16 -> 8 x 8)")
  for B in (128, 1000):
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 32, 16, 16, generator=g).cuda()
    w = (torch.randn(32, 32, 4, 4, generator=g) * 0.05).cuda()
    b = torch.randn(32, generator=g).cuda()
    ref = F.conv2d(torch.relu(x.double()), w.double(), b.double(), stride=2, padding=1)
    for name, fn, arg in (("f32 plan2", L.probe_conv_f32, 2), ("f32 plan1", L.probe_conv_f32, 1), ("f32 plan0", L.probe_conv_f32, 0), ("b16p tm2 cc4", L.probe_conv_b16, 22),
                          ("b16p tm2 cc8", L.probe_conv_b16, 12), ("b16p tm1 cc4", L.probe_conv_b16, 21), ("b16p tm1 cc8", L.probe_conv_b16, 11)):
        y = torch.full((B, 32, 8, 8), float("nan"), device="cuda")
        call = lambda: fn(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, 32, 32, 16, 2, 0, arg, torch.cuda.current_stream().cuda_stream)
        rc = call()
        torch.cuda.synchronize()
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        print(f"B={B:5d} {name:14s} rc={rc} max rel err vs fp64 {err:.2e}   {timed(call):7.1f} us")

  print("conv1 shape (3 ch, 64 x 64 -> 32 x 32)")
  for B in (128, 1000):
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 3, 64, 64, generator=g).cuda()
    w = (torch.randn(32, 3, 4, 4, generator=g) * 0.1).cuda()
    b = torch.randn(32, generator=g).cuda()
    ref = F.conv2d(torch.relu(x.double()), w.double(), b.double(), stride=2, padding=1)
    for name, fn, arg in (("f32", L.probe_conv_f32, 0), ("b16p tm4 cc3", L.probe_conv_b16, 0)):
        y = torch.full((B, 32, 32, 32), float("nan"), device="cuda")
        call = lambda: fn(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, 3, 32, 64, 2, 0, arg, torch.cuda.current_stream().cuda_stream)
        rc = call()
        torch.cuda.synchronize()
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        print(f"B={B:5d} {name:13s} rc={rc} max rel err vs fp64 {err:.2e}   {timed(call):7.1f} us")
