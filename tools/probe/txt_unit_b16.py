"""GEMM skeleton of the wave-per-sequence text layer on fp32 MFMA vs split-bf16 MFMA (txt_unit_b16.hip): 36 weight units per
launch (an encoder layer has 32 + 4 attention units), one wave per sequence, graph-timed (20 launches per graph)."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
from gather_b16 import timed  # noqa: E402
so = os.path.join(here, "libtxtunit.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "txt_unit_b16.hip")])
L = ctypes.CDLL(so)
L.probe_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
L.probe_b16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
U = 36
W = (torch.randn(U, 32, 32) * 0.2).cuda()
Wimg = torch.zeros(U * 3 * 1024, dtype=torch.int16).cuda()      # (timing only: the image's content does not matter)
for N in (128, 1000):
    out = torch.empty(N * 16 * 64).cuda()
    st = lambda: torch.cuda.current_stream().cuda_stream
    a = timed(lambda: L.probe_f32(W.data_ptr(), out.data_ptr(), N, st()))
    b = timed(lambda: L.probe_b16(Wimg.data_ptr(), out.data_ptr(), N, st()))
    print(f"{N:5d} sequences, {U} units: fp32 MFMA {a:6.2f} us   split-bf16 {b:6.2f} us   ({a / b:.2f} x)")
